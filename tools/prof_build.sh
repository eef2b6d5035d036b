# GPU: the build tests, then kernel traces of mpt_build_tree at BASELINE configs 5 and 4 (tools/build_profile.py)
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" != notests ]; then
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "sah or lbvh_build or wide_collapse or tree_was_built" > gpurun_out/build_tests.log 2>&1
echo "tests rc=$?" ; tail -n 15 gpurun_out/build_tests.log
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/build_c5 -o c5 -- python3 tools/build_profile.py c5 3 > gpurun_out/build_c5.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/build_c4 -o c4 -- python3 tools/build_profile.py c4 3 > gpurun_out/build_c4.log 2>&1 &&
BUILD_PHASES=0 timeout -k 10 300 python3 tools/build_profile.py c5 3 > gpurun_out/build_c5_nophase.log 2>&1
grep -h wall_ms gpurun_out/build_c5.log gpurun_out/build_c4.log gpurun_out/build_c5_nophase.log
