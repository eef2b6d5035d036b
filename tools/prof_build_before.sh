# GPU: kernel traces of mpt_build_tree with the round-5 library (ptina_amd/libmiptina_r05.so, built from commit 475e75e) for the before / after table
export TMPDIR=/tmp
mkdir -p gpurun_out
export MIPTINA_LIB=$GRAFT_REPO_ROOT/ptina_amd/libmiptina_r05.so BUILD_TAG=_before
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/build_c5_before -o c5 -- python3 tools/build_profile.py c5 3 > gpurun_out/build_c5_before.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/build_c4_before -o c4 -- python3 tools/build_profile.py c4 3 > gpurun_out/build_c4_before.log 2>&1
grep -h wall_ms gpurun_out/build_c5_before.log gpurun_out/build_c4_before.log
