#!/bin/bash
# A/B builds of the library with extra flags for the production render kernels:
#   tools/ab_build.sh NAME "-DMPT_POOL_LEAVE=24 ..."   ->  ptina_amd/libmiptina_NAME.so   (run with MIPTINA_LIB=...)
# The objects that do not depend on the flags are copied from the main build.
set -e
name=$1; shift
cd "$(dirname "$0")/../ptina_amd/csrc"
mkdir -p _obj_ab_$name
for f in render_strict unit_fast unit_strict aux_kernels lbvh_build sah_build wide_build miptina tree_build oct_build comm; do
  [ -f _obj/$f.o ] && cp -p _obj/$f.o _obj_ab_$name/ || true
done
make -s OBJ=_obj_ab_$name OUT=../libmiptina_$name.so FASTFLAGS="-fno-slp-vectorize $*" 2>&1 | grep -E "error|Error" -A5 || true
ls -la ../libmiptina_$name.so | awk '{print $5, $9}'
