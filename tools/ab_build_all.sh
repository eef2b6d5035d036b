#!/bin/bash
# A/B build of the WHOLE library with extra flags for every object: tools/ab_build_all.sh NAME "-DSB_BINS=64"
set -e
name=$1; shift
cd "$(dirname "$0")/../ptina_amd/csrc"
make -s -j4 OBJ=_obj_ab_$name OUT=../libmiptina_$name.so EXTRA="$*" 2>&1 | grep -E "error|Error" -A5 || true
ls -la ../libmiptina_$name.so | awk '{print $5, $9}'
