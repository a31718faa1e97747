#!/bin/bash
# Ablation builds of libnsr_hip.so (never shipped): tools/build_abl.sh NAME -DNSR_ABL_...  -> tools/abl/libnsr_NAME.so
# run with NSR_LIB_PATH=tools/abl/libnsr_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
objs=$(ls nerfstyle_amd/csrc/_obj/*.o | grep -v field_bwd.o)
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function "$@" -c nerfstyle_amd/csrc/field_bwd.hip -o tools/abl/field_bwd_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/abl/libnsr_$name.so $objs tools/abl/field_bwd_$name.o
echo tools/abl/libnsr_$name.so
