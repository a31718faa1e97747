#!/bin/bash
# Ablation builds of libnsr_hip.so (never shipped): tools/build_abl.sh NAME -DNSR_ABL_...  -> tools/abl/libnsr_NAME.so
# run with NSR_LIB_PATH=tools/abl/libnsr_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
objs=$(ls nerfstyle_amd/csrc/_obj/*.o | grep -v "field_bwd.o\|table_scatter.o")
for f in field_bwd table_scatter; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function "$@" -c nerfstyle_amd/csrc/$f.hip -o tools/abl/${f}_$name.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/abl/libnsr_$name.so $objs tools/abl/field_bwd_$name.o tools/abl/table_scatter_$name.o
echo tools/abl/libnsr_$name.so
