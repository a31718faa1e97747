#!/bin/bash
# Ablation / tuning builds of libnsr_hip.so (never shipped): tools/build_abl.sh NAME -DNSR_ABL_...  -> tools/abl/libnsr_NAME.so
# (the forward, the backward kernels and the table scatter are rebuilt with the extra flags, everything else is reused from csrc/_obj);
# run with NSR_LIB_PATH=<copy of the .so under tools/abl_run/> -- tools/abl/ does not travel to the GPU box.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/abl tools/abl_run
objs=$(ls nerfstyle_amd/csrc/_obj/*.o | grep -v "field_bwd.o\|field_bwd_gout.o\|table_scatter.o\|/field.o")
H="/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function"
$H "$@" -c nerfstyle_amd/csrc/field_bwd.hip -o tools/abl/field_bwd_$name.o &
$H -DNSR_BWD_ASM_WGRAD=1 -mllvm --amdgpu-mfma-vgpr-form "$@" -c nerfstyle_amd/csrc/field_bwd_gout.hip -o tools/abl/field_bwd_gout_$name.o &
$H "$@" -c nerfstyle_amd/csrc/table_scatter.hip -o tools/abl/table_scatter_$name.o &
$H -mllvm --amdgpu-mfma-vgpr-form "$@" -c nerfstyle_amd/csrc/field.hip -o tools/abl/field_$name.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/abl/libnsr_$name.so $objs tools/abl/field_bwd_$name.o tools/abl/field_bwd_gout_$name.o tools/abl/table_scatter_$name.o tools/abl/field_$name.o
echo tools/abl/libnsr_$name.so
