// Translation unit of the GOUT ("gradients out") instantiations of the fused field backward: same source as
// field_bwd.hip, compiled with  -DNSR_BWD_ASM_WGRAD=1 -mllvm --amdgpu-mfma-vgpr-form  (see the note above
// nsr_field_bwd_launch_gout in field_bwd.hip and nerfstyle_amd/build.py).
#define NSR_BWD_TU_GOUT 1
#include "field_bwd.hip"
