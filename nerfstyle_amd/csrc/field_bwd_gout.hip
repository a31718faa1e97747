// Translation unit of the GOUT ("gradients out") instantiations of the fused field backward: same source as
// field_bwd.hip, compiled with  -DNSR_BWD_ASM_WGRAD=1 -mllvm --amdgpu-mfma-vgpr-form  (see the note above
// nsr_field_bwd_launch_gout in field_bwd.hip and nerfstyle_amd/build.py).
//
// Knobs that differ from the fused-tracker unit, measured on the bench frame (48.6 M samples, this kernel alone; rocprofv3 SQ
// counters before: 51 % of the wave's cycles parked in s_waitcnt at one wave per SIMD):
//   NSR_BWD_EARLY_NEXT  the next tile's inputs are requested right after this tile's first layer, and every member of
//                       the current tile's inputs is waited for at the loop top (else: vmcnt(0) mid-tile)   13.2 -> 11.7 ms
//   NSR_MM_AHEAD / NSR_BWD_WQ  weight-fragment LDS reads run ahead of the MFMA stream: four in flight inside a layer, the
//                       next layer's first four requested before the previous layer's packing code          11.7 -> 10.7 ms
//   NSR_BWD_PKMASK      the backward's ReLU masks on packed 16-bit pairs (3 packed instructions per pair instead of a
//                       compare + select per element; 1412 -> 1304 instructions per tile)                  10.6 -> 10.0 ms
// An 8-deep queue for the two 8-fragment layers: no change (measured).  The wgrad operand transposes through LDS
// (ds_write_b64 + ds_read_b64_tr_b16, 49 per tile) instead of an MFMA with the identity + re-rounding: 1304 -> 1240
// instructions per tile and the same time (21.4 vs 21.5 ms for the pair) -- the LDS round trips cost what the MFMAs did.
// (the tracker unit keeps them off -- its scatter already separates the loads from their use, and the read-ahead costs it
// registers: 49.4 -> 51.2 ms).
#define NSR_BWD_TU_GOUT 1
#ifndef NSR_BWD_EARLY_NEXT
#define NSR_BWD_EARLY_NEXT 1
#endif
#ifndef NSR_MM_AHEAD
#define NSR_MM_AHEAD 4
#endif
#ifndef NSR_BWD_WQ
#define NSR_BWD_WQ 1
#endif
#ifndef NSR_BWD_PKMASK
#define NSR_BWD_PKMASK 1
#endif
#include "field_bwd.hip"
