// Internal interface between the fused field backward (field_bwd.hip) and the stand-alone table scatter (table_scatter.hip).
#pragma once
#include "nsr_common.h"

// true when the level table fits the scatter's LDS lattices (10-bit sample-order blocks)
bool nsr_table_scatter_supported(const NsrLevel *levels);
// gin: [M][16] float4 encoder-output gradients written by the GOUT backward; perm: nsr_sample_order's permutation
int nsr_table_scatter_launch(const NsrLevel *levels, const float *bmin, const float *bsize, const float *xyzs, const uint32_t *perm,
                             const int32_t *m_dev, uint32_t M, const void *gin, float *grad_tables, int train_density,
                             int train_color, hipStream_t stream);
