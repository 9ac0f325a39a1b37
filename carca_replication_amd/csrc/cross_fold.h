// Arguments shared by the eval-mode scoring kernels (final LayerNorm + CrossAttentionBlock.forward with decoder.ffn folded
// into the value projection, carca.py:421, :338-349): cross_fold_kernel (cross_score.hip: one or two workgroups per user,
// the latency regime B <= #CUs) and cross_stream_kernel (cross_stream.hip: persistent workgroups that pipeline users,
// B > #CUs).
#pragma once
#include "attn_common.h"

struct FoldArgs {
  const float* p_raw;
  const int32_t* p_ids;
  float* p_normed;
  CarcaTargetGroup g[CARCA_MAX_GROUPS];
  int tile_start[CARCA_MAX_GROUPS + 1];
  int ngroups, ldp, ldo, L, d, residual, nparts;
  const float *ln_w, *ln_b, *wq, *bq, *wk, *bk, *wu, *cu, *ffn_w, *ffn_b;
  float qscale;  // log2(e) / sqrt(dh): scores leave the Q projection in the exp2 domain
  int dbg;       // timing experiments (tuning key 5; wrong results): bit 1 no W_Q traffic, 2 no tile traffic, 3 no LayerNorm,
                 // 4 no W_K traffic
  unsigned long long* stamps;
  int opt;       // cross_stream_kernel: bit 0 = deal a step's later jobs by ticket (tuning key 3; A/B)
  int B;         // users (cross_stream_kernel: a workgroup takes users blockIdx.x, blockIdx.x + gridDim.x, ...)
};
#define FOLD_NEG (-1.0e30f)

// LDS-DMA: 64 lanes x 16 B from per-lane global offsets into 1 KB of LDS at lds_dst + 16 * lane (no registers)
__device__ __forceinline__ void dma16(const float* base, int lane_elem_off, int uniform_elem_off, float* lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(carca_rsrc(base), (__attribute__((address_space(3))) void*)lds_dst, 16,
                                           lane_elem_off * 4, uniform_elem_off * 4, 0, 0);
}

// cross_stream.hip: CARCA_ERR_UNSUPPORTED (and no launch) when no instantiation covers (dpi, dhp, H)
int carca_cross_stream_launch(const FoldArgs& fa, int dpi, int dhp, int H, int B, void* stream);
