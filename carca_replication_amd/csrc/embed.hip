// K0/K1: mask + AllEmbedding forward for the profile and every target group in one call.
//
//   reference: get_mask utils.py:6-7, AllEmbedding.forward carca.py:85-95, encodings carca.py:25-31,54-60
//
// Three launches on the caller's stream:
//   1. gather   zq[r, 0:d]   = items_w[ids[r]] * sqrt(d)                       (HBM/latency bound, <1% of bytes)
//   2. feat     zq[r, d:d+g] = [attrs[r] ; ctx[r]] . feats_w^T + feats_b        (MFMA bound: 97% of the model's flops)
//   3. joint    e[r, :]      = (zq[r] . joint_w^T + joint_b (+ pos[r % T])) * (ids[r] != 0)
// 2 and 3 are carca_gemm_rows (gemm.hip): rows of all segments (profile, target groups) in one launch,
// k over the two column sources [attrs | ctx] without concatenating them in memory.
// The backward counterparts (scatter-add into items_w's gradient, column sums) live here too.
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

// (a wave takes RIF rows at a time, lanes over columns.  RIF = 4 with one round per wave: a row is a chain of two dependent
// round trips -- id, then the row -- and with one row per wave and 2.4 rounds per wave the kernel was 8.2 us of latency for
// 14 MB of traffic; it has its own launch again since the feature GEMM gave its idle CU's slack away, gemm.hip)
template <int RIF>
__global__ void gather_items_kernel(const CarcaGatherArgs ga) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  carca_gather_rows<RIF>(ga, wave, nwaves, lane);
}

}  // namespace

extern "C" int carca_embed_fwd(const CarcaRowSeg* segs, int nseg, int n_attrs, int n_ctx, int d, int g,
                               const float* items_w, const float* feats_w, const float* feats_b,
                               const float* joint_w, const float* joint_b, const float* pos, float* zq, int ld_e,
                               int stages, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(segs && nseg >= 1 && nseg <= CARCA_MAX_SEGS, "embed_fwd: nseg=%d outside 1..%d", nseg, CARCA_MAX_SEGS);
  CARCA_CHECK_ARG(items_w && feats_w && feats_b && joint_w && joint_b && zq, "embed_fwd: null weight/workspace");
  CARCA_CHECK_ARG(n_attrs >= 1 && n_ctx >= 0 && d >= 1 && g >= 1, "embed_fwd: bad dims");
  CARCA_CHECK_ARG(ld_e >= d, "embed_fwd: ld_e=%d < d=%d", ld_e, d);
  const int ldz = d + g;

  CarcaGatherArgs ga{};
  CarcaGemmDesc fa{}, ja{};
  int row_start = 0;
  for (int s = 0; s < nseg; ++s) {
    const CarcaRowSeg& sg = segs[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.T >= 1 && sg.ids && (sg.attrs || sg.attrs_table) && sg.e_out &&
                        (n_ctx == 0 || sg.ctx),
                    "embed_fwd: segment %d has null pointers or no rows", s);
    CARCA_CHECK_ARG(!sg.add_pos || pos, "embed_fwd: segment %d wants pos but pos is NULL", s);
    ga.ids[s] = sg.ids;
    ga.row_start[s] = row_start;
    CarcaGemmSeg& f = fa.seg[s];
    f.a0 = sg.attrs_table ? sg.attrs_table : sg.attrs; f.a1 = sg.ctx;
    f.a0_gather = sg.attrs_table ? max(1, sg.attrs_table_rows) : 0; f.c = zq + (size_t)row_start * ldz + d; f.ids = sg.ids;
    f.rows = sg.rows; f.T = sg.T; f.add_pos = 0;
    f.a0_bstride = sg.attrs_table ? 0 : sg.attrs_bstride; f.a1_bstride = sg.ctx_bstride;
    CarcaGemmSeg& j = ja.seg[s];
    j.a0 = zq + (size_t)row_start * ldz; j.a1 = nullptr; j.c = sg.e_out; j.ids = sg.ids;
    j.rows = sg.rows; j.T = sg.T; j.add_pos = sg.add_pos;
    row_start += sg.rows;
  }
  ga.row_start[nseg] = row_start;
  ga.nseg = nseg;
  const int total_rows = row_start;

  ga.items_w = items_w; ga.zq = zq; ga.d = d; ga.ldz = ldz; ga.total_rows = total_rows; ga.scale = (float)sqrt((double)d);
  auto launch_gather = [&]() -> int {
    const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);  // (18: one row per wave, 2048 blocks at most -- A/B switch)
    if (variant == 18) {
      hipLaunchKernelGGL(gather_items_kernel<1>, dim3(min((total_rows + 3) / 4, 2048)), dim3(256), 0, stream, ga);
    } else {
      hipLaunchKernelGGL(gather_items_kernel<4>, dim3(min((total_rows + 15) / 16, 8192)), dim3(256), 0, stream, ga);
    }
    CARCA_LAUNCH_CHECK();
    return CARCA_OK;
  };
  fa.nseg = nseg;
  fa.lda0 = n_attrs; fa.lda1 = n_ctx; fa.K0 = n_attrs; fa.K1 = n_ctx;
  fa.bt0 = feats_w; fa.ldb0 = n_attrs + n_ctx;
  fa.bt1 = feats_w + n_attrs; fa.ldb1 = n_attrs + n_ctx;
  fa.N = g; fa.ldc = ldz; fa.ncols_out = g; fa.bias = feats_b;
  // q of a padding slot (id 0) is never used -- e is masked (carca.py:92-94), the backward sees de = 0 there --, so the
  // product may leave those rows out (gemm_rows_skc_kernel) and write zeros instead
  fa.mask_rows = 1;
  if ((stages & CARCA_EMBED_GATHER) && (stages & CARCA_EMBED_FEAT)) {
    // both asked for in one call: the gather rides in the feature GEMM's launch when that leaves a CU idle
    int rode = 0;
    const int rc = carca_gemm_rows_passenger(&fa, &ga, &rode, stream_);  // q = [attrs ; ctx] W_f^T + b_f  (carca.py:86)
    if (rc != CARCA_OK) return rc;
    if (!rode)
      if (int rc2 = launch_gather()) return rc2;
  } else if (stages & CARCA_EMBED_GATHER) {
    if (int rc = launch_gather()) return rc;
  } else if (stages & CARCA_EMBED_FEAT) {
    const int rc = carca_gemm_rows(&fa, stream_);
    if (rc != CARCA_OK) return rc;
  }
  if (stages & CARCA_EMBED_JOINT) {  // e = ([z ; q] W_j^T + b_j (+ pos)) * mask  (carca.py:89-94)
    ja.nseg = nseg;
    ja.lda0 = ldz; ja.K0 = ldz; ja.bt0 = joint_w; ja.ldb0 = ldz;
    ja.N = d; ja.ldc = ld_e; ja.ncols_out = ld_e; ja.bias = joint_b; ja.pos = pos; ja.mask_rows = 1;
    const int rc = carca_gemm_rows(&ja, stream_);
    if (rc != CARCA_OK) return rc;
  }
  return CARCA_OK;
}

// The joint product with the item term taken from a PROJECTED table (CarcaForwardDesc.z_table, inference):
//   e = (q W_jq^T + z_table[id] + b_j (+ pos)) * mask,   z_table = sqrt(d) items_w W_jz^T,
// over the g columns of q alone -- zq's z columns are never written (no gather launch) and never read.
int carca_embed_joint_ztab(const CarcaRowSeg* segs, int nseg, int d, int g, const float* joint_w, const float* joint_b,
                           const float* pos, const float* zq, int ld_e, const float* z_table, int ld_z_table,
                           void* stream_) {
  CARCA_CHECK_ARG(segs && nseg >= 1 && nseg <= CARCA_MAX_SEGS && joint_w && joint_b && zq && z_table && ld_z_table >= d,
                  "embed_joint: bad arguments");
  const int ldz = d + g;
  CarcaGemmDesc ja{};
  int row_start = 0;
  for (int s = 0; s < nseg; ++s) {
    const CarcaRowSeg& sg = segs[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.T >= 1 && sg.ids && sg.e_out, "embed_joint: segment %d malformed", s);
    CarcaGemmSeg& j = ja.seg[s];
    j.a0 = zq + (size_t)row_start * ldz + d; j.c = sg.e_out; j.ids = sg.ids;
    j.rows = sg.rows; j.T = sg.T; j.add_pos = sg.add_pos;
    row_start += sg.rows;
  }
  ja.nseg = nseg;
  ja.lda0 = ldz; ja.K0 = g; ja.bt0 = joint_w + d; ja.ldb0 = ldz;
  ja.N = d; ja.ldc = ld_e; ja.ncols_out = ld_e; ja.bias = joint_b; ja.pos = pos; ja.mask_rows = 1;
  ja.add_table = z_table; ja.ld_add_table = ld_z_table;
  return carca_gemm_rows(&ja, stream_);
}
