// K0/K1: mask + AllEmbedding forward for the profile and every target group in one call.
//
//   reference: get_mask utils.py:6-7, AllEmbedding.forward carca.py:85-95, encodings carca.py:25-31,54-60
//
// Three launches on the caller's stream:
//   1. gather   zq[r, 0:d]   = items_w[ids[r]] * sqrt(d)                       (HBM/latency bound, <1% of bytes)
//   2. feat     zq[r, d:d+g] = [attrs[r] ; ctx[r]] . feats_w^T + feats_b        (MFMA bound: 97% of the model's flops)
//   3. joint    e[r, :]      = (zq[r] . joint_w^T + joint_b (+ pos[r % T])) * (ids[r] != 0)
// 2 and 3 are the same tiled fp32-MFMA kernel, D[m][n] = sum_k A[m][k] * Bt[n][k]:
//   - A rows come from up to 4 row segments (profile, target groups) so that one launch fills the chip;
//   - k runs over up to two column sources ([attrs | ctx] are never concatenated in memory);
//   - block tile BM x BN, one wave per 32 x BN strip, v_mfma_f32_32x32x2_f32, operands staged
//     global -> registers -> LDS (rows padded to BK+4 floats: conflict-free ds_read_b128);
//   - blocks that share an A row block get ids 8 apart, i.e. the same XCD under round-robin
//     placement, so the A tile is fetched from HBM once and re-read from that XCD's L2 (speed only).
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

struct SegDev {
  const float* a0;     // k-source 0 rows [rows, lda0]
  const float* a1;     // k-source 1 rows [rows, lda1] (or null)
  float* c;            // output rows [rows, ldc] (already offset to the first output column)
  const int32_t* ids;  // [rows]
  int rows, T, add_pos, rb_start;
};

struct GemmArgs {
  SegDev seg[CARCA_MAX_SEGS];
  int nseg, nrb, ncb;
  int lda0, lda1, K0, K1;
  const float* Bt;  // [N, ldb]
  int ldb, N, ldc, ncols_out;  // columns N..ncols_out-1 are written as zeros (JOINT only)
  const float* bias;
  const float* pos;  // [T, N] or null
};

constexpr int MODE_FEAT = 0, MODE_JOINT = 1;

template <int BM, int BN, int BK, int MODE>
__global__ __launch_bounds__((BM / 32) * 64) void gemm_rows_kernel(const GemmArgs args) {
  constexpr int NW = BM / 32, NT = NW * 64, LS = BK + 4, TN = BN / 32;
  constexpr int C4 = BK / 4;  // float4 slots per tile row
  constexpr int A_SLOTS = BM * C4, B_SLOTS = BN * C4;
  constexpr int A_PER = (A_SLOTS + NT - 1) / NT, B_PER = (B_SLOTS + NT - 1) / NT;
  static_assert(BK % 8 == 0 && BN % 32 == 0 && BM % 32 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float As[BM * LS];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LS];

  // ---- block id -> (row block, col block); same row block => same id mod 8 (same XCD) ----------
  const int id = blockIdx.x;
  const int per = 8 * args.ncb;
  const int grp = id / per, j = id - grp * per;
  const int cb = j >> 3, rb = grp * 8 + (j & 7);
  if (rb >= args.nrb) return;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < args.nseg && rb >= args.seg[i].rb_start) s = i;
  const SegDev sg = args.seg[s];
  const int row0 = (rb - sg.rb_start) * BM;  // first row of this block inside the segment
  const int n0 = cb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = (args.K0 + BK - 1) / BK, nt1 = (args.K1 + BK - 1) / BK;
  const int ntiles = nt0 + nt1;

  f32x4 ra[A_PER], rbv[B_PER];

  auto load_tile = [&](int t) {
    const bool src1 = t >= nt0;
    const int k0 = (src1 ? t - nt0 : t) * BK;
    const int klen = src1 ? args.K1 : args.K0;
    const float* abase = src1 ? sg.a1 : sg.a0;
    const int lda = src1 ? args.lda1 : args.lda0;
    const int bcol = (src1 ? args.K0 : 0) + k0;
    const bool full = k0 + BK <= klen;  // block-uniform
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      const int gr = min(row0 + r, sg.rows - 1);
      const float* p = abase + (size_t)gr * lda + k0 + c4 * 4;
      if (full) {
        ra[i] = *reinterpret_cast<const f32x4_u*>(p);
      } else {
        const int kk = k0 + c4 * 4;
        f32x4 v;
        v[0] = kk + 0 < klen ? p[0] : 0.f;
        v[1] = kk + 1 < klen ? p[1] : 0.f;
        v[2] = kk + 2 < klen ? p[2] : 0.f;
        v[3] = kk + 3 < klen ? p[3] : 0.f;
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      const int gn = min(n0 + r, args.N - 1);
      const float* p = args.Bt + (size_t)gn * args.ldb + bcol + c4 * 4;
      if (full) {
        rbv[i] = *reinterpret_cast<const f32x4_u*>(p);
      } else {
        const int kk = k0 + c4 * 4;
        f32x4 v;
        v[0] = kk + 0 < klen ? p[0] : 0.f;
        v[1] = kk + 1 < klen ? p[1] : 0.f;
        v[2] = kk + 2 < klen ? p[2] : 0.f;
        v[3] = kk + 3 < klen ? p[3] : 0.f;
        rbv[i] = v;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&As[r * LS + c4 * 4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&Bs[r * LS + c4 * 4]) = rbv[i];
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * LS + 4 * lh];
  const float* b_frag = &Bs[lr * LS + 4 * lh];

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) load_tile(t + 1);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag + kg * 8);
      f32x4 b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS + kg * 8);
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
    }
    __syncthreads();
    if (t + 1 < ntiles) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 ---------------------
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + tn * 32 + lr;
    const bool n_ok = n < args.N;
    const float bias = n_ok ? args.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= sg.rows) continue;
      if (MODE == MODE_FEAT) {
        if (n_ok) sg.c[(size_t)row * args.ldc + n] = acc[tn][r] + bias;
      } else {
        if (n < args.ncols_out) {
          float v = 0.f;
          if (n_ok) {
            v = acc[tn][r] + bias;
            if (sg.add_pos) v += args.pos[(size_t)(row % sg.T) * args.N + n];
            v = sg.ids[row] != 0 ? v : 0.f;  // e * mask (carca.py:94); ids==0 rows become exact zeros
          }
          sg.c[(size_t)row * args.ldc + n] = v;
        }
      }
    }
  }
}

struct GatherArgs {
  const int32_t* ids[CARCA_MAX_SEGS];
  int row_start[CARCA_MAX_SEGS + 1];
  int nseg;
};

// zq[r, 0:d] = items_w[ids[r]] * sqrt(d)   (carca.py:87-88); one wave per row, lanes over columns
__global__ void gather_items_kernel(const GatherArgs ga, const float* __restrict__ items_w, int d, float sqrt_d,
                                    float* __restrict__ zq, int ldz, int total_rows) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < total_rows; row += nwaves) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < ga.nseg && row >= ga.row_start[i]) s = i;
    const int id = ga.ids[s][row - ga.row_start[s]];
    const float* src = items_w + (size_t)id * d;
    float* dst = zq + (size_t)row * ldz;
    for (int c = lane; c < d; c += 64) dst[c] = src[c] * sqrt_d;
  }
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
  constexpr int BM = 128, BN = 96, BK = 32;
  const int ldz = d + g;

  GatherArgs ga{};
  GemmArgs fa{}, ja{};
  int row_start = 0, rb = 0;
  for (int s = 0; s < nseg; ++s) {
    const CarcaRowSeg& sg = segs[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.T >= 1 && sg.ids && sg.attrs && sg.e_out && (n_ctx == 0 || sg.ctx),
                    "embed_fwd: segment %d has null pointers or no rows", s);
    CARCA_CHECK_ARG(!sg.add_pos || pos, "embed_fwd: segment %d wants pos but pos is NULL", s);
    ga.ids[s] = sg.ids;
    ga.row_start[s] = row_start;
    fa.seg[s] = SegDev{sg.attrs, sg.ctx, zq + (size_t)row_start * ldz + d, sg.ids, sg.rows, sg.T, 0, rb};
    ja.seg[s] = SegDev{zq + (size_t)row_start * ldz, nullptr, sg.e_out, sg.ids, sg.rows, sg.T, sg.add_pos, rb};
    row_start += sg.rows;
    rb += (sg.rows + BM - 1) / BM;
  }
  ga.row_start[nseg] = row_start;
  ga.nseg = nseg;
  const int total_rows = row_start, nrb = rb;

  if (stages & CARCA_EMBED_GATHER) {
    const int blocks = min((total_rows + 3) / 4, 2048);
    hipLaunchKernelGGL(gather_items_kernel, dim3(blocks), dim3(256), 0, stream, ga, items_w, d,
                       (float)sqrt((double)d), zq, ldz, total_rows);
    CARCA_LAUNCH_CHECK();
  }
  if (stages & CARCA_EMBED_FEAT) {
    fa.nseg = nseg; fa.nrb = nrb; fa.ncb = (g + BN - 1) / BN;
    fa.lda0 = n_attrs; fa.lda1 = n_ctx; fa.K0 = n_attrs; fa.K1 = n_ctx;
    fa.Bt = feats_w; fa.ldb = n_attrs + n_ctx; fa.N = g; fa.ldc = ldz; fa.ncols_out = g;
    fa.bias = feats_b; fa.pos = nullptr;
    const int grid = ((nrb + 7) / 8) * 8 * fa.ncb;
    hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, BK, MODE_FEAT>), dim3(grid), dim3((BM / 32) * 64), 0, stream, fa);
    CARCA_LAUNCH_CHECK();
  }
  if (stages & CARCA_EMBED_JOINT) {
    ja.nseg = nseg; ja.nrb = nrb; ja.ncb = (ld_e + BN - 1) / BN;
    ja.lda0 = ldz; ja.lda1 = 0; ja.K0 = ldz; ja.K1 = 0;
    ja.Bt = joint_w; ja.ldb = ldz; ja.N = d; ja.ldc = ld_e; ja.ncols_out = ld_e;
    ja.bias = joint_b; ja.pos = pos;
    const int grid = ((nrb + 7) / 8) * 8 * ja.ncb;
    hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, BK, MODE_JOINT>), dim3(grid), dim3((BM / 32) * 64), 0, stream, ja);
    CARCA_LAUNCH_CHECK();
  }
  return CARCA_OK;
}
