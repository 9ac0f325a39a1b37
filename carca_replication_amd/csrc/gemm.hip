// Dense building blocks on v_mfma_f32_32x32x2_f32 (exact fp32):
//
//   carca_gemm_rows   C[m][n]  = sum_k A[m][k] * Bt[n][k] (+ epilogue)     rows = users x slots
//   carca_gemm_wgrad  dW[n][k] += sum_r dY[r][n] * X[r][k]                  contraction over rows
//
// gemm_rows is the kernel behind AllEmbedding's feats_embed / joint_embed (carca.py:86,89; 97% of
// the model's flops at n_attrs = 4096) and every input-gradient product of the backward pass;
// gemm_wgrad produces every weight gradient (the feats_embed one is as large as the forward GEMM).
//
// gemm_rows:  block tile 128 x 96, K step 32, 4 waves, wave w owns rows 32w..32w+31 x all 96
//   columns (3 accumulator tiles).  Operands are staged global -> registers -> LDS with rows padded
//   to 36 floats so that the 16-byte fragment reads are bank-conflict free; the next K tile's global
//   loads are in flight while the current one is multiplied.  144 registers per lane -> 3 blocks per
//   CU, which is what C2's 755 blocks want (256 CUs x 3 = 768 slots: one resident round).
//   Block ids that share an A row block are 8 apart (same XCD under round-robin placement) so the
//   A tile is fetched from HBM once and re-read from that XCD's L2 by the other column blocks
//   (a speed choice only, never correctness).
// gemm_wgrad: block tile 96 (n) x 128 (k), 32 rows per step; both operands are read from their
//   row-major LDS tiles TRANSPOSED (lane = n resp. k, one ds_read_b32 per MFMA operand), so neither
//   dY nor X is ever transposed in memory.  Row splits combine through fp32 atomics.
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

struct GemmDev {
  CarcaGemmDesc d;
  int rb_start[CARCA_MAX_SEGS + 1];
  int nrb, ncb;
};

template <int BM, int BN, int BK, int PF = 1>
__global__ __launch_bounds__((BM / 32) * 64) void gemm_rows_kernel(const GemmDev args) {
  constexpr int NW = BM / 32, NT = NW * 64, LS = BK + 4, TN = BN / 32;
  constexpr int C4 = BK / 4;  // float4 slots per tile row
  constexpr int A_SLOTS = BM * C4, B_SLOTS = BN * C4;
  constexpr int A_PER = (A_SLOTS + NT - 1) / NT, B_PER = (B_SLOTS + NT - 1) / NT;
  static_assert(BK % 8 == 0 && BN % 32 == 0 && BM % 32 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float As[BM * LS];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LS];

  const CarcaGemmDesc& D = args.d;
  // ---- block id -> (row block, col block); same row block => same id mod 8 (same XCD) ----------
  const int id = blockIdx.x;
  const int per = 8 * args.ncb;
  const int grp = id / per, j = id - grp * per;
  const int cb = j >> 3, rb = grp * 8 + (j & 7);
  if (rb >= args.nrb) return;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = (rb - args.rb_start[s]) * BM;  // first row of this block inside the segment
  const int n0 = cb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = (D.K0 + BK - 1) / BK, nt1 = (D.K1 + BK - 1) / BK;
  const int ntiles = nt0 + nt1;

  f32x4 ra[PF][A_PER], rbv[PF][B_PER];  // PF tiles of global loads in flight (register ring, compile-time indexed)
  // element offsets of this thread's A rows inside each k-source (loop invariant; handles [B, T, K] views)
  size_t aoff0[A_PER], aoff1[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int slot = tid + i * NT;
    const int gr = min(row0 + slot / C4, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    aoff0[i] = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
               : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                               : (size_t)gr * D.lda0;
    aoff1[i] = sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
  }

  auto load4 = [](const float* p, bool full, int kk, int klen) -> f32x4 {
    if (full) return *reinterpret_cast<const f32x4_u*>(p);
    f32x4 v;
    v[0] = kk + 0 < klen ? p[0] : 0.f;
    v[1] = kk + 1 < klen ? p[1] : 0.f;
    v[2] = kk + 2 < klen ? p[2] : 0.f;
    v[3] = kk + 3 < klen ? p[3] : 0.f;
    return v;
  };
  auto load_tile = [&](int t, f32x4 (&ra_)[A_PER], f32x4 (&rb_)[B_PER]) {
    const bool src1 = t >= nt0;
    const int k0 = (src1 ? t - nt0 : t) * BK;
    const int klen = src1 ? D.K1 : D.K0;
    const float* abase = src1 ? sg.a1 : sg.a0;
    const float* bbase = src1 ? D.bt1 : D.bt0;
    const int ldb = src1 ? D.ldb1 : D.ldb0;
    const bool full = k0 + BK <= klen;  // block-uniform
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int c4 = slot % C4;
      ra_[i] = load4(abase + (src1 ? aoff1[i] : aoff0[i]) + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      const int gn = min(n0 + r, D.N - 1);
      rb_[i] = load4(bbase + (size_t)gn * ldb + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
  };
  auto store_tile = [&](const f32x4 (&ra_)[A_PER], const f32x4 (&rb_)[B_PER]) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&As[r * LS + c4 * 4]) = ra_[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&Bs[r * LS + c4 * 4]) = rb_[i];
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

  auto compute_tile = [&]() {
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
  };
  if constexpr (PF == 1) {
    load_tile(0, ra[0], rbv[0]);
    store_tile(ra[0], rbv[0]);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      if (t + 1 < ntiles) load_tile(t + 1, ra[0], rbv[0]);
      compute_tile();
      __syncthreads();
      if (t + 1 < ntiles) {
        store_tile(ra[0], rbv[0]);
        __syncthreads();
      }
    }
  } else {
    // short K loops at low occupancy (narrow outputs): the global latency of a tile is longer than its MFMAs,
    // so keep PF tiles of loads in flight in a register ring; the ring slot is a compile-time index.
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (u < ntiles) load_tile(u, ra[u], rbv[u]);
    for (int t0 = 0; t0 < ntiles; t0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int t = t0 + u;
        if (t < ntiles) {
          store_tile(ra[u], rbv[u]);
          __syncthreads();
          if (t + PF < ntiles) load_tile(t + PF, ra[u], rbv[u]);
          compute_tile();
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 ---------------------
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + tn * 32 + lr;
    if (n >= D.ncols_out) continue;
    const bool n_ok = n < D.N;
    const float bias = (n_ok && D.bias) ? D.bias[n] : 0.f;
    const float cv = (n_ok && D.colvec) ? D.colvec[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row >= sg.rows) continue;
      float v = 0.f;
      if (n_ok) {
        v = (D.alpha != 0.f ? D.alpha * acc[tn][r] : acc[tn][r]) + bias;
        if (sg.add_pos) v += D.pos[(size_t)(row % sg.T) * D.N + n];
        if (sg.add) v += sg.add[(size_t)row * D.ld_add + n];
        if (sg.rowscale) v += sg.rowscale[row] * cv;
        if (sg.gate) {
          const float gv = sg.gate[(size_t)row * D.ld_gate + n];
          const float gs = D.gate_scale != 0.f ? D.gate_scale : 1.0f;
          v *= gv > 0.f ? gs : ((gv < 0.f || !D.gate_zero_drops) ? D.gate_slope * gs : 0.f);
        }
        if (D.mask_rows) v = sg.ids[row] != 0 ? v : 0.f;  // e * mask (carca.py:94): exact zeros
      }
      sg.c[(size_t)row * D.ldc + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
struct WgradDev {
  CarcaWgradDesc d;
  int chunk_start[CARCA_MAX_SEGS + 1];  // 32-row chunks per segment, prefix sums
  int nnb, nkb, nkb0, nsplit, chunks_per_split;
};

template <int BNO, int BKO, int BR>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradDev args) {
  static_assert(BNO == 96 && BKO == 128 && BR == 32, "tile shape baked into the lane maps below");
  constexpr int NT = 256;
  __shared__ __attribute__((aligned(16))) float Ys[BR * BNO];  // dY tile [row][n]
  __shared__ __attribute__((aligned(16))) float Xs[BR * BKO];  // X  tile [row][k]

  const CarcaWgradDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b = blockIdx.x;
  const int kb = b % args.nkb;
  b /= args.nkb;
  const int nb = b % args.nnb;
  const int split = b / args.nnb;
  const bool src1 = kb >= args.nkb0;  // this block's dW columns come from the second X source
  const int n0 = nb * BNO, k0 = (src1 ? kb - args.nkb0 : kb) * BKO;
  const int klen = src1 ? D.K1 : D.K;
  const int ldx = src1 ? D.ld_x1 : D.ld_x;
  const int c_begin = split * args.chunks_per_split;
  const int c_end = min(c_begin + args.chunks_per_split, args.chunk_start[D.nseg]);
  const bool n_full = n0 + BNO <= D.N, k_full = k0 + BKO <= klen;

  constexpr int Y4 = BNO / 4, X4 = BKO / 4;                            // float4 per tile row
  constexpr int Y_PER = BR * Y4 / NT, X_PER = BR * X4 / NT;            // 3, 4
  f32x4 ry[Y_PER], rx[X_PER];

  auto load_chunk = [&](int c) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && c >= args.chunk_start[i]) s = i;
    const CarcaWgradSeg sg = D.seg[s];
    const int r0 = (c - args.chunk_start[s]) * BR;
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int slot = tid + i * NT;
      const int r = slot / Y4, c4 = slot - r * Y4;
      const int row = r0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      bool ok = row < sg.rows;
      if (ok && D.mask_rows) ok = sg.ids[row] != 0;
      if (ok) {
        const float* p = sg.dy + (size_t)row * D.ld_dy + n0 + c4 * 4;
        if (n_full) {
          v = *reinterpret_cast<const f32x4_u*>(p);
        } else {
          const int nn = n0 + c4 * 4;
          v[0] = nn + 0 < D.N ? p[0] : 0.f;
          v[1] = nn + 1 < D.N ? p[1] : 0.f;
          v[2] = nn + 2 < D.N ? p[2] : 0.f;
          v[3] = nn + 3 < D.N ? p[3] : 0.f;
        }
      }
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
      const int slot = tid + i * NT;
      const int r = slot / X4, c4 = slot - r * X4;
      const int row = r0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < sg.rows) {
        const int64_t bs = src1 ? sg.x1_bstride : sg.x_bstride;
        const size_t roff = (!src1 && sg.x_gather) ? (size_t)sg.ids[row] * ldx
                            : bs                    ? (size_t)(row / sg.T) * bs + (size_t)(row % sg.T) * ldx
                                                    : (size_t)row * ldx;
        const float* p = (src1 ? sg.x1 : sg.x) + roff + k0 + c4 * 4;
        if (k_full) {
          v = *reinterpret_cast<const f32x4_u*>(p);
        } else {
          const int kk = k0 + c4 * 4;
          v[0] = kk + 0 < klen ? p[0] : 0.f;
          v[1] = kk + 1 < klen ? p[1] : 0.f;
          v[2] = kk + 2 < klen ? p[2] : 0.f;
          v[3] = kk + 3 < klen ? p[3] : 0.f;
        }
      }
      rx[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int slot = tid + i * NT;
      *reinterpret_cast<f32x4*>(&Ys[slot * 4]) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
      const int slot = tid + i * NT;
      *reinterpret_cast<f32x4*>(&Xs[slot * 4]) = rx[i];
    }
  };

  // wave w owns k columns 32w..32w+31 of the block's 128 and all three 32-wide n tiles
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // thread t < 96 of a kb == 0 block sums column n0 + t of dY

  const int lr = lane & 31, lh = lane >> 5;
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    if (c + 1 < c_end) load_chunk(c + 1);
    // D[m = n index][n = k index] = sum_r Ys[r][m] * Xs[r][n]:  A lane (i, kk) = Ys[2s + kk][i]
#pragma unroll 4
    for (int st = 0; st < BR / 2; ++st) {
      const int r = 2 * st + lh;
      const float xb = Xs[r * BKO + wave * 32 + lr];
      const float y0 = Ys[r * BNO + lr], y1 = Ys[r * BNO + 32 + lr], y2 = Ys[r * BNO + 64 + lr];
      acc[0] = mfma32(y0, xb, acc[0]);
      acc[1] = mfma32(y1, xb, acc[1]);
      acc[2] = mfma32(y2, xb, acc[2]);
    }
    if (D.db && kb == 0 && tid < BNO) {
#pragma unroll 8
      for (int r = 0; r < BR; ++r) bsum += Ys[r * BNO + tid];
    }
    __syncthreads();
    if (c + 1 < c_end) {
      store_chunk();
      __syncthreads();
    }
  }

  // D row (= n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (= k) = lane&31
  const int k = k0 + wave * 32 + lr;
  if (k < klen) {
    const int kcol = (src1 ? D.K : 0) + k;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < D.N) atomicAdd(&D.dw[(size_t)n * D.ldw + kcol], acc[t][r]);
      }
  }
  if (D.db && kb == 0 && tid < BNO && n0 + tid < D.N) atomicAdd(&D.db[n0 + tid], bsum);
}

}  // namespace

template <int BM, int BN, int BK, int PF>
static int launch_gemm_rows(const CarcaGemmDesc* desc, hipStream_t stream) {
  GemmDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (desc->seg[s].rows + BM - 1) / BM;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = (desc->ncols_out + BN - 1) / BN;
  const int grid = ((rb + 7) / 8) * 8 * g.ncb;
  hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, BK, PF>), dim3(grid), dim3((BM / 32) * 64), 0, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_gemm_rows(const CarcaGemmDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(desc && desc->nseg >= 1 && desc->nseg <= CARCA_MAX_SEGS, "gemm_rows: bad segment count");
  CARCA_CHECK_ARG(desc->bt0 && desc->K0 >= 1 && desc->N >= 1 && desc->ldc >= desc->N && desc->lda0 >= desc->K0 &&
                      desc->ldb0 >= desc->K0,
                  "gemm_rows: bad k-source 0 / output geometry");
  CARCA_CHECK_ARG(desc->K1 == 0 || (desc->bt1 && desc->lda1 >= desc->K1 && desc->ldb1 >= desc->K1),
                  "gemm_rows: bad k-source 1");
  CARCA_CHECK_ARG(desc->ncols_out >= desc->N && desc->ncols_out <= desc->ldc, "gemm_rows: ncols_out outside [N, ldc]");
  int rb128 = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.a0 && sg.c && (desc->K1 == 0 || sg.a1), "gemm_rows: segment %d malformed", s);
    CARCA_CHECK_ARG(sg.T >= 1 || (!sg.a0_bstride && !sg.a1_bstride && !sg.add_pos), "gemm_rows: segment %d needs T >= 1",
                    s);
    CARCA_CHECK_ARG(!(sg.add_pos && (!desc->pos || sg.T < 1 || !sg.ids)) && !(desc->mask_rows && !sg.ids) &&
                        !(sg.rowscale && !desc->colvec) && !(sg.a0_gather && !sg.ids),
                    "gemm_rows: segment %d epilogue needs a pointer that is NULL", s);
    rb128 += (sg.rows + 127) / 128;
  }
  // Narrow outputs (the joint embedding, every d-wide product of the backward pass) give too few 128 x 96 blocks
  // to fill 256 CUs and leave one long MFMA chain per wave: 32-column blocks triple the wave count instead
  // (the A tile is re-read from L2 by the three column blocks of a row block, which share an XCD).
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  const bool narrow = variant != 1 && rb128 * ((desc->ncols_out + 95) / 96) < 384;
  if (narrow) return launch_gemm_rows<128, 32, 32, 4>(desc, stream);
  return launch_gemm_rows<128, 96, 32, 1>(desc, stream);
}

extern "C" int carca_gemm_wgrad(const CarcaWgradDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(desc && desc->nseg >= 1 && desc->nseg <= CARCA_MAX_SEGS, "gemm_wgrad: bad segment count");
  CARCA_CHECK_ARG(desc->dw && desc->N >= 1 && desc->K >= 1 && desc->K1 >= 0 && desc->ldw >= desc->K + desc->K1 &&
                      desc->ld_dy >= desc->N && desc->ld_x >= desc->K && (desc->K1 == 0 || desc->ld_x1 >= desc->K1),
                  "gemm_wgrad: bad geometry");
  constexpr int BNO = 96, BKO = 128, BR = 32;
  WgradDev g{};
  g.d = *desc;
  int chunks = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaWgradSeg& sg = desc->seg[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.dy && sg.x && !(desc->mask_rows && !sg.ids) && (desc->K1 == 0 || sg.x1),
                    "gemm_wgrad: segment %d malformed", s);
    CARCA_CHECK_ARG(sg.T >= 1 || (!sg.x_bstride && !sg.x1_bstride), "gemm_wgrad: segment %d needs T >= 1", s);
    CARCA_CHECK_ARG(!(sg.x_gather && !sg.ids), "gemm_wgrad: segment %d gathers without ids", s);
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.chunk_start[s] = chunks;
    chunks += (sg.rows + BR - 1) / BR;
  }
  g.chunk_start[desc->nseg] = chunks;
  g.nnb = (desc->N + BNO - 1) / BNO;
  g.nkb0 = (desc->K + BKO - 1) / BKO;
  g.nkb = g.nkb0 + (desc->K1 + BKO - 1) / BKO;
  // row splits: fill the chip's 4 x 256 resident slots in ONE round (119 registers -> 4 blocks per CU;
  // measured at C2: 512 slots 1017 us, 768 972, 1024 809, 1536 865), but keep >= 4 chunks (128 rows) per split
  const int tiles = g.nnb * g.nkb;
  const int slots = carca_tuning(CARCA_TUNE_WGRAD_SLOTS) > 0 ? carca_tuning(CARCA_TUNE_WGRAD_SLOTS) : 1024;
  int nsplit = tiles >= slots ? 1 : slots / tiles;
  nsplit = max(1, min(nsplit, (chunks + 3) / 4));
  g.chunks_per_split = (chunks + nsplit - 1) / nsplit;
  g.nsplit = (chunks + g.chunks_per_split - 1) / g.chunks_per_split;
  hipLaunchKernelGGL((gemm_wgrad_kernel<BNO, BKO, BR>), dim3(tiles * g.nsplit), dim3(256), 0, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}
