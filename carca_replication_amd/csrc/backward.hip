// Backward-pass kernels that are not plain GEMMs (those are carca_gemm_rows / carca_gemm_wgrad):
//   carca_layernorm_bwd     nn.LayerNorm backward (the autograd of carca.py:298,304,421)
//   carca_embed_scatter     gradient of nn.Embedding(padding_idx=0) * sqrt(d)  (carca.py:87-88)
//   carca_colsum            (weighted, masked, position-folded) column sums: dpos, d ffn.weight, ...
//   carca_sa_attn_bwd       attention core of MultiHeadAttention (carca.py:246-260) inside SelfAttentionBlock
//   carca_cross_attn_bwd    same inside CrossAttentionBlock + sigmoid/ffn head (carca.py:340-347)
// The reference has no hand-written backward: it is torch.autograd over those lines; the formulas
// below are their derivatives, checked against the reference's own parameter gradients (fixture G2).
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, two columns per lane (d <= 128)
//   xh = (x - mean) * rstd ; g = dy * gamma ; dx = rstd * (g - mean(g) - xh * mean(g * xh)) (+ addend)
//   dgamma += dy * xh ; dbeta += dy    (per-block partial sums, then one atomic per column)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int ld_dy,
                                                            const float* __restrict__ x, int ld_x,
                                                            const float* __restrict__ gamma, int rows, int d,
                                                            const float* __restrict__ addend, int ld_add,
                                                            float* __restrict__ dx, int ld_dx, int ncols_out,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  __shared__ float red[2][4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = lane, c1 = lane + 64;
  const bool ok0 = c0 < d, ok1 = c1 < d;
  const float g0 = ok0 ? gamma[c0] : 0.f, g1 = ok1 ? gamma[c1] : 0.f;
  const float inv_d = 1.0f / (float)d;
  float dg0 = 0.f, dg1 = 0.f, db0 = 0.f, db1 = 0.f;
  // RPW rows per wave and pass, all their loads issued before the first reduction: the four wave reductions of a row
  // are a ~1 us dependent chain, and rows are independent (one row per pass read 16 us at 6400 rows, four 6 us)
  constexpr int RPW = 4;
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += gridDim.x * 4 * RPW) {
    float xv0[RPW], xv1[RPW], yv0[RPW], yv1[RPW], av0[RPW], av1[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = min(row0 + i, rows - 1);
      const float* xr = x + (size_t)row * ld_x;
      const float* dr = dy + (size_t)row * ld_dy;
      xv0[i] = ok0 ? xr[c0] : 0.f;
      xv1[i] = ok1 ? xr[c1] : 0.f;
      yv0[i] = ok0 ? dr[c0] : 0.f;
      yv1[i] = ok1 ? dr[c1] : 0.f;
      av0[i] = av1[i] = 0.f;
      if (addend) {
        const float* ar = addend + (size_t)row * ld_add;
        av0[i] = ok0 ? ar[c0] : 0.f;
        av1[i] = ok1 ? ar[c1] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      if (row0 + i >= rows) break;
      const float x0 = xv0[i], x1 = xv1[i], y0 = yv0[i], y1 = yv1[i];
      const float mean = wave_sum(x0 + x1) * inv_d;
      const float e0 = ok0 ? x0 - mean : 0.f, e1 = ok1 ? x1 - mean : 0.f;
      const float rstd = 1.0f / sqrtf(wave_sum(e0 * e0 + e1 * e1) * inv_d + 1e-5f);
      const float h0 = e0 * rstd, h1 = e1 * rstd;
      const float a0 = y0 * g0, a1 = y1 * g1;
      const float m1 = wave_sum(a0 + a1) * inv_d;
      const float m2 = wave_sum(a0 * h0 + a1 * h1) * inv_d;
      const float o0 = rstd * (a0 - m1 - h0 * m2) + av0[i], o1 = rstd * (a1 - m1 - h1 * m2) + av1[i];
      dg0 += y0 * h0;
      dg1 += y1 * h1;
      db0 += y0;
      db1 += y1;
      float* outr = dx + (size_t)(row0 + i) * ld_dx;
      if (c0 < ncols_out) outr[c0] = ok0 ? o0 : 0.f;
      if (c1 < ncols_out) outr[c1] = ok1 ? o1 : 0.f;
    }
  }
  if (dgamma) {
    red[0][wave][c0] = dg0;
    red[0][wave][c1] = dg1;
    red[1][wave][c0] = db0;
    red[1][wave][c1] = db1;
    __syncthreads();
    if (tid < 128 && tid < d) {
      const float sg = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
      const float sb = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
      grad_add(&dgamma[tid], sg);
      grad_add(&dbeta[tid], sb);
    }
  }
}

// The same for rows that can be read 16 bytes per lane (padded internal buffers: every stride a multiple of 4): two rows
// per wave instruction -- lane (half = row of the pair, c4 = four contiguous columns) -- so that the four reductions of a
// row run over 32 lanes (five steps) on half the instructions, PAIRS row pairs in flight per wave, one block per CU.
template <int PAIRS>
__global__ __launch_bounds__(256) void layernorm_bwd_pairs_kernel(const float* __restrict__ dy, int ld_dy,
                                                                  const float* __restrict__ x, int ld_x,
                                                                  const float* __restrict__ gamma, int rows, int d,
                                                                  const float* __restrict__ addend, int ld_add,
                                                                  float* __restrict__ dx, int ld_dx, int ncols_out,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, c4 = lane & 31;
  const bool col_ok = 4 * c4 < ((d + 3) & ~3);
  const f32x4 g4 = col_ok ? gload4(gamma, 4 * c4) : zero4();  // (gamma is a [d] parameter: read below d only)
  f32x4 gm;
#pragma unroll
  for (int e = 0; e < 4; ++e) gm[e] = (4 * c4 + e < d) ? g4[e] : 0.f;
  const float inv_d = 1.0f / (float)d;
  f32x4 dg = zero4(), db = zero4();
  for (int p0 = (blockIdx.x * 4 + wave) * PAIRS; 2 * p0 < rows; p0 += gridDim.x * 4 * PAIRS) {
    f32x4 xv[PAIRS], yv[PAIRS], av[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int row = min(2 * (p0 + i) + half, rows - 1);
      const int c = col_ok ? 4 * c4 : 0;
      xv[i] = gload4(x, row * ld_x + c);
      yv[i] = gload4(dy, row * ld_dy + c);
      av[i] = addend ? gload4(addend, row * ld_add + c) : zero4();
    }
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
      const int row = 2 * (p0 + i) + half;
      f32x4 xr = xv[i], yr = yv[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = 4 * c4 + e < d && row < rows;
        xr[e] = ok ? xr[e] : 0.f;
        yr[e] = ok ? yr[e] : 0.f;
      }
      const float mean = half32_sum((xr[0] + xr[1]) + (xr[2] + xr[3])) * inv_d;
      f32x4 ev;
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[e] = (4 * c4 + e < d) ? xr[e] - mean : 0.f;
      const float rstd = 1.0f / sqrtf(half32_sum((ev[0] * ev[0] + ev[1] * ev[1]) + (ev[2] * ev[2] + ev[3] * ev[3])) * inv_d + 1e-5f);
      const f32x4 hv = ev * rstd, aa = yr * gm;
      const float m1 = half32_sum((aa[0] + aa[1]) + (aa[2] + aa[3])) * inv_d;
      const float m2 = half32_sum((aa[0] * hv[0] + aa[1] * hv[1]) + (aa[2] * hv[2] + aa[3] * hv[3])) * inv_d;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (4 * c4 + e < d) ? rstd * (aa[e] - m1 - hv[e] * m2) + av[i][e] : 0.f;
      if (row < rows) {
        dg = dg + yr * hv;
        db = db + yr;
        if (4 * c4 < ncols_out) *reinterpret_cast<f32x4*>(dx + (size_t)row * ld_dx + 4 * c4) = o;
      }
    }
  }
  if (dgamma) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the two rows of the pairs, then the four waves
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(dg[e]), __float_as_uint(dg[e]), false, false);
      dg[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
      r = __builtin_amdgcn_permlane32_swap(__float_as_uint(db[e]), __float_as_uint(db[e]), false, false);
      db[e] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if (half == 0) {
      *reinterpret_cast<f32x4*>(&red[0][wave][4 * c4]) = dg;
      *reinterpret_cast<f32x4*>(&red[1][wave][4 * c4]) = db;
    }
    __syncthreads();
    if (tid < 128 && tid < d) {
      grad_add(&dgamma[tid], red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid]);
      grad_add(&dbeta[tid], red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid]);
    }
  }
}

// d_items[ids[r]][c] += scale * dz[r][c] for ids[r] != 0 (padding_idx = 0 gets no gradient)
__global__ void embed_scatter_kernel(const float* __restrict__ dz, int ld_dz, const int32_t* __restrict__ ids,
                                     int rows, int d, float scale, float* __restrict__ d_items) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int row = wave; row < rows; row += nwaves) {
    const int id = ids[row];
    if (id == 0) continue;
    const float* src = dz + (size_t)row * ld_dz;
    float* dst = d_items + (size_t)id * d;
    for (int c = lane; c < d; c += 64) grad_add(&dst[c], scale * src[c]);
  }
}

// the same over several row segments ([profile | positives | negatives] of the train step) in one launch
struct ScatterSegs {
  const float* dz[CARCA_MAX_SEGS];
  const int32_t* ids[CARCA_MAX_SEGS];
  int row_end[CARCA_MAX_SEGS];  // running row totals
  int nseg;
};
__global__ void embed_scatter_segs_kernel(ScatterSegs S, int ld_dz, int d, float scale, float* __restrict__ d_items) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int rows = S.row_end[S.nseg - 1];
  for (int row = wave; row < rows; row += nwaves) {
    int s = 0;
    while (row >= S.row_end[s]) ++s;  // (wave-uniform: scalar loads of the kernel arguments)
    const int r = row - (s ? S.row_end[s - 1] : 0);
    const int id = S.ids[s][r];
    if (id == 0) continue;
    const float* src = S.dz[s] + (size_t)r * ld_dz;
    float* dst = d_items + (size_t)id * d;
    for (int c = lane; c < d; c += 64) grad_add(&dst[c], scale * src[c]);
  }
}

// LayerNorm backward for rows wider than 128 features (d <= 1024; the composed path of long_profile.py): one wave per row,
// a lane holds columns lane + 64 j; gamma / beta gradients are summed per wave over its rows and added once per column.
#define LNB_WIDE_MAX 16
__global__ __launch_bounds__(256) void layernorm_bwd_wide_kernel(const float* __restrict__ dy, int ld_dy,
                                                                 const float* __restrict__ x, int ld_x,
                                                                 const float* __restrict__ gamma, int rows, int d,
                                                                 const float* __restrict__ addend, int ld_add,
                                                                 float* __restrict__ dx, int ld_dx, int ncols_out,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)d;
  float dg[LNB_WIDE_MAX], db[LNB_WIDE_MAX], gm[LNB_WIDE_MAX];
#pragma unroll
  for (int j = 0; j < LNB_WIDE_MAX; ++j) {
    dg[j] = db[j] = 0.f;
    gm[j] = lane + 64 * j < d ? gamma[lane + 64 * j] : 0.f;
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ld_x;
    const float* dr = dy + (size_t)row * ld_dy;
    float xv[LNB_WIDE_MAX], yv[LNB_WIDE_MAX];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      xv[j] = c < d ? xr[c] : 0.f;
      yv[j] = c < d ? dr[c] : 0.f;
      s += xv[j];
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_WIDE_MAX; ++j) {
      xv[j] = lane + 64 * j < d ? xv[j] - mean : 0.f;
      q += xv[j] * xv[j];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_WIDE_MAX; ++j) {
      xv[j] *= rstd;  // x hat
      const float a = yv[j] * gm[j];
      s1 += a;
      s2 += a * xv[j];
    }
    const float m1 = wave_sum(s1) * inv_d, m2 = wave_sum(s2) * inv_d;
    float* outr = dx + (size_t)row * ld_dx;
    const float* ar = addend ? addend + (size_t)row * ld_add : nullptr;
#pragma unroll
    for (int j = 0; j < LNB_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      if (c < ncols_out) outr[c] = c < d ? rstd * (yv[j] * gm[j] - m1 - xv[j] * m2) + (ar ? ar[c] : 0.f) : 0.f;
      dg[j] += yv[j] * xv[j];
      db[j] += yv[j];
    }
  }
  if (dgamma) {
#pragma unroll
    for (int j = 0; j < LNB_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      if (c < d) {
        grad_add(&dgamma[c], dg[j]);
        grad_add(&dbeta[c], db[j]);
      }
    }
  }
}

// out[(row % T)][c] += sum over this block's rows of  w(row) * x[row][c],  w = rowscale * (ids != 0)
// grid.x blocks of 256 threads; thread = column (cols <= 256), rows strided by block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld_x, int rows, int cols,
                                                     const float* __restrict__ rowscale,
                                                     const int32_t* __restrict__ ids, int T,
                                                     float* __restrict__ out) {
  for (int c = threadIdx.x; c < cols; c += 256) {  // (one pass for cols <= 256: every row GEMM-sized caller; wider rows loop)
  if (T == 1) {
    float s = 0.f;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
      float w = rowscale ? rowscale[row] : 1.f;
      if (ids && ids[row] == 0) w = 0.f;
      s += w * x[(size_t)row * ld_x + c];
    }
    grad_add(&out[c], s);
  } else {
    // one block per position t (grid.x == T): rows t, t+T, t+2T, ...
    const int t = blockIdx.x;
    float s = 0.f;
    for (int row = t; row < rows; row += T) {
      float w = rowscale ? rowscale[row] : 1.f;
      if (ids && ids[row] == 0) w = 0.f;
      s += w * x[(size_t)row * ld_x + c];
    }
    grad_add(&out[(size_t)t * cols + c], s);
  }
  }
}

// in-place dropout of a [rows, cols] matrix + its keep-mask (CARCA.dropout on the profile embedding)
__global__ void dropout_fwd_kernel(float* __restrict__ x, int rows, int cols, int ld, const DropCfg dc_arg, unsigned site,
                                   uint8_t* __restrict__ mask) {
  const DropCfg dc = drop_resolve(dc_arg);
  const int total = rows * cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / cols, c = i - r * cols;
    const bool keep = drop_keep(dc, site, (unsigned)i);
    float* p = x + (size_t)r * ld + c;
    *p = keep ? *p * dc.scale : 0.f;
    if (mask) mask[i] = keep ? 1 : 0;
  }
}
// out = x * mask * scale (dropout backward); pad columns [cols, ncols_out) are zeroed
__global__ void mask_mul_kernel(const float* __restrict__ x, int ld_x, const uint8_t* __restrict__ mask, int ld_m,
                                float scale, float* __restrict__ out, int ld_out, int rows, int cols, int ncols_out) {
  const int total = rows * ncols_out;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / ncols_out, c = i - r * ncols_out;
    float v = 0.f;
    if (c < cols) v = mask[(size_t)r * ld_m + c] ? x[(size_t)r * ld_x + c] * scale : 0.f;
    out[(size_t)r * ld_out + c] = v;
  }
}

// ---------------------------------------------------------------------------------------------------
// attention backward
// ---------------------------------------------------------------------------------------------------
#define ATT_SP 72  // row stride of the [key][query] probability / dS images (64 + 8)

// scale + masked softmax of a lane's score registers (same arithmetic as attend_head)
__device__ __forceinline__ void masked_softmax(f32x4 (&p)[ATT_LT], unsigned okbits, float sqrt_dh) {
  const float inv_sqrt_dh = 1.0f / sqrt_dh;
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (okbits >> (4 * kt + r)) & 1u;
      const float sc = p[kt][r] * inv_sqrt_dh;
      p[kt][r] = sc;
      mx = ok ? fmaxf(mx, sc) : mx;
    }
  mx = quad4_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (okbits >> (4 * kt + r)) & 1u;
      const float e = ok ? __expf(p[kt][r] - mx) : 0.f;
      p[kt][r] = e;
      sum += e;
    }
  sum = quad4_sum(sum);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt) p[kt] = p[kt] * inv;
}

// Phase 1 of one (16-query tile, head): recompute P, form dS, emit dQ, park P^T and dS^T in LDS.
//   dofrag(ft) must return the lane's Bt fragment of dO: dO[q][hDHP + 16ft + 4mq + 0..3]
//   returns p (probabilities) so that the cross kernel can also form O.
template <int DHP, int SO, typename DoFrag>
__device__ __forceinline__ void attn_bwd_phase1(const float* Qs, const float* Ks, const float* V, int vstride,
                                                int L, int h, int qrow, int qcol, int nkt, unsigned okbits,
                                                float sqrt_dh, DoFrag dofrag, float* PT, float* DST,
                                                float* __restrict__ dq_row, f32x4 (&p)[ATT_LT], int lane,
                                                const uint8_t* __restrict__ mrow = nullptr, float dscale = 1.f) {
  constexpr int NFH = DHP / 16;
  const int ln = lane & 15, mq = lane >> 4;
  f32x4 qf[NFH], dof[NFH];
#pragma unroll
  for (int ft = 0; ft < NFH; ++ft) {
    qf[ft] = lds4(Qs + qrow * SO + h * DHP + 16 * ft + 4 * mq);
    dof[ft] = dofrag(ft);
  }
  f32x4 dp[ATT_LT];
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt) {
    p[kt] = zero4();
    dp[kt] = zero4();
    if (kt < nkt) {
      const float* krow = Ks + (16 * kt + ln) * SO + h * DHP + 4 * mq;
      // V: the user's rows staged in LDS like K (stride SO), or straight from HBM where LDS has no room (stride DPO)
      const float* vrow = V + (size_t)min(16 * kt + ln, L - 1) * vstride + h * DHP + 4 * mq;
      f32x4 s = zero4(), t = zero4();
#pragma unroll
      for (int ft = 0; ft < NFH; ++ft) {
        s = mfma16_group(lds4(krow + 16 * ft), qf[ft], s);    // S^T[key][q]
        t = mfma16_group(*reinterpret_cast<const f32x4*>(vrow + 16 * ft), dof[ft], t);   // dP^T[key][q] = V[key] . dO[q]
      }
      p[kt] = s;
      dp[kt] = t;
    }
  }
  masked_softmax(p, okbits, sqrt_dh);
  // attention-weight dropout: O = (P * M) V, so dP = (dO V^T) * M and dV uses P * M   (M = keep / (1 - p))
  if (mrow) {
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        const float m = (key < L && mrow[key]) ? dscale : 0.f;
        dp[kt][r] *= m;
      }
  }
  float dot = 0.f;
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) dot += p[kt][r] * dp[kt][r];
  dot = quad4_sum(dot);
  f32x4 ds[ATT_LT];
  const float inv_sqrt_dh = 1.0f / sqrt_dh;
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[kt][r] = p[kt][r] * (dp[kt][r] - dot) * inv_sqrt_dh;
  if (mrow) {  // from here on p means P * M (what multiplied V in the forward)
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        p[kt][r] *= (key < L && mrow[key]) ? dscale : 0.f;
      }
  }
  // dQ^T[f][q] = sum_key K[key][f] dS^T[key][q]   (A read transposed: one ds_read_b32 per step)
#pragma unroll
  for (int ft = 0; ft < NFH; ++ft) {
    f32x4 acc = zero4();
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc = mfma16(Ks[(16 * kt + 4 * mq + r) * SO + h * DHP + 16 * ft + ln], ds[kt][r], acc);
      }
    if (dq_row) *reinterpret_cast<f32x4*>(dq_row + h * DHP + 16 * ft + 4 * mq) = acc;
  }
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      PT[(16 * kt + 4 * mq + r) * ATT_SP + qcol] = p[kt][r];
      DST[(16 * kt + 4 * mq + r) * ATT_SP + qcol] = ds[kt][r];
    }
}

// Phase 2 tile: out[key][f] = sum_q  M[key][q] * R[q][f]   (M = DST -> dK with R = Q ; M = PT -> dV with R = dO)
//   rfrag(qt, s) must return R[16qt + 4mq + s][hDHP + 16ft + ln]  (the transposed Bt operand)
template <typename RFrag>
__device__ __forceinline__ f32x4 attn_bwd_phase2_tile(const float* M, int kt, int qt_begin, int qt_end, RFrag rfrag,
                                                      int lane) {
  const int ln = lane & 15, mq = lane >> 4;
  f32x4 acc = zero4();
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const f32x4 a = lds4(M + (16 * kt + ln) * ATT_SP + 16 * qt + 4 * mq);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a[s], rfrag(qt, s), acc);
  }
  return acc;
}

// ---- SelfAttentionBlock's attention core --------------------------------------------------------------
template <int DPI, int DHP, int NH>
constexpr bool sa_bwd_v_in_lds() {
  return sizeof(float) * (4 * ATT_LMAX * AttGeom<DPI, DHP, NH>::SO + 2 * ATT_LMAX * ATT_SP) <= 160 * 1024;
}

template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(512) void sa_attn_bwd_kernel(const float* __restrict__ qh, const float* __restrict__ kh,
                                                          const float* __restrict__ vh,
                                                          const float* __restrict__ d_attn /*[B*L, ld] plain*/,
                                                          int ld_da, const int32_t* __restrict__ ids,
                                                          float* __restrict__ dqh, float* __restrict__ dkh,
                                                          float* __restrict__ dvh, int L, int dh,
                                                          const uint8_t* __restrict__ m_attn, float dscale, int nparts) {
  using G = AttGeom<DPI, DHP, NH>;
  constexpr int SO = G::SO, DPO = G::DPO, NW = 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                  // [64][SO]
  float* Ks = Qs + ATT_LMAX * SO;   // [64][SO]
  float* Os = Ks + ATT_LMAX * SO;   // [64][SO]  dO
  constexpr bool V_LDS = sa_bwd_v_in_lds<DPI, DHP, NH>();  // (d = 128: four [64][136] images + P / dS exceed 160 KB)
  float* Vs = Os + ATT_LMAX * SO;   // [64][SO] when V_LDS
  float* PT = Vs + (V_LDS ? ATT_LMAX * SO : 0);  // [64][ATT_SP]
  float* DST = PT + ATT_LMAX * ATT_SP;

  // With fewer users than CUs the HEADS of a user are shared by two workgroups (their outputs are disjoint column
  // ranges of dQ / dK / dV: nothing passes between them, no atomics).
  const int u = blockIdx.x / nparts, part = blockIdx.x - u * nparts;
  const int h_per = (NH + nparts - 1) / nparts, h_lo = part * h_per, h_hi = min(NH, h_lo + h_per);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LT = (L + 15) >> 4;
  const int32_t* uid = ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);
  const size_t ubase = (size_t)u * L;

  // ---- load Q, K, dO rows (16 B per thread, coalesced) ----------------------------------------------
  constexpr int V4 = DPO / 4;
  // (every request goes out before the first is waited for: unconditional loads of clamped rows / columns -- under the
  // `r < L` branch each of the three loop iterations was a round trip of its own)
  constexpr int ST_IT = (ATT_LMAX * V4 + 511) / 512;
  {
    f32x4 qq[ST_IT], kk[ST_IT], oo[ST_IT], vv[ST_IT];
#pragma unroll
    for (int j = 0; j < ST_IT; ++j) {
      const int i = tid + 512 * j, r = min(i / V4, L - 1), c4 = i % V4;
      const size_t off = (ubase + r) * DPO + 4 * c4;
      qq[j] = glb4(qh + off);
      kk[j] = glb4(kh + off);
      vv[j] = V_LDS ? glb4(vh + off) : zero4();
      const float* dar = d_attn + (ubase + r) * ld_da;  // plain feature order -> head-padded order
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int jf = unpad_feature(4 * c4 + e, dh, DHP);
        const float dv = dar[jf >= 0 ? jf : 0];
        oo[j][e] = jf >= 0 ? dv : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < ST_IT; ++j) {
      const int i = tid + 512 * j, r = i / V4, c4 = i - r * V4;
      if (i < 16 * LT * V4) {
        const bool live = r < L;
        *reinterpret_cast<f32x4*>(Qs + r * SO + 4 * c4) = live ? qq[j] : zero4();
        *reinterpret_cast<f32x4*>(Ks + r * SO + 4 * c4) = live ? kk[j] : zero4();
        *reinterpret_cast<f32x4*>(Os + r * SO + 4 * c4) = live ? oo[j] : zero4();
        if (V_LDS) *reinterpret_cast<f32x4*>(Vs + r * SO + 4 * c4) = live ? vv[j] : zero4();
      }
    }
  }
  __syncthreads();

  const float sqrt_dh = sqrtf((float)dh);
  const int ln = lane & 15, mq = lane >> 4;
  const float* vsrc = V_LDS ? Vs : vh + ubase * DPO;
  const int vstride = V_LDS ? SO : DPO;
#pragma unroll 1
  for (int h = h_lo; h < h_hi; ++h) {
    for (int qt = wave; qt < LT; qt += NW) {
      const int q = 16 * qt + ln;
      const bool q_ok = (pmask >> q) & 1ull;
      unsigned okbits = 0;
#pragma unroll
      for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * kt + 4 * mq + r;
          okbits |= ((q_ok && key <= q && ((pmask >> key) & 1ull)) ? 1u : 0u) << (4 * kt + r);
        }
      f32x4 p[ATT_LT];
      auto dofrag = [&](int ft) { return lds4(Os + q * SO + h * DHP + 16 * ft + 4 * mq); };
      float* dq_row = q < L ? dqh + (ubase + q) * DPO : nullptr;
      const uint8_t* mrow = m_attn ? m_attn + (((size_t)u * NH + h) * L + (q < L ? q : 0)) * L : nullptr;
      attn_bwd_phase1<DHP, SO>(Qs, Ks, vsrc, vstride, L, h, q, q, qt + 1, okbits, sqrt_dh, dofrag, PT, DST, dq_row, p, lane,
                               mrow, dscale);
    }
    __syncthreads();
    // dK / dV tiles: job = (which, kt, ft); queries that can see key tile kt are tiles qt >= kt
    const int njobs = 2 * LT * G::NFH;
    for (int job = wave; job < njobs; job += NW) {
      const int which = job / (LT * G::NFH);
      const int jj = job - which * LT * G::NFH;
      const int kt = jj / G::NFH, ft = jj - kt * G::NFH;
      const float* R = which == 0 ? Qs : Os;
      auto rfrag = [&](int qt, int s) { return R[(16 * qt + 4 * mq + s) * SO + h * DHP + 16 * ft + ln]; };
      const f32x4 acc = attn_bwd_phase2_tile(which == 0 ? DST : PT, kt, kt, LT, rfrag, lane);
      float* out = which == 0 ? dkh : dvh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        if (key < L) out[(ubase + key) * DPO + h * DHP + 16 * ft + ln] = acc[r];
      }
    }
    __syncthreads();
  }
}

// ---- CrossAttentionBlock's attention core + scoring head ------------------------------------------------
struct CrossBwdGroups {
  CarcaCrossBwdGroup g[CARCA_MAX_GROUPS];
  int n;
};

template <int DPI, int DHP, int NH>
constexpr int cross_bwd_head_slots() {
  using G = AttGeom<DPI, DHP, NH>;
  return sizeof(float) * (3 * ATT_LMAX * G::SO + 4 * ATT_LMAX * ATT_SP + 64 + (1 + 8) * G::DPO + 64) <= 160 * 1024 ? 2 : 1;
}

template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(512) void cross_attn_bwd_kernel(const float* __restrict__ kh,
                                                             const float* __restrict__ vh,
                                                             const int32_t* __restrict__ p_ids,
                                                             const CrossBwdGroups groups,
                                                             const float* __restrict__ ffn_w_pad,
                                                             float* __restrict__ dkh, float* __restrict__ dvh,
                                                             float* __restrict__ d_ffn_w_pad, int L, int dh,
                                                             int training, float dscale, int nparts) {
  using G = AttGeom<DPI, DHP, NH>;
  constexpr int SO = G::SO, DPO = G::DPO, NW = 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Qs = lds;                       // [64][SO] one chunk of <= 64 targets
  float* Ks = Qs + ATT_LMAX * SO;        // [64][SO]
  float* Vs = Ks + ATT_LMAX * SO;        // [64][SO]
  // Two heads of a workgroup in flight where LDS has room for a second P^T / dS^T pair (d <= 96): phase 1 of a chunk has
  // QT <= 4 (tile, head) units per head -- with one head at a time half of the eight waves sat it out, and the workgroups
  // that own two of the three heads set the kernel's length.
  constexpr int HS = cross_bwd_head_slots<DPI, DHP, NH>();
  constexpr int PSZ = 2 * ATT_LMAX * ATT_SP;  // one slot: P^T then dS^T
  float* PT0 = Vs + ATT_LMAX * SO;       // [HS] x { [64][ATT_SP] P^T, [64][ATT_SP] dS^T }
  float* dls = PT0 + HS * PSZ;           // [64] dlogit of the chunk
  float* wps = dls + 64;                 // [DPO] ffn weight, head-padded
  float* dwp = wps + DPO;                // [NW][DPO] its gradient, accumulated over the user's targets: one slot per
                                         // wave (the target tiles are dealt to the waves statically), summed in wave
                                         // order at the end -- an LDS atomic would add in the order the waves arrive
  int* ids_s = reinterpret_cast<int*>(dwp + NW * DPO);  // [64] target ids of the chunk

  // heads shared by two workgroups when users <= CUs / 2 (see sa_attn_bwd_kernel); dlogit is written by the first
  const int u = blockIdx.x / nparts, part = blockIdx.x - u * nparts;
  const int h_per = (NH + nparts - 1) / nparts, h_lo = part * h_per, h_hi = min(NH, h_lo + h_per);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LT = (L + 15) >> 4;
  const int32_t* uid = p_ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);
  const size_t ubase = (size_t)u * L;
  constexpr int V4 = DPO / 4;

  // (every request of a staging pass goes out before the first is waited for: unconditional loads of clamped rows -- a load
  // under a branch is waited for right there, i.e. one round trip per loop iteration)
  constexpr int ST_IT = (ATT_LMAX * V4 + 511) / 512;  // 16-byte slots per thread of a [64][DPO] image
  {
    f32x4 kk[ST_IT], vv[ST_IT];
#pragma unroll
    for (int j = 0; j < ST_IT; ++j) {
      const int i = tid + 512 * j, r = min(i / V4, L - 1), c4 = i % V4;
      kk[j] = glb4(kh + (ubase + r) * DPO + 4 * c4);
      vv[j] = glb4(vh + (ubase + r) * DPO + 4 * c4);
    }
#pragma unroll
    for (int j = 0; j < ST_IT; ++j) {
      const int i = tid + 512 * j, r = i / V4, c4 = i - r * V4;
      if (i < 16 * LT * V4) {
        *reinterpret_cast<f32x4*>(Ks + r * SO + 4 * c4) = r < L ? kk[j] : zero4();
        *reinterpret_cast<f32x4*>(Vs + r * SO + 4 * c4) = r < L ? vv[j] : zero4();
      }
    }
  }
  for (int i = tid; i < DPO; i += 512) wps[i] = ffn_w_pad[i];
  for (int i = tid; i < NW * DPO; i += 512) dwp[i] = 0.f;

  const float sqrt_dh = sqrtf((float)dh);
  const int ln = lane & 15, mq = lane >> 4;
  bool first_pass = true;
  for (int gi = 0; gi < groups.n; ++gi) {
    const CarcaCrossBwdGroup grp = groups.g[gi];
    for (int n0 = 0; n0 < grp.N; n0 += 64) {
      const int nq = min(64, grp.N - n0), QT = (nq + 15) >> 4;
      const size_t gbase = (size_t)u * grp.N + n0;
      const size_t ybase = (size_t)u * (grp.ld_y ? grp.ld_y : grp.N) + n0;  // y and dy may be column blocks of [B, sum N]
      __syncthreads();  // previous chunk's phase 2 is done with Qs / dls
      {
        f32x4 qq[ST_IT];
        const int t64 = min(tid & 63, nq - 1);
        const float yv = grp.y[ybase + t64], dyv = grp.dy[ybase + t64];
        const int idv = grp.ids[gbase + t64];
#pragma unroll
        for (int j = 0; j < ST_IT; ++j) {
          const int i = tid + 512 * j, r = min(i / V4, nq - 1), c4 = i % V4;
          qq[j] = glb4(grp.qh + (gbase + r) * DPO + 4 * c4);
        }
#pragma unroll
        for (int j = 0; j < ST_IT; ++j) {
          const int i = tid + 512 * j, r = i / V4, c4 = i - r * V4;
          if (i < 16 * QT * V4) *reinterpret_cast<f32x4*>(Qs + r * SO + 4 * c4) = r < nq ? qq[j] : zero4();
        }
        if (tid < 64) {
          const float dl = tid < nq ? dyv * yv * (1.0f - yv) : 0.f;  // d sigmoid
          if (tid < nq && grp.dlogit && part == 0) grp.dlogit[gbase + tid] = dl;
          dls[tid] = dl;
        } else if (tid < 128) {
          ids_s[tid - 64] = tid - 64 < nq ? idv : 0;
        }
      }
      __syncthreads();
#pragma unroll 1
      for (int h0 = h_lo; h0 < h_hi; h0 += HS) {
        const int nh = min(HS, h_hi - h0);
        for (int unit = wave; unit < nh * QT; unit += NW) {
          const int hs = unit / QT, qt = unit - hs * QT, h = h0 + hs;
          float* PT = PT0 + hs * PSZ;
          float* DST = PT + ATT_LMAX * ATT_SP;
          const int qloc = 16 * qt + ln;      // row in the chunk
          const int nslot = n0 + qloc;        // target slot in the group
          const bool in_range = qloc < nq;
          const bool q_ok = in_range && ids_s[qloc] != 0;
          unsigned okbits = 0;
#pragma unroll
          for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = 16 * kt + 4 * mq + r;
              const bool ok = q_ok && ((pmask >> key) & 1ull) && (!training || key < nslot);
              okbits |= (ok ? 1u : 0u) << (4 * kt + r);
            }
          const int nkt = training ? min(LT, (n0 >> 4) + qt + 1) : LT;
          const float dl = dls[qloc];
          f32x4 p[ATT_LT];
          auto dofrag = [&](int ft) { return lds4(wps + h * DHP + 16 * ft + 4 * mq) * dl; };  // dO = dl (x) w_pad
          float* dq_row = in_range ? grp.dqh + (gbase + qloc) * DPO : nullptr;
          const uint8_t* mrow =
              grp.m_attn ? grp.m_attn + (((size_t)u * NH + h) * grp.N + (in_range ? nslot : 0)) * L : nullptr;
          attn_bwd_phase1<DHP, SO>(Qs, Ks, Vs, SO, L, h, qloc, qloc, nkt, okbits, sqrt_dh, dofrag, PT, DST,
                                   dq_row, p, lane, mrow, dscale);
          // d w_pad[f] += sum_q dl[q] * O[q][f],  O^T[f][q] = sum_key V[key][f] P^T[key][q]
#pragma unroll
          for (int ft = 0; ft < G::NFH; ++ft) {
            f32x4 acc = zero4();
#pragma unroll
            for (int kt = 0; kt < ATT_LT; ++kt)
              if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  acc = mfma16(Vs[(16 * kt + 4 * mq + r) * SO + h * DHP + 16 * ft + ln], p[kt][r], acc);
              }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = row16_sum(acc[r] * dl);
              if (ln == 0) dwp[wave * DPO + h * DHP + 16 * ft + 4 * mq + r] += v;  // (this wave's own slot)
            }
          }
        }
        __syncthreads();
        const int per_head = 2 * LT * G::NFH, njobs = nh * per_head;
        for (int job = wave; job < njobs; job += NW) {
          const int hs = job / per_head, j1 = job - hs * per_head, h = h0 + hs;
          const float* PT = PT0 + hs * PSZ;
          const float* DST = PT + ATT_LMAX * ATT_SP;
          const int which = j1 / (LT * G::NFH);
          const int jj = j1 - which * LT * G::NFH;
          const int kt = jj / G::NFH, ft = jj - kt * G::NFH;
          f32x4 acc;
          // the same lane owns an output element in every chunk: plain read-modify-write, no race -- the old values are
          // requested BEFORE the tile's MFMAs (clamped rows, unconditional) and added behind them
          float* out = which == 0 ? dkh : dvh;
          float* dst0 = out + ubase * DPO + h * DHP + 16 * ft + ln;
          float oldv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) oldv[r] = dst0[(size_t)min(16 * kt + 4 * mq + r, L - 1) * DPO];
          if (which == 0) {
            auto rfrag = [&](int qt, int s) { return Qs[(16 * qt + 4 * mq + s) * SO + h * DHP + 16 * ft + ln]; };
            acc = attn_bwd_phase2_tile(DST, kt, 0, QT, rfrag, lane);
          } else {
            const float wv = wps[h * DHP + 16 * ft + ln];
            auto rfrag = [&](int qt, int s) { return dls[16 * qt + 4 * mq + s] * wv; };
            acc = attn_bwd_phase2_tile(PT, kt, 0, QT, rfrag, lane);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * mq + r;
            if (key < L) dst0[(size_t)key * DPO] = first_pass ? acc[r] : oldv[r] + acc[r];
          }
        }
        __syncthreads();
      }
      first_pass = false;
    }
  }
  for (int i = tid; i < DPO; i += 512) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += dwp[w * DPO + i];
    grad_add(&d_ffn_w_pad[i], s);
  }
}

template <int DPI, int DHP, int NH>
int launch_sa_attn_bwd(const float* qh, const float* kh, const float* vh, const float* d_attn, int ld_da,
                       const int32_t* ids, float* dqh, float* dkh, float* dvh, int B, int L, int d,
                       const uint8_t* m_attn, float dscale, hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  const size_t lds_bytes = sizeof(float) * ((sa_bwd_v_in_lds<DPI, DHP, NH>() ? 4 : 3) * ATT_LMAX * G::SO + 2 * ATT_LMAX * ATT_SP);
  auto kern = sa_attn_bwd_kernel<DPI, DHP, NH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("sa_attn_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int nparts = (NH >= 2 && carca_tuning(CARCA_TUNE_ATTN_VARIANT) != 1 && 2 * B <= carca_num_cus()) ? 2 : 1;
  hipLaunchKernelGGL(kern, dim3(B * nparts), dim3(512), lds_bytes, stream, qh, kh, vh, d_attn, ld_da, ids, dqh, dkh, dvh, L,
                     d / NH, m_attn, dscale, nparts);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

template <int DPI, int DHP, int NH>
int launch_cross_attn_bwd(const float* kh, const float* vh, const int32_t* p_ids, const CrossBwdGroups& groups,
                          const float* ffn_w_pad, float* dkh, float* dvh, float* d_ffn_w_pad, int B, int L, int d,
                          int training, float dscale, hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  const size_t lds_bytes = sizeof(float) * (3 * ATT_LMAX * G::SO + 2 * cross_bwd_head_slots<DPI, DHP, NH>() * ATT_LMAX * ATT_SP + 64 +
                                            (1 + 8) * G::DPO + 64);
  auto kern = cross_attn_bwd_kernel<DPI, DHP, NH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("cross_attn_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int nparts = (NH >= 2 && carca_tuning(CARCA_TUNE_ATTN_VARIANT) != 1 && 2 * B <= carca_num_cus()) ? 2 : 1;
  hipLaunchKernelGGL(kern, dim3(B * nparts), dim3(512), lds_bytes, stream, kh, vh, p_ids, groups, ffn_w_pad, dkh, dvh,
                     d_ffn_w_pad, L, d / NH, training, dscale, nparts);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

extern "C" int carca_layernorm_bwd(const float* dy, int ld_dy, const float* x, int ld_x, const float* gamma, int rows,
                                   int d, const float* addend, int ld_add, float* dx, int ld_dx, int ncols_out,
                                   float* dgamma, float* dbeta, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(dy && x && gamma && dx && rows >= 1 && d >= 1, "layernorm_bwd: null pointer or bad dims");
  CARCA_CHECK_SUPPORTED(d <= 64 * LNB_WIDE_MAX && ncols_out <= 64 * LNB_WIDE_MAX, "layernorm_bwd: d=%d / ncols_out=%d > %d", d,
                        ncols_out, 64 * LNB_WIDE_MAX);
  CARCA_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma and dbeta go together");
  CARCA_CHECK_ARG(ncols_out <= ld_dx && ld_dy >= d && ld_x >= d && (!addend || ld_add >= d), "layernorm_bwd: bad strides");
  if (d > 128 || ncols_out > 128) {  // (rows beyond the fused kernels' width: long_profile.py's composed path)
    const int blocks = min((rows + 15) / 16, 256);
    hipLaunchKernelGGL(layernorm_bwd_wide_kernel, dim3(blocks), dim3(256), 0, stream, dy, ld_dy, x, ld_x, gamma, rows, d,
                       addend, ld_add, dx, ld_dx, ncols_out, dgamma, dbeta);
    CARCA_LAUNCH_CHECK();
    return CARCA_OK;
  }
  // padded internal buffers (every stride a multiple of 4 floats, width <= the strides): the row-pair kernel, one block
  // per CU, four pairs in flight per wave
  const bool vec = ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dx % 4 == 0 && (!addend || ld_add % 4 == 0) &&
                   ((d + 3) & ~3) <= ld_dy && ((d + 3) & ~3) <= ld_x && (!addend || ((d + 3) & ~3) <= ld_add) &&
                   ncols_out % 4 == 0 && (size_t)rows * (size_t)max(max(ld_dy, ld_x), max(ld_dx, ld_add)) < (1u << 29) &&
                   carca_tuning(6) != 1;
  if (vec) {
    const int nb = min((rows + 31) / 32, carca_num_cus());
    hipLaunchKernelGGL(layernorm_bwd_pairs_kernel<4>, dim3(nb), dim3(256), 0, stream, dy, ld_dy, x, ld_x, gamma, rows, d,
                       addend, ld_add, dx, ld_dx, ncols_out, dgamma, dbeta);
    CARCA_LAUNCH_CHECK();
    return CARCA_OK;
  }
  // few blocks: every block ends with 2d atomics on the SAME dgamma / dbeta addresses, which serialise in L2
  // (6400 rows: 400 blocks 18.7 us, 200 blocks one row per pass 16 us)
  const int blocks = min((rows + 63) / 64, 128);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, ld_dy, x, ld_x, gamma, rows, d,
                     addend, ld_add, dx, ld_dx, ncols_out, dgamma, dbeta);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_embed_scatter(const float* dz, int ld_dz, const int32_t* ids, int rows, int d, float scale,
                                   float* d_items, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(dz && ids && d_items && rows >= 1 && d >= 1 && ld_dz >= d, "embed_scatter: bad arguments");
  const int blocks = min((rows + 3) / 4, 2048);
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(blocks), dim3(256), 0, stream, dz, ld_dz, ids, rows, d, scale, d_items);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

int carca_embed_scatter_segs(const float* const* dz, int ld_dz, const int32_t* const* ids, const int* rows, int nseg,
                             int d, float scale, float* d_items, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(dz && ids && rows && d_items && nseg >= 1 && nseg <= CARCA_MAX_SEGS && d >= 1 && ld_dz >= d,
                  "embed_scatter: bad arguments");
  ScatterSegs S = {};
  int total = 0;
  for (int s = 0; s < nseg; ++s) {
    CARCA_CHECK_ARG(dz[s] && ids[s] && rows[s] >= 1, "embed_scatter: segment %d malformed", s);
    S.dz[s] = dz[s]; S.ids[s] = ids[s]; S.row_end[s] = (total += rows[s]);
  }
  S.nseg = nseg;
  const int blocks = min((total + 3) / 4, 2048);
  hipLaunchKernelGGL(embed_scatter_segs_kernel, dim3(blocks), dim3(256), 0, stream, S, ld_dz, d, scale, d_items);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_dropout_fwd(float* x, int rows, int cols, int ld, const CarcaDropout* drop, uint8_t* mask,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(x && drop && rows >= 1 && cols >= 1 && ld >= cols && drop->p >= 0.f && drop->p < 1.f,
                  "dropout_fwd: bad arguments");
  const DropCfg dc = make_drop(drop);
  if (!dc.thresh) return CARCA_OK;
  const int blocks = min((rows * cols + 255) / 256, 2048);
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, rows, cols, ld, dc, drop->site, mask);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_mask_mul(const float* x, int ld_x, const uint8_t* mask, int ld_m, float scale, float* out,
                              int ld_out, int rows, int cols, int ncols_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(x && mask && out && rows >= 1 && cols >= 1 && ld_x >= cols && ld_m >= cols && ncols_out >= cols &&
                      ld_out >= ncols_out,
                  "mask_mul: bad arguments");
  const int blocks = min((rows * ncols_out + 255) / 256, 2048);
  hipLaunchKernelGGL(mask_mul_kernel, dim3(blocks), dim3(256), 0, stream, x, ld_x, mask, ld_m, scale, out, ld_out, rows,
                     cols, ncols_out);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_colsum(const float* x, int ld_x, int rows, int cols, const float* rowscale, const int32_t* ids,
                            int T, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(x && out && rows >= 1 && cols >= 1 && ld_x >= cols && T >= 1, "colsum: bad arguments");
  const int blocks = T == 1 ? min(rows, 512) : T;
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, stream, x, ld_x, rows, cols, rowscale, ids, T, out);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_sa_attn_bwd(const float* qh, const float* kh, const float* vh, const float* d_attn, int ld_da,
                                 const int32_t* ids, float* dqh, float* dkh, float* dvh, int B, int L, int d, int H,
                                 const uint8_t* m_attn, float drop_scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(qh && kh && vh && d_attn && ids && dqh && dkh && dvh && ld_da >= d, "sa_attn_bwd: bad arguments");
  CARCA_CHECK_ARG(B >= 1 && L >= 1 && d >= 1 && H >= 1 && d % H == 0, "sa_attn_bwd: bad dims");
  CARCA_CHECK_SUPPORTED(L <= CARCA_MAX_L, "sa_attn_bwd: L=%d > %d", L, CARCA_MAX_L);
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK) return CARCA_ERR_UNSUPPORTED;
  CARCA_ATT_DISPATCH(launch_sa_attn_bwd, qh, kh, vh, d_attn, ld_da, ids, dqh, dkh, dvh, B, L, d, m_attn, drop_scale,
                     stream);
  carca_set_error("sa_attn_bwd: no kernel built for d=%d H=%d", d, H);
  return CARCA_ERR_UNSUPPORTED;
}

extern "C" int carca_cross_attn_bwd(const float* kh, const float* vh, const int32_t* p_ids,
                                    const CarcaCrossBwdGroup* groups, int ngroups, const float* ffn_w_pad, float* dkh,
                                    float* dvh, float* d_ffn_w_pad, int B, int L, int d, int H, int training,
                                    float drop_scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(kh && vh && p_ids && groups && ffn_w_pad && dkh && dvh && d_ffn_w_pad, "cross_attn_bwd: null pointer");
  CARCA_CHECK_ARG(ngroups >= 1 && ngroups <= CARCA_MAX_GROUPS && B >= 1 && L >= 1 && d >= 1 && H >= 1 && d % H == 0,
                  "cross_attn_bwd: bad dims");
  CARCA_CHECK_SUPPORTED(L <= CARCA_MAX_L, "cross_attn_bwd: L=%d > %d", L, CARCA_MAX_L);
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK) return CARCA_ERR_UNSUPPORTED;
  CrossBwdGroups gd{};
  for (int i = 0; i < ngroups; ++i) {
    const CarcaCrossBwdGroup& g = groups[i];
    CARCA_CHECK_ARG(g.qh && g.y && g.dy && g.ids && g.dqh && g.N >= 1, "cross_attn_bwd: group %d malformed", i);
    gd.g[i] = g;
  }
  gd.n = ngroups;
  CARCA_ATT_DISPATCH(launch_cross_attn_bwd, kh, vh, p_ids, gd, ffn_w_pad, dkh, dvh, d_ffn_w_pad, B, L, d, training,
                     drop_scale, stream);
  carca_set_error("cross_attn_bwd: no kernel built for d=%d H=%d", d, H);
  return CARCA_ERR_UNSUPPORTED;
}
