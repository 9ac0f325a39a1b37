// The reference's ablation decoders as row kernels on the same padded layouts:
//   DotProduct.forward          carca.py:352-367   y = sigmoid(p . o)        train: slot-wise, eval: last profile slot
//   WeightedDotProduct.forward  carca.py:370-399   p'[t] = p[t] sum_{j<=t} gamma^j; optional L2 normalisation;
//                                                  y = sigmoid(p' . o) or (p' . o + 1) / 2
// plus the stand-alone final LayerNorm (carca.py:421) that the cross-attention kernel otherwise fuses.
// One wave per row, two features per lane (d <= 128): HBM/latency-bound row work, nothing to tile.
// Also here, because it is the same row-dot shape: the reference's KNN baseline model (knn.py:8-21),
//   y[b][t] = attrs(last profile slot of b) . attrs(target t)   over n_attrs features, no link, no parameters.
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                            int ldy, int rows, int d, const float* __restrict__ w,
                                                            const float* __restrict__ b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ldx;
    float v0 = lane < d ? xr[lane] : 0.f, v1 = lane + 64 < d ? xr[lane + 64] : 0.f;
    row_layernorm(v0, v1, lane, d, w, b);
    float* yr = y + (size_t)row * ldy;
    if (lane < ldy) yr[lane] = v0;
    if (lane + 64 < ldy) yr[lane + 64] = v1;
  }
}

// The same LayerNorm for rows WIDER than 128 features (d <= 1024: the composed path of long_profile.py takes any d the
// reference takes -- its only constraint is d % H == 0, carca.py:208; the fused per-user kernels stop at 128).  One wave per
// row, a lane holds columns lane + 64 j.
#define LN_WIDE_MAX 16  // 64-column groups per row
__global__ __launch_bounds__(256) void layernorm_fwd_wide_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                                 int ldy, int rows, int d, const float* __restrict__ w,
                                                                 const float* __restrict__ b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)d;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ldx;
    float v[LN_WIDE_MAX];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < d ? xr[c] : 0.f;
      s += v[j];
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < d ? v[j] - mean : 0.f;
      q += v[j] * v[j];
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
    float* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int j = 0; j < LN_WIDE_MAX; ++j) {
      const int c = lane + 64 * j;
      if (c < ldy) yr[c] = c < d ? v[j] * rstd * w[c] + b[c] : 0.f;
    }
  }
}

// y[b][t] = link(p_row . o[b][t]); p_row = p[b][t] (slotwise, T == L) or p[b][L-1]
__global__ __launch_bounds__(256) void dot_score_fwd_kernel(const float* __restrict__ p, int ldp,
                                                            const float* __restrict__ o, int ldo,
                                                            float* __restrict__ y, int B, int L, int T, int d,
                                                            int slotwise, int link) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < B * T; row += gridDim.x * 4) {
    const int b = row / T, t = row - b * T;
    const float* pr = p + (size_t)(b * L + (slotwise ? t : L - 1)) * ldp;
    const float* orow = o + (size_t)row * ldo;
    float s = 0.f;
    if (lane < d) s += pr[lane] * orow[lane];
    if (lane + 64 < d) s += pr[lane + 64] * orow[lane + 64];
    s = wave_sum(s);
    if (lane == 0) y[row] = link == 0 ? 1.0f / (1.0f + expf(-s)) : (s + 1.0f) * 0.5f;
  }
}

// dl = dy * dlink/ds; d_o[row] = dl * p_row; d_p[p_row] += dl * o[row]
__global__ __launch_bounds__(256) void dot_score_bwd_kernel(const float* __restrict__ p, int ldp,
                                                            const float* __restrict__ o, int ldo,
                                                            const float* __restrict__ y, const float* __restrict__ dy,
                                                            float* __restrict__ dp, int ld_dp, float* __restrict__ d_o,
                                                            int ld_do, int B, int L, int T, int d, int slotwise,
                                                            int link) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < B * T; row += gridDim.x * 4) {
    const int b = row / T, t = row - b * T;
    const size_t prow = (size_t)(b * L + (slotwise ? t : L - 1));
    const float yy = y[row];
    const float dl = dy[row] * (link == 0 ? yy * (1.0f - yy) : 0.5f);
    const float* pr = p + prow * ldp;
    const float* orow = o + (size_t)row * ldo;
    float* dor = d_o + (size_t)row * ld_do;
    float* dpr = dp + prow * ld_dp;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h;
      if (c < ld_do) dor[c] = c < d ? dl * pr[c] : 0.f;
      if (c < d) {
        if (slotwise)
          dpr[c] += dl * orow[c];  // one writer per element and launch; groups are separate, stream-ordered launches
        else
          atomicAdd(&dpr[c], dl * orow[c]);
      }
    }
  }
}

// out[b][t][:] = c_t x[b][t][:], c_t = sum_{j<=t} gamma^j: what WeightedDotProduct's repeat / tril / sum amounts to
// (carca.py:376-378,385-386: the history is repeated along a NEW axis, so slots are scaled, not mixed).  Self-adjoint.
__global__ void slot_decay_scale_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo, int rows,
                                        int L, int d, float gamma) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ldo) return;
  const int row = i / ldo, f = i - row * ldo;
  const int t = row % L;
  float c = 0.f;
  for (int j = 0; j <= t; ++j) c += powf(gamma, (float)j);
  out[i] = f < d ? c * x[(size_t)row * ldx + f] : 0.f;
}

// torch.nn.functional.normalize(x, dim=-1): y = x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                         int ldy, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ldx;
    const float v0 = lane < d ? xr[lane] : 0.f, v1 = lane + 64 < d ? xr[lane + 64] : 0.f;
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(v0 * v0 + v1 * v1)), 1e-12f);
    float* yr = y + (size_t)row * ldy;
    if (lane < ldy) yr[lane] = v0 * inv;
    if (lane + 64 < ldy) yr[lane + 64] = v1 * inv;
  }
}
// dx = (dy - y (y . dy)) / max(||x||, 1e-12), y = normalize(x)   (rows with ||x|| < 1e-12: dx = dy / 1e-12 like autograd)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ dy, int ld_dy,
                                                         float* __restrict__ dx, int ld_dx, int rows, int d) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float* xr = x + (size_t)row * ldx;
    const float* gr = dy + (size_t)row * ld_dy;
    const float v0 = lane < d ? xr[lane] : 0.f, v1 = lane + 64 < d ? xr[lane + 64] : 0.f;
    const float g0 = lane < d ? gr[lane] : 0.f, g1 = lane + 64 < d ? gr[lane + 64] : 0.f;
    const float nrm = sqrtf(wave_sum(v0 * v0 + v1 * v1));
    const bool tiny = nrm < 1e-12f;
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);
    const float y0 = v0 * inv, y1 = v1 * inv;
    const float dot = tiny ? 0.f : wave_sum(y0 * g0 + y1 * g1);
    float* outr = dx + (size_t)row * ld_dx;
    if (lane < ld_dx) outr[lane] = lane < d ? (g0 - y0 * dot) * inv : 0.f;
    if (lane + 64 < ld_dx) outr[lane + 64] = lane + 64 < d ? (g1 - y1 * dot) * inv : 0.f;
  }
}

// KNN.forward (knn.py:13-19): one wave per target row; the user's last profile row is re-read by its T targets out of
// L1/L2 (16 KB at n_attrs = 4096), the target rows stream from HBM exactly once: HBM-bound, B*T*F*4 bytes.
// table_rows > 0: rows are gathered from the attribute table by id (ids outside the table read as zero rows).
template <bool VEC>
__global__ __launch_bounds__(256) void knn_score_kernel(const float* __restrict__ p_a, long p_bstride,
                                                        const float* __restrict__ o_a, long o_bstride,
                                                        const int32_t* __restrict__ p_x, const int32_t* __restrict__ o_x,
                                                        int table_rows, float* __restrict__ y, int B, int L, int T, int F) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * 4 + wave; row < B * T; row += gridDim.x * 4) {
    const int b = row / T;
    const float *pr, *orow;
    bool live = true;
    if (table_rows > 0) {
      const int pi = p_x[(size_t)b * L + L - 1], oi = o_x[row];
      live = pi >= 0 && pi < table_rows && oi >= 0 && oi < table_rows;
      pr = p_a + (size_t)(live ? pi : 0) * F;
      orow = p_a + (size_t)(live ? oi : 0) * F;
    } else {
      pr = p_a + (size_t)b * p_bstride + (size_t)(L - 1) * F;
      orow = o_a + (size_t)b * o_bstride + (size_t)(row - b * T) * F;
    }
    float s = 0.f;
    if (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(pr);
      const float4* o4 = reinterpret_cast<const float4*>(orow);
      const int n4 = F >> 2;
      int i = lane;
      for (; i + 192 < n4; i += 256) {  // 4 independent 16-byte loads per operand in flight
        const float4 a0 = o4[i], a1 = o4[i + 64], a2 = o4[i + 128], a3 = o4[i + 192];
        const float4 c0 = p4[i], c1 = p4[i + 64], c2 = p4[i + 128], c3 = p4[i + 192];
        s += a0.x * c0.x + a0.y * c0.y + a0.z * c0.z + a0.w * c0.w;
        s += a1.x * c1.x + a1.y * c1.y + a1.z * c1.z + a1.w * c1.w;
        s += a2.x * c2.x + a2.y * c2.y + a2.z * c2.z + a2.w * c2.w;
        s += a3.x * c3.x + a3.y * c3.y + a3.z * c3.z + a3.w * c3.w;
      }
      for (; i < n4; i += 64) {
        const float4 a0 = o4[i], c0 = p4[i];
        s += a0.x * c0.x + a0.y * c0.y + a0.z * c0.z + a0.w * c0.w;
      }
    } else {
      for (int i = lane; i < F; i += 64) s += pr[i] * orow[i];
    }
    s = wave_sum(s);
    if (lane == 0) y[row] = live ? s : 0.f;
  }
}

inline int row_blocks(int rows) { return min((rows + 3) / 4, 4096); }

__global__ void add_positions_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ pos,
                                     float* __restrict__ out, int ldo, long rows, int T, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  const long r = i / d;
  const int c = (int)(i - r * d);
  out[r * ldo + c] = x[r * ldx + c] + pos[(r % T) * d + c];
}

// One wave per (user b, head h, query t): lanes walk the keys in strides of 64; scores are kept in registers for up to
// MHA_KPL x 64 keys and recomputed beyond (never needed at the reference's sizes).  Same arithmetic order as
// carca.py:251-259: additive mask before the scale, softmax, re-mask.
#define MHA_KPL 16
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                       const float* __restrict__ v, int ldk,
                                                       const int32_t* __restrict__ q_ids, const int32_t* __restrict__ k_ids,
                                                       int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                                                       float* __restrict__ out, int ldo, float* __restrict__ w_out,
                                                       DropCfg drop, unsigned site, unsigned char* __restrict__ m_out) {
  drop = drop_resolve(drop);
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= (long)B * H * Tq) return;
  const int t = (int)(wid % Tq);
  const int h = (int)((wid / Tq) % H);
  const int b = (int)(wid / ((long)Tq * H));
  const int dh = d / H;
  const float* qr = q + ((long)b * Tq + t) * ldq + h * dh;
  const bool q_ok = q_ids[(long)b * Tq + t] != 0;
  const float sqrt_dh = sqrtf((float)dh);
  float sc[MHA_KPL];
  bool ok[MHA_KPL];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i) {
    const int j = lane + 64 * i;
    sc[i] = -3.0e38f;
    ok[i] = false;
    if (j < Tk) {
      const float* kr = k + ((long)b * Tk + j) * ldk + h * dh;
      float dot = 0.f;
      for (int c = 0; c < dh; ++c) dot += qr[c] * kr[c];
      ok[i] = q_ok && k_ids[(long)b * Tk + j] != 0 && (!has_causal || j - t <= causal);
      sc[i] = ((ok[i] ? 0.0f : -4294967296.0f) + dot) / sqrt_dh;
      mx = fmaxf(mx, sc[i]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i)
    if (lane + 64 * i < Tk) {
      sc[i] = expf(sc[i] - mx);
      sum += sc[i];
    }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i) {
    const int j = lane + 64 * i;
    if (j < Tk) {
      sc[i] = ok[i] ? sc[i] / sum : 0.f;  // softmax, then "* attn_mask" (carca.py:256)
      if (w_out) w_out[(((long)h * B + b) * Tq + t) * Tk + j] = sc[i];  // (before dropout, carca.py:262-263)
      if (drop.thresh) {  // nn.Dropout on the weights (carca.py:258): element (b, h, t, j) of the site
        const long e = (((long)b * H + h) * Tq + t) * Tk + j;
        const bool keep = drop_keep(drop, site, (unsigned)e);
        if (m_out) m_out[e] = keep ? 1 : 0;
        sc[i] = keep ? sc[i] * drop.scale : 0.f;
      }
    }
  }
  // out[c] = sum_j W[j] v[j][c]: a lane owns the head's columns lane and lane + 64; the weights travel by shuffle, which
  // every lane executes (the key loop is uniform)
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i)
    for (int l = 0; l < 64 && l + 64 * i < Tk; ++l) {
      const float wj = __shfl(sc[i], l);
      const float* vr = v + ((long)b * Tk + l + 64 * i) * ldk + h * dh;
      if (lane < dh) acc0 += wj * vr[lane];
      if (lane + 64 < dh) acc1 += wj * vr[lane + 64];
    }
  float* orow = out + ((long)b * Tq + t) * ldo + h * dh;
  if (lane < dh) orow[lane] = acc0;
  if (lane + 64 < dh) orow[lane + 64] = acc1;
}

// Backward of mha_core_kernel, one wave per (user, head, query) like the forward: the weights are recomputed, then
//   dW_j = dO . v_j (+ the caller's gradient of the returned weights),  dS_j = W_j (dW_j - sum_k W_k dW_k) / sqrt(dh)
// (masked entries: W = 0 and no gradient, carca.py:256),  dq = sum_j dS_j k_j  (this wave's row: plain store),
// dk_j += dS_j q,  dv_j += W_j dO  (rows shared by the queries of a user: accumulated -- grad_add).
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                           const float* __restrict__ v, int ldk,
                                                           const int32_t* __restrict__ q_ids, const int32_t* __restrict__ k_ids,
                                                           int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                                                           const float* __restrict__ d_out, int ldo,
                                                           const float* __restrict__ d_w, float* __restrict__ dq,
                                                           float* __restrict__ dk, float* __restrict__ dv,
                                                           const unsigned char* __restrict__ keep, float keep_scale) {
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= (long)B * H * Tq) return;
  const int t = (int)(wid % Tq);
  const int h = (int)((wid / Tq) % H);
  const int b = (int)(wid / ((long)Tq * H));
  const int dh = d / H;
  const float* qr = q + ((long)b * Tq + t) * ldq + h * dh;
  const float* dor = d_out ? d_out + ((long)b * Tq + t) * ldo + h * dh : nullptr;
  const bool q_ok = q_ids[(long)b * Tq + t] != 0;
  const float sqrt_dh = sqrtf((float)dh);
  float sc[MHA_KPL], dwj[MHA_KPL], mj[MHA_KPL];  // mj: the dropout multiplier of weight j (keep / (1 - p), 1 without dropout)
  bool ok[MHA_KPL];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i) {
    const int j = lane + 64 * i;
    sc[i] = -3.0e38f;
    dwj[i] = 0.f;
    mj[i] = 1.f;
    ok[i] = false;
    if (j < Tk) {
      const float* kr = k + ((long)b * Tk + j) * ldk + h * dh;
      const float* vr = v + ((long)b * Tk + j) * ldk + h * dh;
      float dot = 0.f, dd = 0.f;
      for (int c = 0; c < dh; ++c) {
        dot += qr[c] * kr[c];
        if (dor) dd += dor[c] * vr[c];
      }
      ok[i] = q_ok && k_ids[(long)b * Tk + j] != 0 && (!has_causal || j - t <= causal);
      sc[i] = ((ok[i] ? 0.0f : -4294967296.0f) + dot) / sqrt_dh;
      mx = fmaxf(mx, sc[i]);
      if (keep) {
        mj[i] = keep[(((long)b * H + h) * Tq + t) * Tk + j] ? keep_scale : 0.f;
        dd *= mj[i];  // (out = (W * m) v: d out / d W_j carries the multiplier; the returned weights are pre-dropout)
      }
      if (d_w) dd += d_w[(((long)h * B + b) * Tq + t) * Tk + j];
      dwj[i] = dd;
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i)
    if (lane + 64 * i < Tk) {
      sc[i] = expf(sc[i] - mx);
      sum += sc[i];
    }
  sum = wave_sum(sum);
  float wd = 0.f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i) {
    sc[i] = (lane + 64 * i < Tk && ok[i]) ? sc[i] / sum : 0.f;
    wd += sc[i] * dwj[i];
  }
  wd = wave_sum(wd);
  float ds[MHA_KPL];
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i) ds[i] = sc[i] * (dwj[i] - wd) / sqrt_dh;
  // dq[c] = sum_j dS_j k_j[c]: a lane owns the head's columns lane and lane + 64, the dS travel by shuffle
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < MHA_KPL; ++i)
    for (int l = 0; l < 64 && l + 64 * i < Tk; ++l) {
      const float dsj = __shfl(ds[i], l), wj = __shfl(sc[i] * mj[i], l);
      if (dsj == 0.f && wj == 0.f) continue;  // (uniform: the shuffled values are the same in every lane)
      const long krow = ((long)b * Tk + l + 64 * i) * ldk + h * dh;
      if (lane < dh) {
        a0 += dsj * k[krow + lane];
        grad_add(&dk[krow + lane], dsj * qr[lane]);
        if (dor) grad_add(&dv[krow + lane], wj * dor[lane]);
      }
      if (lane + 64 < dh) {
        a1 += dsj * k[krow + lane + 64];
        grad_add(&dk[krow + lane + 64], dsj * qr[lane + 64]);
        if (dor) grad_add(&dv[krow + lane + 64], wj * dor[lane + 64]);
      }
    }
  float* dqr = dq + ((long)b * Tq + t) * ldq + h * dh;
  if (lane < dh) dqr[lane] = a0;
  if (lane + 64 < dh) dqr[lane + 64] = a1;
}

}  // namespace

extern "C" int carca_layernorm_fwd(const float* x, int ldx, float* y, int ldy, int rows, int d, const float* w,
                                   const float* b, void* stream_) {
  CARCA_CHECK_ARG(x && y && w && b && rows >= 1 && d >= 1 && ldx >= d && ldy >= d, "layernorm_fwd: bad arguments");
  CARCA_CHECK_SUPPORTED(d <= 64 * LN_WIDE_MAX && ldy <= 64 * LN_WIDE_MAX, "layernorm_fwd: d=%d / ldy=%d > %d", d, ldy, 64 * LN_WIDE_MAX);
  if (d > 128 || ldy > 128)
    hipLaunchKernelGGL(layernorm_fwd_wide_kernel, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream_, x, ldx, y, ldy,
                       rows, d, w, b);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream_, x, ldx, y, ldy,
                       rows, d, w, b);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_dot_score_fwd(const float* p, int ldp, const float* o, int ldo, float* y, int B, int L, int T, int d,
                                   int slotwise, int link, void* stream_) {
  CARCA_CHECK_ARG(p && o && y && B >= 1 && L >= 1 && T >= 1 && d >= 1 && ldp >= d && ldo >= d, "dot_score_fwd: bad arguments");
  CARCA_CHECK_ARG(!slotwise || T == L, "dot_score_fwd: slot-wise scoring needs T == L (got T=%d, L=%d)", T, L);
  CARCA_CHECK_SUPPORTED(d <= 128, "dot_score_fwd: d=%d > 128", d);
  hipLaunchKernelGGL(dot_score_fwd_kernel, dim3(row_blocks(B * T)), dim3(256), 0, (hipStream_t)stream_, p, ldp, o, ldo, y,
                     B, L, T, d, slotwise, link);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_dot_score_bwd(const float* p, int ldp, const float* o, int ldo, const float* y, const float* dy,
                                   float* dp, int ld_dp, float* d_o, int ld_do, int B, int L, int T, int d, int slotwise,
                                   int link, void* stream_) {
  CARCA_CHECK_ARG(p && o && y && dy && dp && d_o && B >= 1 && L >= 1 && T >= 1 && d >= 1 && ldp >= d && ldo >= d &&
                      ld_dp >= d && ld_do >= d,
                  "dot_score_bwd: bad arguments");
  CARCA_CHECK_ARG(!slotwise || T == L, "dot_score_bwd: slot-wise scoring needs T == L");
  CARCA_CHECK_SUPPORTED(d <= 128 && ld_do <= 128, "dot_score_bwd: d=%d / ld_do=%d > 128", d, ld_do);
  hipLaunchKernelGGL(dot_score_bwd_kernel, dim3(row_blocks(B * T)), dim3(256), 0, (hipStream_t)stream_, p, ldp, o, ldo, y,
                     dy, dp, ld_dp, d_o, ld_do, B, L, T, d, slotwise, link);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_knn_score(const float* p_a, int64_t p_bstride, const float* o_a, int64_t o_bstride,
                               const int32_t* p_x, const int32_t* o_x, int table_rows, float* y, int B, int L, int T,
                               int F, void* stream_) {
  CARCA_CHECK_ARG(p_a && y && B >= 1 && L >= 1 && T >= 1 && F >= 1 && table_rows >= 0, "knn_score: bad arguments");
  CARCA_CHECK_ARG(table_rows > 0 || (p_bstride >= (int64_t)L * F && o_bstride >= (int64_t)T * F),
                  "knn_score: user strides %lld / %lld shorter than a user's rows", (long long)p_bstride,
                  (long long)o_bstride);
  CARCA_CHECK_ARG(table_rows > 0 ? (p_x && o_x) : o_a != nullptr, "knn_score: %s", table_rows > 0 ? "table mode needs p_x and o_x" : "dense mode needs o_a");
  CARCA_CHECK_SUPPORTED((long long)B * T < (1ll << 31), "knn_score: B*T=%lld rows", (long long)B * T);
  const bool vec = F % 4 == 0 && ((uintptr_t)p_a & 15) == 0 &&
                   (table_rows > 0 || (((uintptr_t)o_a & 15) == 0 && p_bstride % 4 == 0 && o_bstride % 4 == 0));
  const dim3 grid(min((B * T + 3) / 4, 16384));
  if (vec)
    hipLaunchKernelGGL(knn_score_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream_, p_a, (long)p_bstride, o_a,
                       (long)o_bstride, p_x, o_x, table_rows, y, B, L, T, F);
  else
    hipLaunchKernelGGL(knn_score_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream_, p_a, (long)p_bstride, o_a,
                       (long)o_bstride, p_x, o_x, table_rows, y, B, L, T, F);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_slot_decay_scale(const float* x, int ldx, float* out, int ldo, int B, int L, int d, float gamma,
                                     void* stream_) {
  CARCA_CHECK_ARG(x && out && B >= 1 && L >= 1 && d >= 1 && ldx >= d && ldo >= d, "slot_decay_scale: bad arguments");
  hipLaunchKernelGGL(slot_decay_scale_kernel, dim3((B * L * ldo + 255) / 256), dim3(256), 0, (hipStream_t)stream_, x, ldx,
                     out, ldo, B * L, L, d, gamma);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_l2norm_fwd(const float* x, int ldx, float* y, int ldy, int rows, int d, void* stream_) {
  CARCA_CHECK_ARG(x && y && rows >= 1 && d >= 1 && ldx >= d && ldy >= d, "l2norm_fwd: bad arguments");
  CARCA_CHECK_SUPPORTED(d <= 128 && ldy <= 128, "l2norm_fwd: d=%d / ldy=%d > 128", d, ldy);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream_, x, ldx, y, ldy, rows,
                     d);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_l2norm_bwd(const float* x, int ldx, const float* dy, int ld_dy, float* dx, int ld_dx, int rows,
                                int d, void* stream_) {
  CARCA_CHECK_ARG(x && dy && dx && rows >= 1 && d >= 1 && ldx >= d && ld_dy >= d && ld_dx >= d, "l2norm_bwd: bad arguments");
  CARCA_CHECK_SUPPORTED(d <= 128 && ld_dx <= 128, "l2norm_bwd: d=%d / ld_dx=%d > 128", d, ld_dx);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream_, x, ldx, dy, ld_dy, dx,
                     ld_dx, rows, d);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_add_positions(const float* x, int ldx, const float* pos, float* out, int ldo, int B, int T, int d,
                                   void* stream_) {
  CARCA_CHECK_ARG(x && pos && out && B >= 1 && T >= 1 && d >= 1 && ldx >= d && ldo >= d, "add_positions: bad arguments");
  const long n = (long)B * T * d;
  hipLaunchKernelGGL(add_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, ldx,
                     pos, out, ldo, (long)B * T, T, d);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_mha_core_drop(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                                   const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                                   float* out, int ldo, float* w_out, const CarcaDropout* drop, uint8_t* keep_out,
                                   void* stream_) {
  CARCA_CHECK_ARG(q && k && v && q_ids && k_ids && out, "mha_core: null pointer");
  CARCA_CHECK_ARG(B >= 1 && Tq >= 1 && Tk >= 1 && d >= 1 && H >= 1 && d % H == 0 && ldq >= d && ldk >= d && ldo >= d,
                  "mha_core: bad dims");
  CARCA_CHECK_SUPPORTED(Tk <= 64 * MHA_KPL && d / H <= 128, "mha_core: Tk=%d > %d keys per query, or d/H=%d > 128", Tk,
                        64 * MHA_KPL, d / H);
  CARCA_CHECK_ARG(!(drop && drop->p > 0.f) || (drop->p < 1.f), "mha_core: dropout p must be < 1");
  const long waves = (long)B * H * Tq;
  hipLaunchKernelGGL(mha_core_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, q, ldq, k, v, ldk,
                     q_ids, k_ids, B, Tq, Tk, d, H, has_causal, causal, out, ldo, w_out, make_drop(drop),
                     drop ? drop->site : 0u, keep_out);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_mha_core(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                              const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                              float* out, int ldo, float* w_out, void* stream_) {
  return carca_mha_core_drop(q, ldq, k, v, ldk, q_ids, k_ids, B, Tq, Tk, d, H, has_causal, causal, out, ldo, w_out, nullptr,
                             nullptr, stream_);
}

extern "C" int carca_mha_core_bwd_drop(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                                       const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                                       const float* d_out, int ldo, const float* d_w, float* dq, float* dk, float* dv,
                                       const uint8_t* keep, float keep_scale, void* stream_) {
  CARCA_CHECK_ARG(q && k && v && q_ids && k_ids && (d_out || d_w) && dq && dk && dv, "mha_core_bwd: null pointer");
  CARCA_CHECK_ARG(B >= 1 && Tq >= 1 && Tk >= 1 && d >= 1 && H >= 1 && d % H == 0 && ldq >= d && ldk >= d &&
                      (!d_out || ldo >= d),
                  "mha_core_bwd: bad dims");
  CARCA_CHECK_SUPPORTED(Tk <= 64 * MHA_KPL && d / H <= 128, "mha_core_bwd: Tk=%d > %d keys per query, or d/H=%d > 128", Tk,
                        64 * MHA_KPL, d / H);
  const long waves = (long)B * H * Tq;
  hipLaunchKernelGGL(mha_core_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, q, ldq, k, v,
                     ldk, q_ids, k_ids, B, Tq, Tk, d, H, has_causal, causal, d_out, ldo, d_w, dq, dk, dv, keep, keep_scale);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_mha_core_bwd(const float* q, int ldq, const float* k, const float* v, int ldk, const int32_t* q_ids,
                                  const int32_t* k_ids, int B, int Tq, int Tk, int d, int H, int has_causal, int causal,
                                  const float* d_out, int ldo, const float* d_w, float* dq, float* dk, float* dv,
                                  void* stream_) {
  return carca_mha_core_bwd_drop(q, ldq, k, v, ldk, q_ids, k_ids, B, Tq, Tk, d, H, has_causal, causal, d_out, ldo, d_w, dq, dk,
                                 dv, nullptr, 1.0f, stream_);
}
