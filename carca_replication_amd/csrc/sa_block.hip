// K2: SelfAttentionBlock.forward (carca.py:297-318) incl. MultiHeadAttention.forward (carca.py:228-265),
// causal = 0, eval mode / dropout p = 0.  One 4-wave workgroup per user, the whole block fused:
//   phase A  x -> LDS; q = LayerNorm1(x) -> LDS                               (one wave per row)
//   phase B  K = x W_K^T + b_K  [key][head-padded f];  V^T  [head-padded f][key]   (un-normed x, carca.py:299)
//   phase C  per 16-query tile, one wave, no further workgroup barrier:
//            Q^T -> scores^T -> masked softmax -> O^T (registers) -> + q (normed residual, carca.py:301-302)
//            -> LayerNorm2 -> ffn_1 -> LeakyReLU(0.01) -> ffn_2 -> + s (carca.py:304-316) -> y
// Rows that are padding (ids == 0) are computed like any other: the reference does not re-mask after a
// block (carca.py:318), they carry LayerNorm(0) = beta forward and are never attended.
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(256) void sa_block_kernel(const float* __restrict__ x, int ldx,
                                                       const int32_t* __restrict__ ids, float* __restrict__ y,
                                                       int ldy, int L, int d, int dh, const CarcaSaWeights w,
                                                       int residual, const CarcaSaSave sv) {
  using G = AttGeom<DPI, DHP, NH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                    // [64][SI]  x, later s2 = LayerNorm2(.)
  float* Qn = Xs + ATT_LMAX * G::SI;  // [64][SI]  LayerNorm1(x)
  float* Ks = Qn + ATT_LMAX * G::SI;  // [64][SO]
  float* Vt = Ks + ATT_LMAX * G::SO;  // [DPO][ATT_SK]

  const int u = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LT = (L + 15) >> 4;
  const int32_t* uid = ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);

  // ---- phase A -----------------------------------------------------------------------------------
  for (int r = wave; r < 16 * LT; r += 4) {
    float v0 = 0.f, v1 = 0.f;
    if (r < L) {
      const float* xr = x + ((size_t)u * L + r) * ldx;
      v0 = lane < d ? xr[lane] : 0.f;
      v1 = lane + 64 < d ? xr[lane + 64] : 0.f;
    }
    if (lane < DPI) Xs[r * G::SI + lane] = v0;
    if (lane + 64 < DPI) Xs[r * G::SI + lane + 64] = v1;
    if (r < L) row_layernorm(v0, v1, lane, d, w.ln1_w, w.ln1_b);
    if (lane < DPI) Qn[r * G::SI + lane] = v0;
    if (lane + 64 < DPI) Qn[r * G::SI + lane + 64] = v1;
    if (sv.qn && r < L) {
      float* qr = sv.qn + ((size_t)u * L + r) * DPI;
      if (lane < DPI) qr[lane] = v0;
      if (lane + 64 < DPI) qr[lane + 64] = v1;
    }
  }
  __syncthreads();

  // ---- phase B: K and V^T tiles, round-robin over the 4 waves ---------------------------------------
  {
    const int nk = G::NF * LT;
    for (int job = wave; job < 2 * nk; job += 4) {
      const bool isv = job >= nk;
      const int jj = isv ? job - nk : job;
      const int ft = jj / LT, st = jj - ft * LT;
      if (!isv)
        proj_tile_feat_major<DPI>(w.wk, w.bk, Xs, G::SI, Ks, G::SO, ft, st, lane,
                                  sv.kh ? sv.kh + (size_t)u * L * G::DPO : nullptr, G::DPO, L);
      else
        proj_tile_slot_major<DPI>(w.wv, w.bv, Xs, G::SI, Vt, ATT_SK, ft, st, lane,
                                  sv.vh ? sv.vh + (size_t)u * L * G::DPO : nullptr, G::DPO, L);
    }
  }
  __syncthreads();

  // ---- phase C ------------------------------------------------------------------------------------
  const float sqrt_dh = sqrtf((float)dh);
  const int ln = lane & 15, mq = lane >> 4;
  for (int qt = wave; qt < LT; qt += 4) {
    const int q = 16 * qt + ln;  // this lane's query slot
    const bool q_ok = (pmask >> q) & 1ull;
    f32x4 qfrag[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) qfrag[kg] = lds4(Qn + q * G::SI + 16 * kg + 4 * mq);

    // key 16kt+4mq+r may be attended iff both slots are real items and key <= query (tril, diagonal 0)
    unsigned okbits = 0;
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        const bool ok = q_ok && key <= q && ((pmask >> key) & 1ull);
        okbits |= (ok ? 1u : 0u) << (4 * kt + r);
      }
    const int nkt = qt + 1;  // key tiles that can hold a key <= query

    f32x4 o[G::NF];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 oh[G::NFH], p[ATT_LT];
      attend_head<DPI, DHP, NH>(qfrag, w.wq, w.bq, Ks, Vt, h, nkt, okbits, sqrt_dh, oh, p, lane,
                                (sv.qh && q < L) ? sv.qh + ((size_t)u * L + q) * G::DPO : nullptr);
#pragma unroll
      for (int ft = 0; ft < G::NFH; ++ft) o[h * G::NFH + ft] = oh[ft];
    }

    // s = attention + q (normed residual); LayerNorm2 over the d real features of the row
    float part = 0.f;
#pragma unroll
    for (int fi = 0; fi < G::NF; ++fi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = unpad_feature(16 * fi + 4 * mq + r, dh, DHP);
        float v = 0.f;
        if (j >= 0) v = o[fi][r] + (residual ? Qn[q * G::SI + j] : 0.f);
        o[fi][r] = v;
        part += v;
        if (sv.r && j >= 0 && q < L) sv.r[((size_t)u * L + q) * DPI + j] = v;  // LayerNorm2 input
      }
    const float inv_d = 1.0f / (float)d;
    const float mean = quad4_sum(part) * inv_d;
    part = 0.f;
#pragma unroll
    for (int fi = 0; fi < G::NF; ++fi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = unpad_feature(16 * fi + 4 * mq + r, dh, DHP);
        const float dv = j >= 0 ? o[fi][r] - mean : 0.f;
        o[fi][r] = dv;
        part += dv * dv;
      }
    const float rstd = 1.0f / sqrtf(quad4_sum(part) * inv_d + 1e-5f);
    // s2 -> LDS rows of this wave (Xs is dead after phase B; its pad columns are already 0)
#pragma unroll
    for (int fi = 0; fi < G::NF; ++fi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = unpad_feature(16 * fi + 4 * mq + r, dh, DHP);
        if (j >= 0) Xs[q * G::SI + j] = o[fi][r] * rstd * w.ln2_w[j] + w.ln2_b[j];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    f32x4 s2[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) {
      s2[kg] = lds4(Xs + q * G::SI + 16 * kg + 4 * mq);
      if (sv.s2 && q < L) *reinterpret_cast<f32x4*>(sv.s2 + ((size_t)u * L + q) * DPI + 16 * kg + 4 * mq) = s2[kg];
    }
    // ffn_1 + LeakyReLU: H1^T[f][query], kept in registers as the next product's Bt operand
    f32x4 h1[G::NKG];
#pragma unroll
    for (int ft = 0; ft < G::NKG; ++ft) {
      const float* wrow = w.w1 + (size_t)(16 * ft + ln) * DPI + 4 * mq;
      f32x4 acc = zero4();
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(glb4(wrow + 16 * kg), s2[kg], acc);
      acc = acc + glb4(w.b1 + 16 * ft + 4 * mq);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.f ? acc[r] : 0.01f * acc[r];
      h1[ft] = acc;
      if (sv.h1 && q < L) *reinterpret_cast<f32x4*>(sv.h1 + ((size_t)u * L + q) * DPI + 16 * ft + 4 * mq) = acc;
    }
    // ffn_2 + residual with s2, straight to global (pad columns come out as exact zeros)
#pragma unroll
    for (int ft = 0; ft < G::NKG; ++ft) {
      const float* wrow = w.w2 + (size_t)(16 * ft + ln) * DPI + 4 * mq;
      f32x4 acc = zero4();
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(glb4(wrow + 16 * kg), h1[kg], acc);
      acc = acc + glb4(w.b2 + 16 * ft + 4 * mq);
      if (residual) acc = acc + s2[ft];
      if (q < L && 16 * ft + 4 * mq < ldy)
        *reinterpret_cast<f32x4*>(y + ((size_t)u * L + q) * ldy + 16 * ft + 4 * mq) = acc;
    }
  }
}

template <int DPI, int DHP, int NH>
int launch_sa(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d,
              const CarcaSaWeights& w, int residual, const CarcaSaSave& sv, hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  const size_t lds_bytes = sizeof(float) * (2 * ATT_LMAX * G::SI + ATT_LMAX * G::SO + G::DPO * ATT_SK);
  auto kern = sa_block_kernel<DPI, DHP, NH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("sa_block_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds_bytes, stream, x, ldx, ids, y, ldy, L, d, d / NH, w, residual, sv);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

extern "C" int carca_sa_block_fwd(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d,
                                  int H, const CarcaSaWeights* w, int residual, const CarcaSaSave* save,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(x && ids && y && w, "sa_block_fwd: null pointer");
  CARCA_CHECK_ARG(B >= 1 && L >= 1 && d >= 1 && H >= 1 && d % H == 0, "sa_block_fwd: bad dims B=%d L=%d d=%d H=%d", B,
                  L, d, H);
  CARCA_CHECK_SUPPORTED(L <= CARCA_MAX_L, "sa_block_fwd: L=%d > %d profile slots per workgroup", L, CARCA_MAX_L);
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK) return CARCA_ERR_UNSUPPORTED;
  CARCA_CHECK_ARG(ldx >= d && ldy >= dpi && ldy % 4 == 0, "sa_block_fwd: need ldx >= d, ldy >= %d and ldy %% 4 == 0",
                  dpi);
  CarcaSaSave sv{};
  if (save) sv = *save;
  CARCA_ATT_DISPATCH(launch_sa, x, ldx, ids, y, ldy, B, L, d, *w, residual, sv, stream);
  carca_set_error("sa_block_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
  return CARCA_ERR_UNSUPPORTED;
}
