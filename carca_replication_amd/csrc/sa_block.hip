// K2: SelfAttentionBlock.forward (carca.py:297-318) incl. MultiHeadAttention.forward (carca.py:228-265),
// causal = 0, eval mode / dropout p = 0.  One 16-wave workgroup per user, the whole block fused:
//   A  x -> LDS; q = LayerNorm1(x) -> LDS                                (one wave per row)
//   B  K = x W_K^T + b_K [key][head-padded f]; V^T [head-padded f][key]   (un-normed x, carca.py:299)
//   C1 per (16-query tile, head): Q^T -> scores^T -> masked softmax -> O^T in registers, + q (normed
//      residual, carca.py:301-302) -> LDS
//   C2 LayerNorm2 rows; C3 ffn_1 + LeakyReLU(0.01); C4 ffn_2 + s (carca.py:304-316) -> y
// Rows that are padding (ids == 0) are computed like any other: the reference does not re-mask after a
// block (carca.py:318), they carry LayerNorm(0) = beta forward and are never attended.
#include <hip/hip_ext.h>
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

// 16 waves per user: the block is split into many short wave jobs so that four waves per SIMD hide the
// latency of each other's weight-fragment loads (a 4-wave, one-job-per-query-tile version took 1.5x longer)
//   A0 x -> LDS (16-byte coalesced)      A1 LayerNorm1 rows (wave per row)        | barrier after each
//   B  K / V^T tiles                      C1 (query tile, head): attention + residual -> R (LDS, plain order)
//   C2 LayerNorm2 rows in place           C3 (query tile, f tile): ffn_1 + LeakyReLU -> H1 (LDS)
//   C4 (query tile, f tile): ffn_2 + residual -> y
template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(1024) void sa_block_kernel_w16(const float* __restrict__ x, int ldx,
                                                            const int32_t* __restrict__ ids,
                                                            float* __restrict__ y, int ldy, int L, int d, int dh,
                                                            const CarcaSaWeights w, int residual,
                                                            const CarcaSaSave sv, const DropCfg dc_arg, unsigned site,
                                                            unsigned long long* stamps, int nparts) {
  using G = AttGeom<DPI, DHP, NH>;
  const DropCfg dc = drop_resolve(dc_arg);
  static_assert(G::SO >= G::SI, "H1 reuses the K image");
#define SA_STAMP(i)                                                                       \
  do {                                                                                    \
    if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  SA_STAMP(0);
  constexpr int NW = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Xs = lds;                    // [64][SI]  x -> R -> S2
  float* Qn = Xs + ATT_LMAX * G::SI;  // [64][SI]
  float* Ks = Qn + ATT_LMAX * G::SI;  // [64][SO]  K -> H1
  float* Vt = Ks + ATT_LMAX * G::SO;  // [DPO][ATT_SK]

  // With fewer users than CUs a user is shared by TWO workgroups: each owns a set of 16-query tiles (balanced for
  // the causal cost: {1,2} | {0,3} of four) and everything row-wise about them; K / V^T are projected by both, but
  // only for the key tiles its queries can attend.  No data passes between the two.
  const int u = blockIdx.x / nparts, part = blockIdx.x - u * nparts;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it (jobs, tiles, heads) is uniform
  const int LT = (L + 15) >> 4;
  // own query tiles (bit t of tmask), their list, and the number of key tiles they need
  const unsigned tmask = nparts == 1 ? (1u << LT) - 1u
                         : LT == 4   ? (part == 0 ? 0x6u : 0x9u)
                         : LT == 3   ? (part == 0 ? 0x4u : 0x3u)
                                     : (part == 0 ? 0x2u : 0x1u);
  unsigned own_packed = 0;  // 4 bits per own tile index
  int n_own = 0, kmax = 0;
#pragma unroll
  for (int t = 0; t < ATT_LT; ++t)
    if (t < LT && ((tmask >> t) & 1u)) {
      own_packed |= (unsigned)t << (4 * n_own);
      ++n_own;
      kmax = t + 1;
    }
  auto own = [&](int i) { return (int)((own_packed >> (4 * i)) & 15u); };
  const bool owns_last = (tmask >> (LT - 1)) & 1u;  // this workgroup projects every key: it saves K / V for backward
  const int32_t* uid = ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);
  const size_t ubase = (size_t)u * L;
  const int ln = lane & 15, mq = lane >> 4;

  // ---- A0 ---------------------------------------------------------------------------------------------
  constexpr int V4 = DPI / 4;
  const bool vec_ok = (ldx % 4 == 0) && ldx >= DPI;  // internal buffers: padded rows with zeroed pad columns
  if (vec_ok) {  // (both passes' loads in flight before the first LDS write: clamped rows, no load under a branch)
    constexpr int A0_IT = (ATT_LMAX * V4 + 1023) / 1024;
    f32x4 xv[A0_IT];
#pragma unroll
    for (int j = 0; j < A0_IT; ++j) {
      const int i = tid + 1024 * j, r = min(i / V4, L - 1), c4 = i % V4;
      xv[j] = gload4(x, (int)((ubase + r) * ldx) + 4 * c4);
    }
#pragma unroll
    for (int j = 0; j < A0_IT; ++j) {
      const int i = tid + 1024 * j, r = i / V4, c4 = i - r * V4;
      if (i < 16 * LT * V4) *reinterpret_cast<f32x4*>(Xs + r * G::SI + 4 * c4) = r < L ? xv[j] : zero4();
    }
  } else {
    for (int i = tid; i < 16 * LT * V4; i += 1024) {
      const int r = i / V4, c4 = i - r * V4;
      f32x4 v = zero4();
      if (r < L) {
        const float* xr = x + (ubase + r) * ldx + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 4 * c4 + e < d ? xr[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(Xs + r * G::SI + 4 * c4) = v;
    }
  }
  __syncthreads();
  SA_STAMP(1);
  // ---- A1 ---------------------------------------------------------------------------------------------
  for (int ri = wave; ri < 16 * n_own; ri += NW) {
    const int r = 16 * own(ri >> 4) + (ri & 15);
    float v0 = lane < DPI ? Xs[r * G::SI + lane] : 0.f;
    float v1 = lane + 64 < DPI ? Xs[r * G::SI + lane + 64] : 0.f;
    if (lane >= d) v0 = 0.f;
    if (lane + 64 >= d) v1 = 0.f;
    if (r < L) row_layernorm(v0, v1, lane, d, w.ln1_w, w.ln1_b);
    if (lane < DPI) Qn[r * G::SI + lane] = v0;
    if (lane + 64 < DPI) Qn[r * G::SI + lane + 64] = v1;
    if (sv.qn && r < L) {
      float* qr = sv.qn + (ubase + r) * DPI;
      if (lane < DPI) qr[lane] = v0;
      if (lane + 64 < DPI) qr[lane + 64] = v1;
    }
  }
  SA_STAMP(2);
  // ---- B (reads Xs only, so no barrier is needed between A1 and B) ---------------------------------------
  {
    const int nk = G::NF * kmax;  // only the key tiles the own queries can attend (causal)
    for (int job = wave; job < 2 * nk; job += NW) {
      const bool isv = job >= nk;
      const int jj = isv ? job - nk : job;
      const int ft = jj / kmax, st = jj - ft * kmax;
      if (!isv)
        proj_tile_feat_major<DPI>(w.wk, w.bk, Xs, G::SI, Ks, G::SO, ft, st, lane,
                                  (sv.kh && owns_last) ? sv.kh + ubase * G::DPO : nullptr, G::DPO, L);
      else
        proj_tile_slot_major<DPI>(w.wv, w.bv, Xs, G::SI, Vt, ATT_SK, ft, st, lane,
                                  (sv.vh && owns_last) ? sv.vh + ubase * G::DPO : nullptr, G::DPO, L);
    }
  }
  __syncthreads();
  SA_STAMP(3);

  // ---- C1: attention per (query tile, head) -----------------------------------------------------------------
  const float sqrt_dh = sqrtf((float)dh);
  for (int job = wave; job < n_own * NH; job += NW) {
    const int qi = job / NH, h = job - qi * NH;
    const int qt = own(qi);
    const int q = 16 * qt + ln;
    const bool q_ok = (pmask >> q) & 1ull;
    f32x4 qfrag[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) qfrag[kg] = lds4(Qn + q * G::SI + 16 * kg + 4 * mq);
    // bit (4 kt + r) = this lane's query may attend key 16 kt + 4 mq + r: real keys (pmask) at or before it (q < 64)
    const unsigned long long allowed = q_ok ? pmask & (~0ull >> (63 - q)) : 0ull;
    unsigned okbits = 0;
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt) okbits |= ((unsigned)(allowed >> (16 * kt + 4 * mq)) & 15u) << (4 * kt);
    f32x4 oh[G::NFH], p[ATT_LT];
    const unsigned midx = (unsigned)((((size_t)u * NH + h) * L + (q < L ? q : 0)) * L);
    attend_head<DPI, DHP, NH, true>(qfrag, w.wq, w.bq, Ks, Vt, h, qt + 1, okbits, sqrt_dh, oh, p, lane,
                              (sv.qh && q < L) ? sv.qh + (ubase + q) * G::DPO : nullptr, &dc, site, midx,
                              (sv.m_attn && q < L) ? sv.m_attn + midx : nullptr, L);
    // r = attention (+ q): plain feature order, into the dead x image (pad columns stay 0)
#pragma unroll
    for (int ft = 0; ft < G::NFH; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fr = 16 * ft + 4 * mq + r;
        if (fr < dh) {
          const int j = h * dh + fr;
          const float v = oh[ft][r] + (residual ? Qn[q * G::SI + j] : 0.f);
          Xs[q * G::SI + j] = v;
          if (sv.r && q < L) sv.r[(ubase + q) * DPI + j] = v;
        }
      }
  }
  __syncthreads();
  SA_STAMP(4);
  // ---- C2: LayerNorm2 rows, in place -------------------------------------------------------------------------
  for (int ri = wave; ri < 16 * n_own; ri += NW) {
    const int r = 16 * own(ri >> 4) + (ri & 15);
    float v0 = lane < d ? Xs[r * G::SI + lane] : 0.f;
    float v1 = lane + 64 < d ? Xs[r * G::SI + lane + 64] : 0.f;
    row_layernorm(v0, v1, lane, d, w.ln2_w, w.ln2_b);
    if (lane < DPI) Xs[r * G::SI + lane] = v0;
    if (lane + 64 < DPI) Xs[r * G::SI + lane + 64] = v1;
    if (sv.s2 && r < L) {
      float* sr = sv.s2 + (ubase + r) * DPI;
      if (lane < DPI) sr[lane] = v0;
      if (lane + 64 < DPI) sr[lane + 64] = v1;
    }
  }
  __syncthreads();
  SA_STAMP(5);
  // ---- C3: ffn_1 + LeakyReLU per (query tile, f tile) -> H1 (over the dead K image) ---------------------------
  float* H1 = Ks;
  for (int job = wave; job < n_own * G::NKG; job += NW) {
    const int qi = job / G::NKG, ft = job - qi * G::NKG;
    const int qt = own(qi);
    const int q = 16 * qt + ln;
    const float* srow = Xs + q * G::SI + 4 * mq;
    f32x4 wf[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) wf[kg] = gload4(w.w1, wfrag_off(ft, kg, G::NKG, lane));
    const f32x4 bias = gload4(w.b1, 16 * ft + 4 * mq);
    CARCA_PIN_LOADS();
    f32x4 acc = zero4();
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(wf[kg], lds4(srow + 16 * kg), acc);
    acc = acc + bias;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.f ? acc[r] : 0.01f * acc[r];
    if (dc.thresh) {  // dropout1 (carca.py:309)
      const unsigned e0 = (unsigned)((ubase + (q < L ? q : 0)) * DPI + 16 * ft + 4 * mq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool keep = drop_keep(dc, site + 1, e0 + r);
        acc[r] = keep ? acc[r] * dc.scale : 0.f;
        if (sv.m_ffn1 && q < L) sv.m_ffn1[e0 + r] = keep ? 1 : 0;
      }
    }
    *reinterpret_cast<f32x4*>(H1 + q * G::SI + 16 * ft + 4 * mq) = acc;
    if (sv.h1 && q < L) *reinterpret_cast<f32x4*>(sv.h1 + (ubase + q) * DPI + 16 * ft + 4 * mq) = acc;
  }
  __syncthreads();
  SA_STAMP(6);
  // ---- C4: ffn_2 + residual per (query tile, f tile) -> y ------------------------------------------------------
  for (int job = wave; job < n_own * G::NKG; job += NW) {
    const int qi = job / G::NKG, ft = job - qi * G::NKG;
    const int qt = own(qi);
    const int q = 16 * qt + ln;
    const float* hrow = H1 + q * G::SI + 4 * mq;
    f32x4 wf[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) wf[kg] = gload4(w.w2, wfrag_off(ft, kg, G::NKG, lane));
    const f32x4 bias = gload4(w.b2, 16 * ft + 4 * mq);
    CARCA_PIN_LOADS();
    f32x4 acc = zero4();
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(wf[kg], lds4(hrow + 16 * kg), acc);
    acc = acc + bias;
    if (dc.thresh) {  // dropout2 (carca.py:312), before the residual
      const unsigned e0 = (unsigned)((ubase + (q < L ? q : 0)) * DPI + 16 * ft + 4 * mq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool keep = drop_keep(dc, site + 2, e0 + r);
        acc[r] = keep ? acc[r] * dc.scale : 0.f;
        if (sv.m_ffn2 && q < L) sv.m_ffn2[e0 + r] = keep ? 1 : 0;
      }
    }
    if (residual) acc = acc + lds4(Xs + q * G::SI + 16 * ft + 4 * mq);
    if (q < L && 16 * ft + 4 * mq < ldy) *reinterpret_cast<f32x4*>(y + (ubase + q) * ldy + 16 * ft + 4 * mq) = acc;
  }
  SA_STAMP(7);
#undef SA_STAMP
}

template <int DPI, int DHP, int NH>
int launch_sa(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d,
              const CarcaSaWeights& w, int residual, const CarcaSaSave& sv, const DropCfg& dc, unsigned site,
              hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  const size_t lds_bytes = sizeof(float) * (2 * ATT_LMAX * G::SI + ATT_LMAX * G::SO + G::DPO * ATT_SK);
  const void* kern = (const void*)sa_block_kernel_w16<DPI, DHP, NH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("sa_block_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  // two workgroups per user while that still fits the chip in one round (tuning key 1: 1 = never, 2 = always)
  static int num_cus = 0;
  if (num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    num_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                  ? prop.multiProcessorCount : 256;
  }
  const int tune = carca_tuning(CARCA_TUNE_ATTN_VARIANT);
  const int nparts = (L > 16 && tune != 1 && (tune == 2 || 2 * B <= num_cus)) ? 2 : 1;
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))  // (timing events bound to this dispatch: carca_forward's ev[4], ev[5])
    hipExtLaunchKernelGGL((sa_block_kernel_w16<DPI, DHP, NH>), dim3(B * nparts), dim3(1024), lds_bytes, stream, e0, e1, 0, x,
                          ldx, ids, y, ldy, L, d, d / NH, w, residual, sv, dc, site, carca_debug_buffer(), nparts);
  else
    hipLaunchKernelGGL((sa_block_kernel_w16<DPI, DHP, NH>), dim3(B * nparts), dim3(1024), lds_bytes, stream, x, ldx, ids,
                       y, ldy, L, d, d / NH, w, residual, sv, dc, site, carca_debug_buffer(), nparts);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

int carca_sa_eval_launch(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d, int H,
                         const CarcaSaWeights* w, int residual, int pads_uniform, hipStream_t stream);  // sa_eval.hip

static int sa_check(const char* who, const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d,
                    int H, const CarcaSaWeights* w, int* dpi) {
  CARCA_CHECK_ARG(x && ids && y && w, "%s: null pointer", who);
  CARCA_CHECK_ARG(B >= 1 && L >= 1 && d >= 1 && H >= 1 && d % H == 0, "%s: bad dims B=%d L=%d d=%d H=%d", who, B, L, d, H);
  CARCA_CHECK_SUPPORTED(L <= CARCA_MAX_L, "%s: L=%d > %d profile slots per workgroup", who, L, CARCA_MAX_L);
  int dhp, dpo;
  if (carca_padded_dims(d, H, dpi, &dhp, &dpo) != CARCA_OK) return CARCA_ERR_UNSUPPORTED;
  CARCA_CHECK_ARG(ldx >= d && ldy >= *dpi && ldy % 4 == 0, "%s: need ldx >= d, ldy >= %d and ldy %% 4 == 0", who, *dpi);
  return CARCA_OK;
}

extern "C" int carca_sa_block_eval(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d, int H,
                                   const CarcaSaWeights* w, int residual, int pads_uniform, void* stream_) {
  int dpi;
  if (int rc = sa_check("sa_block_eval", x, ldx, ids, y, ldy, B, L, d, H, w, &dpi)) return rc;
  // (the eval kernel reads rows with 16-byte loads: padded internal buffers; anything else, and tuning key 6 = 1, takes
  // the training kernel without its extras)
  if (ldx % 4 == 0 && ldx >= dpi && carca_tuning(6) != 1)
    return carca_sa_eval_launch(x, ldx, ids, y, ldy, B, L, d, H, w, residual, pads_uniform, (hipStream_t)stream_);
  return carca_sa_block_fwd(x, ldx, ids, y, ldy, B, L, d, H, w, residual, nullptr, nullptr, stream_);
}

extern "C" int carca_sa_block_fwd(const float* x, int ldx, const int32_t* ids, float* y, int ldy, int B, int L, int d,
                                  int H, const CarcaSaWeights* w, int residual, const CarcaSaSave* save,
                                  const CarcaDropout* drop, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int dpi;
  if (int rc = sa_check("sa_block_fwd", x, ldx, ids, y, ldy, B, L, d, H, w, &dpi)) return rc;
  if (!save && !(drop && drop->p > 0.f) && ldx % 4 == 0 && ldx >= dpi && carca_tuning(6) != 1)
    return carca_sa_eval_launch(x, ldx, ids, y, ldy, B, L, d, H, w, residual, 0, stream);
  int dhp, dpo;
  carca_padded_dims(d, H, &dpi, &dhp, &dpo);
  CarcaSaSave sv{};
  if (save) sv = *save;
  const DropCfg dc = make_drop(drop);
  CARCA_CHECK_ARG(!(drop && drop->p >= 1.0f), "sa_block_fwd: dropout p must be < 1");
  CARCA_ATT_DISPATCH(launch_sa, x, ldx, ids, y, ldy, B, L, d, *w, residual, sv, dc, drop ? drop->site : 0u, stream);
  carca_set_error("sa_block_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
  return CARCA_ERR_UNSUPPORTED;
}
