// K3 + K4: final LayerNorm (carca.py:421) + grouped CrossAttentionBlock.forward (carca.py:338-349).
//
// One 8-wave workgroup per user:
//   phase A  p = LayerNorm(encoder output) -> LDS                                   (one wave per row)
//   phase B  K [key][head-padded f] and V^T [head-padded f][key] of p, ONCE per user (the reference
//            recomputes them per target group, carca.py:424-428; same numbers)
//   phase C  every 16-target tile of every group is one wave job, all in registers:
//            Q^T -> scores^T -> masked softmax -> O^T -> y = sigmoid(w . (O + o) + b)
//            The residual never needs materialising: w.(O + o) = w_pad.O (head-padded order) + w.o.
//   masking  eval: a target attends every real profile slot; train: tril(diagonal=-1), i.e. target
//            slot i attends real profile slots j < i, so the first slot attends nothing and scores
//            sigmoid(w.o + b) (carca.py:339).  Pad targets (id 0) attend nothing.
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

struct GroupsDev {
  CarcaTargetGroup g[CARCA_MAX_GROUPS];
  int tile_start[CARCA_MAX_GROUPS + 1];
  int n;
};

template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(512) void cross_score_kernel(const float* __restrict__ p_raw, int ldp,
                                                          const int32_t* __restrict__ p_ids,
                                                          float* __restrict__ p_normed, const GroupsDev groups,
                                                          int ldo, int L, int d, int dh, const CarcaCaWeights w,
                                                          int residual, int training, const CarcaCaSave sv) {
  using G = AttGeom<DPI, DHP, NH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ps = lds;                    // [64][SI]
  float* Ks = Ps + ATT_LMAX * G::SI;  // [64][SO]
  float* Vt = Ks + ATT_LMAX * G::SO;  // [DPO][ATT_SK]

  const int u = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int LT = (L + 15) >> 4;
  const int32_t* uid = p_ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);

  // ---- phase A: final norm -------------------------------------------------------------------------
  for (int r = wave; r < 16 * LT; r += 8) {
    float v0 = 0.f, v1 = 0.f;
    if (r < L) {
      const float* xr = p_raw + ((size_t)u * L + r) * ldp;
      v0 = lane < d ? xr[lane] : 0.f;
      v1 = lane + 64 < d ? xr[lane + 64] : 0.f;
      if (w.ln_w) row_layernorm(v0, v1, lane, d, w.ln_w, w.ln_b);  // NULL: p is already normed
      if (p_normed) {
        float* pr = p_normed + ((size_t)u * L + r) * ldp;
        if (lane < ldp) pr[lane] = v0;
        if (lane + 64 < ldp) pr[lane + 64] = v1;
      }
    }
    if (lane < DPI) Ps[r * G::SI + lane] = v0;
    if (lane + 64 < DPI) Ps[r * G::SI + lane + 64] = v1;
  }
  __syncthreads();

  // ---- phase B: K and V^T ----------------------------------------------------------------------------
  {
    const int nk = G::NF * LT;
    for (int job = wave; job < 2 * nk; job += 8) {
      const bool isv = job >= nk;
      const int jj = isv ? job - nk : job;
      const int ft = jj / LT, st = jj - ft * LT;
      if (!isv)
        proj_tile_feat_major<DPI>(w.wk, w.bk, Ps, G::SI, Ks, G::SO, ft, st, lane,
                                  sv.kh ? sv.kh + (size_t)u * L * G::DPO : nullptr, G::DPO, L);
      else
        proj_tile_slot_major<DPI>(w.wv, w.bv, Ps, G::SI, Vt, ATT_SK, ft, st, lane,
                                  sv.vh ? sv.vh + (size_t)u * L * G::DPO : nullptr, G::DPO, L);
    }
  }
  __syncthreads();

  // ---- phase C: one wave per 16-target tile ------------------------------------------------------------
  const float sqrt_dh = sqrtf((float)dh);
  const float ffn_b = w.ffn_b[0];
  const int ln = lane & 15, mq = lane >> 4;
  const int ntiles = groups.tile_start[groups.n];
  for (int job = wave; job < ntiles; job += 8) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
      if (i < groups.n && job >= groups.tile_start[i]) gi = i;
    const CarcaTargetGroup grp = groups.g[gi];
    const int qt = job - groups.tile_start[gi];
    const int n = 16 * qt + ln;  // this lane's target slot
    const bool in_range = n < grp.N;
    const size_t row = (size_t)u * grp.N + (in_range ? n : grp.N - 1);
    const float* orow = grp.o + row * ldo + 4 * mq;
    f32x4 qfrag[G::NKG];
#pragma unroll
    for (int kg = 0; kg < G::NKG; ++kg) qfrag[kg] = glb4(orow + 16 * kg);
    const bool q_ok = in_range && grp.ids[row] != 0;

    unsigned okbits = 0;
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        const bool ok = q_ok && ((pmask >> key) & 1ull) && (!training || key < n);
        okbits |= (ok ? 1u : 0u) << (4 * kt + r);
      }
    const int nkt = training ? min(LT, qt + 1) : LT;

    float ypart = 0.f;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      f32x4 oh[G::NFH], p[ATT_LT];
      attend_head<DPI, DHP, NH>(qfrag, w.wq, w.bq, Ks, Vt, h, nkt, okbits, sqrt_dh, oh, p, lane,
                                (sv.qh[gi] && in_range) ? sv.qh[gi] + row * G::DPO : nullptr);
#pragma unroll
      for (int ft = 0; ft < G::NFH; ++ft) {
        const f32x4 wp = glb4(w.ffn_w_pad + h * DHP + 16 * ft + 4 * mq);
#pragma unroll
        for (int r = 0; r < 4; ++r) ypart += wp[r] * oh[ft][r];
      }
    }
    if (residual) {
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) {
        const f32x4 wv = glb4(w.ffn_w + 16 * kg + 4 * mq);
#pragma unroll
        for (int r = 0; r < 4; ++r) ypart += wv[r] * qfrag[kg][r];
      }
    }
    const float logit = quad4_sum(ypart) + ffn_b;
    if (mq == 0 && in_range) grp.y[row] = 1.0f / (1.0f + expf(-logit));
  }
}

template <int DPI, int DHP, int NH>
int launch_cross(const float* p_raw, int ldp, const int32_t* p_ids, float* p_normed, const GroupsDev& groups, int ldo,
                 int B, int L, int d, const CarcaCaWeights& w, int residual, int training, const CarcaCaSave& sv,
                 hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  const size_t lds_bytes = sizeof(float) * (ATT_LMAX * G::SI + ATT_LMAX * G::SO + G::DPO * ATT_SK);
  auto kern = cross_score_kernel<DPI, DHP, NH>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("cross_score_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(512), lds_bytes, stream, p_raw, ldp, p_ids, p_normed, groups, ldo, L, d,
                     d / NH, w, residual, training, sv);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

extern "C" int carca_cross_score_fwd(const float* p_raw, int ldp, const int32_t* p_ids, float* p_normed,
                                     const CarcaTargetGroup* groups, int ngroups, int ldo, int B, int L, int d, int H,
                                     const CarcaCaWeights* w, int residual, int training, const CarcaCaSave* save,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(p_raw && p_ids && groups && w, "cross_score_fwd: null pointer");
  CARCA_CHECK_ARG(ngroups >= 1 && ngroups <= CARCA_MAX_GROUPS, "cross_score_fwd: ngroups=%d outside 1..%d", ngroups,
                  CARCA_MAX_GROUPS);
  CARCA_CHECK_ARG(B >= 1 && L >= 1 && d >= 1 && H >= 1 && d % H == 0, "cross_score_fwd: bad dims");
  CARCA_CHECK_SUPPORTED(L <= CARCA_MAX_L, "cross_score_fwd: L=%d > %d profile slots per workgroup", L, CARCA_MAX_L);
  int dpi, dhp, dpo;
  if (carca_padded_dims(d, H, &dpi, &dhp, &dpo) != CARCA_OK) return CARCA_ERR_UNSUPPORTED;
  CARCA_CHECK_ARG(ldp >= d && ldo >= dpi && ldo % 4 == 0, "cross_score_fwd: need ldp >= d, ldo >= %d, ldo %% 4 == 0",
                  dpi);
  GroupsDev gd{};
  int t = 0;
  for (int i = 0; i < ngroups; ++i) {
    CARCA_CHECK_ARG(groups[i].o && groups[i].ids && groups[i].y && groups[i].N >= 1, "cross_score_fwd: group %d malformed",
                    i);
    gd.g[i] = groups[i];
    gd.tile_start[i] = t;
    t += (groups[i].N + 15) / 16;
  }
  gd.tile_start[ngroups] = t;
  gd.n = ngroups;
  CarcaCaSave sv{};
  if (save) sv = *save;
  CARCA_ATT_DISPATCH(launch_cross, p_raw, ldp, p_ids, p_normed, gd, ldo, B, L, d, *w, residual, training, sv, stream);
  carca_set_error("cross_score_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
  return CARCA_ERR_UNSUPPORTED;
}
