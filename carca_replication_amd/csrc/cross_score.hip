// K3 + K4: final LayerNorm (carca.py:421) + grouped CrossAttentionBlock.forward (carca.py:338-349).
//
// One 16-wave workgroup per user:
//   phase A  p = LayerNorm(encoder output) -> LDS                                   (one wave per row)
//   phase B  K [key][head-padded f] and V^T [head-padded f][key] of p, ONCE per user (the reference
//            recomputes them per target group, carca.py:424-428; same numbers)
//   phase C  every (16-target tile, head) of every group is one wave job, all in registers:
//            Q^T -> scores^T -> masked softmax -> O^T -> partial logit; y = sigmoid(sum_h + w.o + b)
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

// NW = 16 waves per user; phase C is split into (16-target tile, head) wave jobs so that four waves per SIMD
// overlap each other's weight-fragment latency; per-head partial logits meet in LDS.
// NW = 8 is the same kernel for batches of many users per CU: two 8-wave workgroups share a CU (LDS permitting), so
// that one user's prologue / K,V phase / barriers run under the other's target jobs, and a user's N = 101 targets
// (21 jobs) fill 21 of 24 wave slots instead of 21 of 32.
#define CROSS_TPR 16  // target tiles per round (jobs per round = 16 * H >= 16 waves)
template <int DPI, int DHP, int NH, int NW>
__global__ __launch_bounds__(NW * 64, 4) void cross_score_kernel_w16(const float* __restrict__ p_raw, int ldp,
                                                               const int32_t* __restrict__ p_ids,
                                                               float* __restrict__ p_normed,
                                                               const GroupsDev groups, int ldo, int L, int d, int dh,
                                                               const CarcaCaWeights w, int residual, int training,
                                                               const CarcaCaSave sv, const DropCfg dc_arg,
                                                               unsigned site, int nparts, unsigned long long* stamps) {
  const DropCfg dc = drop_resolve(dc_arg);
#define CA_STAMP(i)                                                                                \
  do {                                                                                             \
    if (stamps && threadIdx.x == 0) stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  CA_STAMP(0);
  using G = AttGeom<DPI, DHP, NH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ps = lds;                     // [64][SI]
  float* Ks = Ps + ATT_LMAX * G::SI;   // [64][SO]
  float* Vt = Ks + ATT_LMAX * G::SO;   // [DPO][ATT_SK]
  float* Yp = Ps;                      // [CROSS_TPR][NH][16] partial logits (the final-norm image is dead after phase B)

  // With fewer users than CUs a user's target tiles are shared by TWO workgroups; both build the same final-norm /
  // K / V^T images (nothing passes between them), the first one writes the copies kept for the backward pass.
  const int u = blockIdx.x / nparts, part = blockIdx.x - u * nparts;
  const bool saver = part == 0;
  if (!saver) p_normed = nullptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it (jobs, tiles, heads) is uniform
  const int LT = (L + 15) >> 4;
  const int32_t* uid = p_ids + (size_t)u * L;
  const unsigned long long pmask = __ballot(lane < L && uid[lane < L ? lane : 0] != 0);
  const size_t ubase = (size_t)u * L;
  const int ln = lane & 15, mq = lane >> 4;

  // ---- A: rows -> final norm -> LDS.  A wave takes rows wave, wave+16, ... straight from HBM (lane = column, two per
  // lane), all of them in flight before the first reduction, and normalises them interleaved: a row's two wave
  // reductions are a ~1 k-cycle dependent chain (one row at a time through LDS read 2.8 k + 4.8 k cycles).
  {
    constexpr int RPW = ATT_LMAX / NW;
    float v0[RPW], v1[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = wave + NW * j;
      const int off = (int)((ubase + (r < L ? r : 0)) * ldp);
      const float a = gload1(p_raw, off + (lane < d ? lane : 0)), b = gload1(p_raw, off + (lane + 64 < d ? lane + 64 : 0));
      v0[j] = (r < L && lane < d) ? a : 0.f;
      v1[j] = (r < L && lane + 64 < d) ? b : 0.f;
    }
    CA_STAMP(1);
    if (w.ln_w) {
      const float w0 = lane < d ? w.ln_w[lane] : 0.f, w1 = lane + 64 < d ? w.ln_w[lane + 64] : 0.f;
      const float b0 = lane < d ? w.ln_b[lane] : 0.f, b1 = lane + 64 < d ? w.ln_b[lane + 64] : 0.f;
#pragma unroll
      for (int j = 0; j < RPW; ++j) {  // (rows >= L are normalised too, branch-free, and zeroed again below)
        row_layernorm_regs(v0[j], v1[j], lane, d, w0, w1, b0, b1);
        const bool live = wave + NW * j < L;
        v0[j] = live ? v0[j] : 0.f;
        v1[j] = live ? v1[j] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int r = wave + NW * j;
      if (r < 16 * LT) {
        if (lane < DPI) Ps[r * G::SI + lane] = v0[j];
        if (lane + 64 < DPI) Ps[r * G::SI + lane + 64] = v1[j];
      }
      if (p_normed && r < L) {
        float* pr = p_normed + (ubase + r) * ldp;
        if (lane < ldp) pr[lane] = v0[j];
        if (lane + 64 < ldp) pr[lane + 64] = v1[j];
      }
    }
    __syncthreads();
  }
  CA_STAMP(2);
  // ---- B: K and V^T ---------------------------------------------------------------------------------------
  {
    const int nk = G::NF * LT;
    for (int job = wave; job < 2 * nk; job += NW) {
      const bool isv = job >= nk;
      const int jj = isv ? job - nk : job;
      const int st = jj % LT;
      const int ft = jj / LT;
      if (!isv)
        proj_tile_feat_major<DPI>(w.wk, w.bk, Ps, G::SI, Ks, G::SO, ft, st, lane,
                                  (saver && sv.kh) ? sv.kh + ubase * G::DPO : nullptr, G::DPO, L);
      else
        proj_tile_slot_major<DPI>(w.wv, w.bv, Ps, G::SI, Vt, ATT_SK, ft, st, lane,
                                  (saver && sv.vh) ? sv.vh + ubase * G::DPO : nullptr, G::DPO, L);
    }
  }
  __syncthreads();
  CA_STAMP(3);

  // ---- C: rounds of CROSS_TPR target tiles; job = (tile, head) -----------------------------------------------
  const float sqrt_dh = sqrtf((float)dh);
  const float ffn_b = w.ffn_b[0];
  const int all_tiles = groups.tile_start[groups.n];
  const int per_part = (all_tiles + nparts - 1) / nparts;
  const int ntiles = min(all_tiles, (part + 1) * per_part);
  for (int t0 = part * per_part; t0 < ntiles; t0 += CROSS_TPR) {
    const int nt = min(CROSS_TPR, ntiles - t0);
    for (int job = wave; job < nt * NH; job += NW) {
      const int tl = job / NH, h = job - tl * NH;
      const int tile = t0 + tl;
      int gi = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
        if (i < groups.n && tile >= groups.tile_start[i]) gi = i;
      const CarcaTargetGroup grp = groups.g[gi];
      const int qt = tile - groups.tile_start[gi];
      const int n = 16 * qt + ln;
      const bool in_range = n < grp.N;
      const size_t row = (size_t)u * grp.N + (in_range ? n : grp.N - 1);
      f32x4 qfrag[G::NKG];
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) qfrag[kg] = gload4(grp.o, (int)(row * ldo) + 4 * mq + 16 * kg);
      const bool q_ok = in_range && grp.ids[row] != 0;
      // bit (4 kt + r) = this lane's target may attend key 16 kt + 4 mq + r: real keys (pmask), when training only those
      // before its own slot
      unsigned long long allowed = q_ok ? pmask : 0ull;
      if (training) allowed &= n >= 64 ? ~0ull : (1ull << n) - 1ull;
      unsigned okbits = 0;
#pragma unroll
      for (int kt = 0; kt < ATT_LT; ++kt) okbits |= ((unsigned)(allowed >> (16 * kt + 4 * mq)) & 15u) << (4 * kt);
      const int nkt = training ? min(LT, qt + 1) : LT;
      // the residual part of the logit (w . o, once per target) first: it is the last use of the target row's
      // fragments outside the Q projection, so they die early instead of living through the whole head
      float ypart = 0.f;
      if (residual && h == 0) {
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) {
          const f32x4 wv = gload4(w.ffn_w, 16 * kg + 4 * mq);
#pragma unroll
          for (int r = 0; r < 4; ++r) ypart += wv[r] * qfrag[kg][r];
        }
      }
      f32x4 oh[G::NFH], p[ATT_LT];
      const unsigned midx = (unsigned)((((size_t)u * NH + h) * grp.N + (in_range ? n : 0)) * L);
      attend_head<DPI, DHP, NH>(qfrag, w.wq, w.bq, Ks, Vt, h, nkt, okbits, sqrt_dh, oh, p, lane,
                                (sv.qh[gi] && in_range) ? sv.qh[gi] + row * G::DPO : nullptr, &dc, site + gi, midx,
                                (sv.m_attn[gi] && in_range) ? sv.m_attn[gi] + midx : nullptr, L);
#pragma unroll
      for (int ft = 0; ft < G::NFH; ++ft) {
        const f32x4 wp = gload4(w.ffn_w_pad, h * DHP + 16 * ft + 4 * mq);
#pragma unroll
        for (int r = 0; r < 4; ++r) ypart += wp[r] * oh[ft][r];
      }
      ypart = quad4_sum(ypart);
      if (mq == 0) Yp[(tl * NH + h) * 16 + ln] = ypart;
    }
    __syncthreads();
    if (tid < nt * 16) {
      const int tl = tid >> 4, l16 = tid & 15;
      const int tile = t0 + tl;
      int gi = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
        if (i < groups.n && tile >= groups.tile_start[i]) gi = i;
      const CarcaTargetGroup grp = groups.g[gi];
      const int n = 16 * (tile - groups.tile_start[gi]) + l16;
      if (n < grp.N) {
        float logit = ffn_b;
#pragma unroll
        for (int h = 0; h < NH; ++h) logit += Yp[(tl * NH + h) * 16 + l16];
        grp.y[(size_t)u * grp.N + n] = 1.0f / (1.0f + expf(-logit));
      }
    }
    __syncthreads();
  }
  CA_STAMP(4);
#undef CA_STAMP
}

template <int DPI, int DHP, int NH>
constexpr size_t cross_lds_bytes() {
  using G = AttGeom<DPI, DHP, NH>;
  return sizeof(float) * (ATT_LMAX * G::SI + ATT_LMAX * G::SO + G::DPO * ATT_SK);
}

template <int DPI, int DHP, int NH, int NW>
int launch_cross_nw(const float* p_raw, int ldp, const int32_t* p_ids, float* p_normed, const GroupsDev& groups, int ldo,
                 int B, int L, int d, const CarcaCaWeights& w, int residual, int training, const CarcaCaSave& sv,
                 const DropCfg& dc, unsigned site, int nparts, hipStream_t stream) {
  using G = AttGeom<DPI, DHP, NH>;
  static_assert(CROSS_TPR * NH * 16 <= ATT_LMAX * G::SI, "partial logits must fit the final-norm image they alias");
  const size_t lds_bytes = cross_lds_bytes<DPI, DHP, NH>();
  auto kern = cross_score_kernel_w16<DPI, DHP, NH, NW>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("cross_score_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(B * nparts), dim3(NW * 64), lds_bytes, stream, p_raw, ldp, p_ids, p_normed, groups, ldo, L,
                     d, d / NH, w, residual, training, sv, dc, site, nparts, carca_debug_buffer());
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

// Variant choice (tuning key 1: 1 = one 16-wave workgroup per user, 2 = always two, 3 = one 8-wave workgroup):
//   2 B <= #CUs          two 16-wave workgroups per user, the target tiles halved between them (latency regime)
//   B > #CUs and two workgroups' LDS fit a CU   one 8-wave workgroup per user, two resident per CU (more users than
//                        CUs: 272..448 users take 47 us this way, 52-54 us as two rounds of 16-wave workgroups)
//   otherwise            one 16-wave workgroup per user
template <int DPI, int DHP, int NH>
int launch_cross(const float* p_raw, int ldp, const int32_t* p_ids, float* p_normed, const GroupsDev& groups, int ldo,
                 int B, int L, int d, const CarcaCaWeights& w, int residual, int training, const CarcaCaSave& sv,
                 const DropCfg& dc, unsigned site, hipStream_t stream) {
  const int num_cus = carca_num_cus();
  const int tune = carca_tuning(CARCA_TUNE_ATTN_VARIANT);
  const int nparts = (groups.tile_start[groups.n] > 1 && tune != 1 && tune != 3 && (tune == 2 || 2 * B <= num_cus)) ? 2 : 1;
  constexpr bool pair_fits = 2 * cross_lds_bytes<DPI, DHP, NH>() <= 160 * 1024;
  if (pair_fits && nparts == 1 && (tune == 3 || (tune == 0 && B > num_cus)))
    return launch_cross_nw<DPI, DHP, NH, 8>(p_raw, ldp, p_ids, p_normed, groups, ldo, B, L, d, w, residual, training, sv,
                                            dc, site, nparts, stream);
  return launch_cross_nw<DPI, DHP, NH, 16>(p_raw, ldp, p_ids, p_normed, groups, ldo, B, L, d, w, residual, training, sv,
                                           dc, site, nparts, stream);
}

}  // namespace

extern "C" int carca_cross_score_fwd(const float* p_raw, int ldp, const int32_t* p_ids, float* p_normed,
                                     const CarcaTargetGroup* groups, int ngroups, int ldo, int B, int L, int d, int H,
                                     const CarcaCaWeights* w, int residual, int training, const CarcaCaSave* save,
                                     const CarcaDropout* drop, void* stream_) {
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
  const DropCfg dc = make_drop(drop);
  CARCA_CHECK_ARG(!(drop && drop->p >= 1.0f), "cross_score_fwd: dropout p must be < 1");
  CARCA_ATT_DISPATCH(launch_cross, p_raw, ldp, p_ids, p_normed, gd, ldo, B, L, d, *w, residual, training, sv, dc,
                     drop ? drop->site : 0u, stream);
  carca_set_error("cross_score_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
  return CARCA_ERR_UNSUPPORTED;
}
