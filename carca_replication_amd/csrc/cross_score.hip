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
#include <hip/hip_ext.h>
#include "attn_common.h"
#include "cross_fold.h"
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
        grp.y[(size_t)u * (grp.ldy ? grp.ldy : grp.N) + n] = 1.0f / (1.0f + expf(-logit));
      }
    }
    __syncthreads();
  }
  CA_STAMP(4);
#undef CA_STAMP
}

// ---------------------------------------------------------------------------------------------------------------
// Eval-mode kernel (model.eval(): nothing saved, no dropout, every target attends the whole profile).  Same arithmetic
// contract as the kernel above, restructured:
//   * decoder.ffn is FOLDED into the value projection.  The block's output is the scalar w . (P V + o) (carca.py:340-345),
//     and w_h . (P_h V_h) = P_h u_h with u_h[key] = p[key] . wu[h] + cu[h] (wu, cu: CarcaCaWeights, built at pack
//     time).  V is never formed: phase B projects K (6 feature tiles) and u (ONE tile) instead of K and V (12 tiles),
//     and a (target tile, head) job ends in a per-lane dot of its softmax numerators with u instead of 32 PV MFMAs;
//     the softmax normaliser is applied to that one number instead of to 16 probabilities per lane.
//   * LEADING pad slots are dropped: profiles are left-padded (data.py:113,173), a pad key's weight is an exact 0
//     (carca.py:251-256), so keys are re-based at the first real slot and only ceil(#slots from there / 16) key tiles
//     are projected and scored.  Pads inside the kept range stay masked as before.
//   * masks are ADDED: the score accumulators start from 0 / -1e30 per key (an LDS vector), so the exp2 of a masked
//     score is an exact 0 without a select per score; rows with no allowed key are zeroed by one select.
//   * the first job's target rows are requested before phase A, its W_Q fragments before the barrier ahead of phase C.
// Phase B job = one feature tile x ALL slot tiles of the re-based profile (2 or 4 of them: tiles are projected in
// pairs): the tile's weight fragments are fetched ONCE per workgroup (by the caller, so that the first job's can be
// requested before phase A) and feed that many independent accumulator chains.
//   K:  Ks[16 st + ln][16 ft + 4 mq + r] = sum_k W_K[16 ft + 4 mq + r][k] X[16 st + ln][k] + b_K    (A = W_K, Bt = rows)
//   u:  Ut[h = ln][16 st + 4 mq + r]     = sum_k X[16 st + 4 mq + r][k] wu[h][k] + cu[h]            (A = rows, Bt = wu)
// Tiles beyond the profile are computed and stored like the others (their rows are zeros; phase C scores key tiles in
// pairs too and masks them).
template <int DPI>
struct FoldBW {
  f32x4 wf[DPI / 16];
  f32x4 bk4;  // K: b_K[16 ft + 4 mq ..]
  float cu1;  // u: cu[ln]
  // (both bias forms are fetched and the choice is made where the job runs: a load under a branch, or a select on a
  // loaded value, makes hipcc wait for that load -- and for every load issued before it -- right there)
};
template <int DPI>
__device__ __forceinline__ void fold_b_load(FoldBW<DPI>& w, const float* __restrict__ wk, const float* __restrict__ bk,
                                            const float* __restrict__ wu, const float* __restrict__ cu, int ft, int nf,
                                            int lane, bool dbg16 = false) {
  const bool isu = ft == nf;  // (uniform)
  const float* wp = isu ? wu : wk;
  const int t = isu ? 0 : ft;
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) w.wf[kg] = gload4s(wp, 4 * lane, dbg16 ? 0 : 256 * (t * (DPI / 16) + kg));
  w.cu1 = gload1(cu, lane & 15);
  w.bk4 = gload4s(bk, 4 * (lane >> 4), 16 * t);
}
template <int DPI, int NCH, bool ISU>
__device__ __forceinline__ void fold_b_chains(const FoldBW<DPI>& w, const float* xs, int si, float* Ks, int so, float* Ut,
                                              int ft, int lane, int nh, int c0 = 0) {  // slot tiles c0 .. c0 + NCH - 1
  const int ln = lane & 15, mq = lane >> 4;
  const float* x0 = xs + (16 * c0 + ln) * si + 4 * mq;
  f32x4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = ISU ? f32x4{w.cu1, w.cu1, w.cu1, w.cu1} : w.bk4;
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) {
    f32x4 x[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) x[c] = lds4(x0 + 16 * c * si + 16 * kg);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        acc[c] = ISU ? mfma16(x[c][s], w.wf[kg][s], acc[c]) : mfma16(w.wf[kg][s], x[c][s], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (ISU) {
      if (ln < nh) *reinterpret_cast<f32x4*>(Ut + ln * ATT_SK + 16 * (c0 + c) + 4 * mq) = acc[c];
    } else {
      *reinterpret_cast<f32x4*>(Ks + (16 * (c0 + c) + ln) * so + 16 * ft + 4 * mq) = acc[c];
    }
  }
}
#define FOLD_TPR_S 8  // target tiles staged in LDS per round (STAGE)
#define FOLD_WU_FLOATS(DPI) ((((DPI) / 16 + 3) / 4) * 256)
#define FOLD_UQ 4  // u is computed on the VALU as FOLD_UQ partial sums per (head, slot), added where phase C reads them

// STAGE (the 16-wave workgroups, alone on their CU): W_Q (d <= 96) and the round's target tiles are brought into LDS by
// LDS-DMA requested in the kernel's first instructions -- each byte once per workgroup instead of once per (tile, head)
// job -- and phase C touches global memory only for the targets' ids.  Without it (8-wave workgroups, two or more per
// CU) jobs fetch their operands themselves and the co-resident workgroup covers the latency.
template <int DPI, int DHP, int NH, int NW, bool STAGE>
__global__ __launch_bounds__(NW * 64, 4) void cross_fold_kernel(const FoldArgs a) {
#define CF_STAMP(i)                                                                                  \
  do {                                                                                               \
    if (a.stamps && threadIdx.x == 0 && !(a.dbg & 96)) a.stamps[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
  CF_STAMP(0);
  // diagnostic: per-WAVE stamps instead (dbg bit 32: first instruction of every wave, bit 64: arrival at the A barrier)
  if (a.stamps && (a.dbg & 96) == 32 && (threadIdx.x & 63) == 0) a.stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
  using G = AttGeom<DPI, DHP, NH>;
  constexpr bool STAGE_W = STAGE && DPI <= 96;
  constexpr int TPR = STAGE ? FOLD_TPR_S : CROSS_TPR;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ps = lds;                    // [64][SI]  final-normed profile, re-based at the first real slot
  float* Ks = Ps + ATT_LMAX * G::SI;  // [64][SO]
  float* Ut = Ks + ATT_LMAX * G::SO;  // [FOLD_UQ][NH][ATT_SK]  u^T as partial sums over quarters of the features
  float* Km = Ut + FOLD_UQ * NH * ATT_SK;  // [64] additive key mask: 0 real key, FOLD_NEG pad / beyond the profile; then the slot mask (2 words)
  float* Yp = Km + ATT_LMAX + 4;      // [CROSS_TPR][NH][16] per-head partial logits
  float* Wu = Yp + CROSS_TPR * NH * 16;               // wu as [k group of 4][head (4)][4]: the u tasks' broadcast reads
  float* Ot = Wu + FOLD_WU_FLOATS(DPI);               // STAGE: [TPR][NKG][64 lanes x 4] target tiles, fragment order
  float* Bq = Ot + FOLD_TPR_S * G::NKG * 256;         // STAGE: b_Q [DPO], then decoder.ffn.weight [DPI]
  float* Fw = Bq + 256;
  int* Ids = reinterpret_cast<int*>(Fw + 256);        // STAGE: [TPR * 16] target ids of the round
  float* Wq = Fw + 256 + FOLD_TPR_S * 16;             // STAGE_W: W_Q, fragment order as packed

  const int L = a.L, d = a.d, nparts = a.nparts;
  const int u = blockIdx.x / nparts, part = blockIdx.x - u * nparts;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, mq = lane >> 4;
  const size_t ubase = (size_t)u * L;

  // tiles of this workgroup
  const int all_tiles = a.tile_start[a.ngroups];
  const int per_part = (all_tiles + nparts - 1) / nparts;
  const int t_lo = part * per_part, t_hi = min(all_tiles, t_lo + per_part);
  struct Job {
    int tl, h, gi, qt, n;
    int lrow;            // the lane's target row inside the user's block of the group (clamped into it)
    const float* o;      // the user's block of embedded targets of that group
    const int32_t* ids;
    bool in_range;
  };
  // Group of a PER-LANE tile by selects over the (at most three) groups: indexing a.g[] with a per-lane index makes the
  // compiler fetch the kernel arguments through vector memory, dependent load after dependent load.
  struct Grp {
    int N, ts, ldy;
    const float* o;
    const int32_t* ids;
    float* y;
  };
  auto group_of = [&](int tile) {
    Grp g{a.g[0].N, 0, a.g[0].ldy ? a.g[0].ldy : a.g[0].N, a.g[0].o, a.g[0].ids, a.g[0].y};
#pragma unroll
    for (int i = 1; i < CARCA_MAX_GROUPS; ++i) {
      const bool in = i < a.ngroups && tile >= a.tile_start[i];
      g.N = in ? a.g[i].N : g.N;
      g.ldy = in ? (a.g[i].ldy ? a.g[i].ldy : a.g[i].N) : g.ldy;
      g.ts = in ? a.tile_start[i] : g.ts;
      g.o = in ? a.g[i].o : g.o;
      g.ids = in ? a.g[i].ids : g.ids;
      g.y = in ? a.g[i].y : g.y;
    }
    return g;
  };
  auto decode_tile = [&](int tile, Job& c) {  // (wave-uniform tile: the indexed argument reads are scalar loads)
    c.gi = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
      if (i < a.ngroups && tile >= a.tile_start[i]) c.gi = i;
    c.qt = tile - a.tile_start[c.gi];
    c.n = 16 * c.qt + ln;
    const int N = a.g[c.gi].N;
    c.in_range = c.n < N;
    c.lrow = c.in_range ? c.n : N - 1;
    c.o = a.g[c.gi].o + (size_t)u * N * a.ldo;
    c.ids = a.g[c.gi].ids + (size_t)u * N;
  };
  auto decode = [&](int job, int t0) {
    Job c;
    c.tl = job / NH;
    c.h = job - c.tl * NH;
    decode_tile(t0 + c.tl, c);
    return c;
  };
  auto stage_tiles = [&](int t0, int nt, int w0 = 0, int nw = NW) {  // one DMA per (tile, 16-column group), dealt over
    for (int i = wave - w0; i < nt * G::NKG; i += nw) {                  // the waves w0 .. w0 + nw - 1
      const int tl = i / G::NKG, kg = i - tl * G::NKG;
      Job c;
      decode_tile(t0 + tl, c);
      dma16(c.o, c.lrow * a.ldo + 4 * mq, 16 * kg, Ot + (tl * G::NKG + kg) * 256);
    }
  };
  // target ids of a round, one per thread (STAGE: they travel through LDS like the tiles)
  bool tile_id_ok = false;
  auto load_tile_id = [&](int t0, int nt) {
    const int tile = t0 + min(tid >> 4, nt - 1);
    const Grp g = group_of(tile);
    const int n = 16 * (tile - g.ts) + (tid & 15);
    tile_id_ok = tid < nt * 16 && n < g.N;
    return g.ids[(size_t)u * g.N + min(n, g.N - 1)];  // (unconditional load; masked with tile_id_ok where it is stored)
  };
  const Job job0 = decode(wave, t_lo);  // this wave's first job of the first round, decoded off the critical path
  // Roles (STAGE): the last NDMA waves only REQUEST what phase C reads -- W_Q, the round's target tiles, b_Q, the ffn
  // weight: 62 LDS-DMA instructions, in flight from the kernel's first cycles until the barrier that ends phase B; a CU
  // takes in ~12-20 B per cycle whatever the source, so the sooner the better -- and touch nothing else: a wave that
  // also uses ordinary loads makes hipcc wait vmcnt(0), DMA included, at their first use.  The other NLN waves run the
  // prologue proper.  Without STAGE every wave does.
  constexpr int NDMA = STAGE ? 3 : 0, NLN = NW - NDMA;
  const bool dma_wave = STAGE && wave >= NLN;
  constexpr int PPW = (ATT_LMAX / 2 + NLN - 1) / NLN;  // row pairs per wave
  // Phase B: the K jobs are cut in two (feature tile x HALF of the slot tiles: 2 NF jobs of 48 MFMAs -- three per SIMD at
  // d = 90, on 12 of the staged variant's waves or as 2 + 1 on the 8-wave variant's SIMDs -- instead of NF + 1 jobs of 96,
  // two on three SIMDs and one on the fourth) and run on the first NKW waves.  u = p . wu + cu is VALU work, a slot per
  // lane, beside the MFMAs (as an MFMA job it cost a whole feature tile's MFMAs for NH columns): task (head, quarter of the
  // features) = 6-8 row reads, as many broadcast reads of wu, 12-16 packed FMAs, dealt to the waves whose K jobs end first
  // (U0 ..); the FOLD_UQ partial sums are added where phase C reads them (a fixed order: deterministic).  Wave WUW brings
  // wu into LDS.  (Not the DMA waves' work: hipcc orders an LDS read behind every outstanding LDS-DMA of the wave --
  // measured: they started when their last byte had landed, 5 k cycles into the phase.)
  constexpr int NKW = STAGE ? NLN - 1 : NW;
  constexpr int NKJ = 2 * G::NF;                                   // half jobs
  constexpr int U0 = NKJ % NKW;                                    // first wave with the lighter K load
  constexpr int NUW = (NKW - U0) < 8 ? (NKW - U0) : 8;             // waves that take u tasks
  constexpr int WUW = STAGE ? NKW : NW - 1;                        // the wave that stages wu
  constexpr int PREF_B = STAGE ? G::NF : NKJ;                      // waves whose first half job's weights are prefetched
  static_assert(NH <= 4, "the weight image of u holds four heads");
  static_assert(DPI / 4 % FOLD_UQ == 0, "feature quarters");
  const int half = lane >> 5, c4 = lane & 31;
  const bool col_ok = 4 * c4 < DPI;
  FoldBW<DPI> bw;
  f32x4 qpre[G::NKG];
  int idpre = 0;
  bool have_pre = false;
  unsigned long long pmask = 0;
  int* Pm = reinterpret_cast<int*>(Km + ATT_LMAX);
  auto dma_wq = [&]() {
    if constexpr (STAGE_W)
      if (!(a.dbg & 2))
        for (int c = wave - NLN; c < G::DPO * DPI / 256; c += NDMA) dma16(a.wq, 4 * lane, 256 * c, Wq + 256 * c);
  };
  auto dma_rest = [&]() {
    if constexpr (STAGE) {
      if (!(a.dbg & 4)) stage_tiles(t_lo, min(TPR, t_hi - t_lo), NLN, NDMA);
      if (wave == NW - 1 && lane < G::DPO / 4) dma16(a.bq, 4 * lane, 0, Bq);
      if (wave == NW - 1 && lane < DPI / 4) dma16(a.ffn_w, 4 * lane, 0, Fw);
    }
  };
  if (dma_wave) {
    if (!(a.dbg & 384)) dma_wq();
    if (!(a.dbg & 128)) dma_rest();
  } else {
    // ---- A0: every request of the prologue, none under a branch of its own (behind one, hipcc no longer knows how many
    // loads are in flight and turns the counted wait for the FIRST of them into a wait for nearly all) --------------
    // the profile's ids first; a buffer load like the rest: hipcc counts on in-order return only among loads of one kind
    const int32_t my_id = gload1i(a.p_ids + ubase, lane < L ? lane : L - 1);
    // Rows travel two per wave instruction: lane (half = row of the pair, c4 = 16-byte column group) loads / normalises /
    // stores four contiguous features -- a third of the instructions of a row per wave with a value per lane.
    const float* p_user = a.p_raw + ubase * a.ldp;  // (per-user bases: lane offsets stay small whatever B is)
    f32x4 rv[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      // (a pair beyond the profile is not requested: every load a wave issues is 1 KB through the CU's 64 B/clk return
      // path, wanted or not -- the front of the kernel is bound by that path, stamps in tools/k4_probe.py WAVES=1)
      const int r = 2 * (wave + NLN * j) + half;
      rv[j] = zero4();
      if (2 * (wave + NLN * j) < L) rv[j] = gload4(p_user, (r < L ? r : 0) * a.ldp + (col_ok ? 4 * c4 : 0));
    }
    const f32x4 lnw = gload4(a.ln_w ? a.ln_w : a.bq, col_ok ? 4 * c4 : 0);  // (without a final norm b_Q stands in)
    const f32x4 lnb = gload4(a.ln_w ? a.ln_b : a.bq, col_ok ? 4 * c4 : 0);
    // this wave's phase B job (feature tile `wave`) gets its weight fragments now
    f32x4 wu_pre[(G::NKG + 3) / 4];
    // (only the first round of half jobs prefetches in the front: the second half's waves fetch behind the A barrier, under
    // the first half's MFMAs -- every load issued here lengthens the front by its 1 KB on the return path)
    if (wave < NKW && wave < PREF_B) fold_b_load<DPI>(bw, a.wk, a.bk, a.wu, a.cu, wave % G::NF, G::NF, lane, a.dbg & 16);
    if (wave == WUW) {  // lane (kg' = lane >> 4, mq = (lane >> 2) & 3, h = lane & 3): wu[h][16 kg + 4 mq ..] out of the fragment order
#pragma unroll
      for (int c = 0; c < (G::NKG + 3) / 4; ++c)
        wu_pre[c] = gload4(a.wu, 256 * min(4 * c + (lane >> 4), G::NKG - 1) + 64 * ((lane >> 2) & 3) + 4 * (lane & 3));
    }
    int tile_id0 = 0;
    if constexpr (STAGE) {
      tile_id0 = load_tile_id(t_lo, min(TPR, t_hi - t_lo));
    } else if constexpr (DPI <= 64) {
      // (d > 96: eight fragments per target row and per weight tile; the prefetch would not fit 128 registers)
      // requested by every wave, job or not: a wave without a job fetches job 0's
      const bool pre_ok = wave < min(TPR, t_hi - t_lo) * NH;
      const Job c = decode(pre_ok ? wave : 0, t_lo);
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) qpre[kg] = gload4s(c.o, c.lrow * a.ldo + 4 * mq, 16 * kg);
      idpre = gload1i(c.ids, c.lrow);
      have_pre = pre_ok;
    }
    pmask = __ballot(lane < L && my_id != 0);
    // first real slot; keys are re-based there
    const int s0 = pmask ? (int)__builtin_ctzll(pmask) : L;
    const int nk = L - s0;
    const int LTc = (nk + 15) >> 4;
    CF_STAMP(1);

    // ---- A1: final LayerNorm of the row pairs that hold a slot of the re-based profile, stored at that slot -------------
    const float inv_d = 1.0f / (float)d;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int pr = wave + NLN * j;
      if ((2 * pr + 1 >= s0 || a.p_normed) && 2 * pr < L) {  // (uniform; every row when the normed profile is an output)
        f32x4 v = rv[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (4 * c4 + e < d) ? v[e] : 0.f;
        if (a.ln_w && !(a.dbg & 8)) {
          const float mean = half32_sum((v[0] + v[1]) + (v[2] + v[3])) * inv_d;
          f32x4 dv;
#pragma unroll
          for (int e = 0; e < 4; ++e) dv[e] = (4 * c4 + e < d) ? v[e] - mean : 0.f;
          const float var = half32_sum((dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3])) * inv_d;
          const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (4 * c4 + e < d) ? dv[e] * rstd * lnw[e] + lnb[e] : 0.f;
        }
        rv[j] = v;
      }
    }
    CF_STAMP(6);
    if constexpr (STAGE)
      if (tid < TPR * 16) Ids[tid] = tile_id_ok ? tile_id0 : 0;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int r = 2 * (wave + NLN * j) + half, t = r - s0;
      if (col_ok && r < L && t >= 0) *reinterpret_cast<f32x4*>(Ps + t * G::SI + 4 * c4) = rv[j];
      if (a.p_normed && part == 0 && r < L && 4 * c4 < a.ldp) *reinterpret_cast<f32x4*>(a.p_normed + (ubase + r) * a.ldp + 4 * c4) = rv[j];
    }
    // rows of the (even number of) key tiles phase B reads beyond the profile: zeros, not LDS garbage
    for (int t = nk + 2 * wave + half; t < min(ATT_LMAX, 32 * ((LTc + 1) >> 1)); t += 2 * NLN)
      if (col_ok) *reinterpret_cast<f32x4*>(Ps + t * G::SI + 4 * c4) = zero4();
    if (wave == NLN - 1) Km[lane] = (lane < nk && ((pmask >> (lane + s0)) & 1ull)) ? 0.f : FOLD_NEG;
    if (wave == WUW) {
#pragma unroll
      for (int c = 0; c < (G::NKG + 3) / 4; ++c) *reinterpret_cast<f32x4*>(Wu + 256 * c + 4 * lane) = wu_pre[c];
    }
    if (STAGE && tid == 0) {  // the slot mask, for the waves that did not see the ids
      Pm[0] = (int)(unsigned)pmask;
      Pm[1] = (int)(unsigned)(pmask >> 32);
    }
    CF_STAMP(12);
  }
  // LDS-only barrier, hand-written: the DMA stays in flight across it.  (__syncthreads(), and even a workgroup fence
  // restricted to LDS, make hipcc wait vmcnt(0) here: an outstanding LDS-DMA counts as a pending LDS write.  Nothing the
  // DMA writes is read before the __syncthreads() that ends phase B.)
  if (a.stamps && (a.dbg & 96) == 64 && (threadIdx.x & 63) == 0) a.stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  CF_STAMP(2);
  if (dma_wave) {
    if (a.dbg & 384) dma_wq();
    if (a.dbg & 128) dma_rest();
  }
  if (dma_wave)
    pmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(Pm[1]) << 32) |
            (unsigned)__builtin_amdgcn_readfirstlane(Pm[0]);
  const int s0 = pmask ? (int)__builtin_ctzll(pmask) : L;
  const int nk = L - s0;
  const int LTc = (nk + 15) >> 4;
  // ---- B: K tiles and the u tile, every slot tile in one job ------------------------------------------------------------
  {
    const int npair = (LTc + 1) >> 1;
    if (wave < NKW) {
      bool first = wave < PREF_B;
      for (int job = wave; job < NKJ && npair > 0; job += NKW) {
        const int ft = job % G::NF, hf = job / G::NF;
        if (!first) fold_b_load<DPI>(bw, a.wk, a.bk, a.wu, a.cu, ft, G::NF, lane, a.dbg & 16);
        first = false;
        if (npair > 1) fold_b_chains<DPI, 2, false>(bw, Ps, G::SI, Ks, G::SO, Ut, ft, lane, NH, 2 * hf);
        else fold_b_chains<DPI, 1, false>(bw, Ps, G::SI, Ks, G::SO, Ut, ft, lane, NH, hf);
      }
      if (wave >= U0 && wave < U0 + NUW && npair > 0) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int KQ = DPI / 4 / FOLD_UQ;  // 16-byte feature groups per quarter
        const float* xr = Ps + lane * G::SI;
        for (int task = wave - U0; task < FOLD_UQ * NH; task += NUW) {
          const int h = task % NH, q = task / NH;
          f32x4 xv[KQ], wv[KQ];
#pragma unroll
          for (int i = 0; i < KQ; ++i) xv[i] = lds4(xr + 4 * (q * KQ + i));
#pragma unroll
          for (int i = 0; i < KQ; ++i) wv[i] = lds4(Wu + 16 * (q * KQ + i) + 4 * h);
          f32x2 acc = f32x2{q == 0 ? a.cu[h] : 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < KQ; ++i) {
            acc = f32x2{xv[i][0], xv[i][1]} * f32x2{wv[i][0], wv[i][1]} + acc;
            acc = f32x2{xv[i][2], xv[i][3]} * f32x2{wv[i][2], wv[i][3]} + acc;
          }
          Ut[task * ATT_SK + lane] = lane < 32 * npair ? acc[0] + acc[1] : 0.f;  // (task = q NH + h)
        }
      }
    }
  }
  CF_STAMP(5);
  // (dbg bits 32 + 64: every wave's clock when its phase B work is done)
  if (a.stamps && (a.dbg & 96) == 96 && (threadIdx.x & 63) == 0) a.stamps[blockIdx.x * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();
  __syncthreads();
  CF_STAMP(3);

  // ---- C: rounds of TPR target tiles; job = (tile, head) ----------------------------------------------------------------
  const float ffn_b = a.ffn_b[0];
  for (int t0 = t_lo; t0 < t_hi; t0 += TPR) {
    const int nt = min(TPR, t_hi - t0);
    if (STAGE && t0 != t_lo) {  // (the first round's tiles were requested in the prologue)
      stage_tiles(t0, nt);
      const int id = load_tile_id(t0, nt);
      if (tid < TPR * 16) Ids[tid] = tile_id_ok ? id : 0;
      __syncthreads();
    }
    // (the job body is instantiated twice: once for the prefetched first job, once for the loop -- with one copy the
    // prefetched fragments would stay live through every iteration and spill)
    // wpre / bpre: the head's W_Q fragments and bias when the caller fetched them (the pipelined path below), else null;
    // after_proj(): called once the projection has consumed qfrag / wpre (the caller overwrites them with the next job's)
    auto run_job = [&](const Job& c, const f32x4 (&qfrag)[G::NKG], const f32x4 (*wpre)[G::NKG], const f32x4* bpre, int tgt_id,
                       auto after_proj) __attribute__((always_inline)) {
      const int h = c.h;
      const bool q_ok = c.in_range && tgt_id != 0;
      const int nkt = LTc;  // eval mode: every target sees all of the re-based profile (carca.py:339: causal = None)
      CF_STAMP(7);
      // Q^T tiles of the head, scaled into the exp2 domain
      f32x4 qt[G::NFH];
      if constexpr (STAGE_W && G::NFH == 2) {  // both feature tiles' chains interleaved, fragments from LDS
        const float* w0 = Wq + (h * 2 * G::NKG) * 256 + 4 * lane;
        f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) {
          const f32x4 a0 = lds4(w0 + kg * 256), a1 = lds4(w0 + (G::NKG + kg) * 256);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0 = mfma16(a0[s], qfrag[kg][s], acc0);
            acc1 = mfma16(a1[s], qfrag[kg][s], acc1);
          }
        }
        qt[0] = (acc0 + lds4(Bq + h * DHP + 4 * mq)) * a.qscale;
        qt[1] = (acc1 + lds4(Bq + h * DHP + 16 + 4 * mq)) * a.qscale;
      } else if (wpre) {  // the first feature tile's fragments arrived with the target rows; the others are requested
                          // now and land under the first tile's chain
        f32x4 wr[G::NFH > 1 ? G::NFH - 1 : 1][G::NKG];
#pragma unroll
        for (int ft = 1; ft < G::NFH; ++ft)
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg)
            wr[ft - 1][kg] = gload4s(a.wq, 4 * lane, (a.dbg & 2) ? 0 : 256 * ((h * G::NFH + ft) * G::NKG + kg));
        CARCA_PIN_LOADS();
#pragma unroll
        for (int ft = 0; ft < G::NFH; ++ft) {
          f32x4 acc = zero4();
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(ft == 0 ? wpre[0][kg] : wr[ft > 0 ? ft - 1 : 0][kg], qfrag[kg], acc);
          qt[ft] = (acc + bpre[ft]) * a.qscale;
        }
      } else {
#pragma unroll
        for (int ft = 0; ft < G::NFH; ++ft) {
          f32x4 wf[G::NKG];
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            if constexpr (STAGE_W) wf[kg] = lds4(Wq + ((h * G::NFH + ft) * G::NKG + kg) * 256 + 4 * lane);
            else wf[kg] = gload4s(a.wq, 4 * lane, (a.dbg & 2) ? 0 : 256 * ((h * G::NFH + ft) * G::NKG + kg));
          }
          const f32x4 bias = STAGE ? lds4(Bq + h * DHP + 16 * ft + 4 * mq) : gload4s(a.bq, 4 * mq, h * DHP + 16 * ft);
          if constexpr (!STAGE_W) CARCA_PIN_LOADS();
          f32x4 acc = zero4();
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(wf[kg], qfrag[kg], acc);
          qt[ft] = (acc + bias) * a.qscale;  // (bias after the chain in every variant: same bits whichever runs)
        }
      }
      CF_STAMP(8);
      // residual part of the logit (w . o, once per target) behind the projection's MFMAs
      float ypart = 0.f;
      if (a.residual && h == 0) {
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) {
          const f32x4 wv = STAGE ? lds4(Fw + 16 * kg + 4 * mq) : gload4s(a.ffn_w, 4 * mq, 16 * kg);
#pragma unroll
          for (int r = 0; r < 4; ++r) ypart += wv[r] * qfrag[kg][r];
        }
        ypart = quad4_sum(ypart);
      }
      after_proj();
      // scores^T tiles (rows = keys, cols = targets), accumulated on top of the additive mask; two key tiles at a time
      // (two independent accumulator chains; the second tile of a pair exists in LDS even beyond nkt: masked there)
      f32x4 sc[ATT_LT];
      float mx = FOLD_NEG;
#pragma unroll
      for (int kp = 0; kp < ATT_LT / 2; ++kp) {
        if (2 * kp < nkt) {
          f32x4 acc0 = lds4(Km + 32 * kp + 4 * mq), acc1 = lds4(Km + 32 * kp + 16 + 4 * mq);
          const float* krow = Ks + (32 * kp + ln) * G::SO + h * DHP + 4 * mq;
#pragma unroll
          for (int ft = 0; ft < G::NFH; ++ft) {
            const f32x4 k0 = lds4(krow + 16 * ft), k1 = lds4(krow + 16 * G::SO + 16 * ft);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
              acc0 = mfma16(k0[s_], qt[ft][s_], acc0);
              acc1 = mfma16(k1[s_], qt[ft][s_], acc1);
            }
          }
          sc[2 * kp] = acc0;
          sc[2 * kp + 1] = acc1;
          mx = fmaxf(fmaxf(mx, fmaxf(acc0[0], acc0[1])), fmaxf(acc0[2], acc0[3]));
          mx = fmaxf(fmaxf(mx, fmaxf(acc1[0], acc1[1])), fmaxf(acc1[2], acc1[3]));
        }
      }
      CF_STAMP(9);
      mx = quad4_max(mx);
      float sum = 0.f, dot = 0.f;
#pragma unroll
      for (int kp = 0; kp < ATT_LT / 2; ++kp) {
        if (2 * kp < nkt) {
#pragma unroll
          for (int kt = 2 * kp; kt < 2 * kp + 2; ++kt) {
            f32x4 uv = lds4(Ut + h * ATT_SK + 16 * kt + 4 * mq);
#pragma unroll
            for (int q = 1; q < FOLD_UQ; ++q) uv = uv + lds4(Ut + (q * NH + h) * ATT_SK + 16 * kt + 4 * mq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __builtin_amdgcn_exp2f(sc[kt][r] - mx);
              sum += e;
              dot += e * uv[r];
            }
          }
        }
      }
      sum = quad4_sum(sum);
      dot = quad4_sum(dot);
      // a target with no allowed key (pad target, first slot when training, all-pad profile) attends nothing: exact 0
      const float attn = (q_ok && mx > 0.5f * FOLD_NEG) ? dot / sum : 0.f;
      if (mq == 0) Yp[(c.tl * NH + h) * 16 + ln] = attn + ypart;
      CF_STAMP(10);
    };
    constexpr bool PIPE = !STAGE && G::NFH * G::NKG <= 16;
    if constexpr (PIPE) {
      // Software pipeline over this wave's jobs: the operands of job j+1 (target rows, the head's first W_Q tile, bias,
      // target id) are requested as soon as job j's projection has consumed its own, into the same registers, and
      // land under job j's scores and softmax -- a job then starts its first chain without waiting for memory.
      struct Ops {
        f32x4 q[G::NKG], w[1][G::NKG], b[G::NFH];  // (w: the head's FIRST feature tile; 128 registers do not hold more)
        int id;
      } cur;
      auto load_w = [&](const Job& c) {
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg)
          cur.w[0][kg] = gload4s(a.wq, 4 * lane, (a.dbg & 2) ? 0 : 256 * ((c.h * G::NFH) * G::NKG + kg));
#pragma unroll
        for (int ft = 0; ft < G::NFH; ++ft) cur.b[ft] = gload4s(a.bq, 4 * mq, c.h * DHP + 16 * ft);
      };
      auto load_q = [&](const Job& c) {
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) cur.q[kg] = gload4s(c.o, (a.dbg & 4) ? 0 : c.lrow * a.ldo + 4 * mq, 16 * kg);
        cur.id = gload1i(c.ids, c.lrow);
      };
      const int njobs = nt * NH;
      if (wave < njobs) {
        Job cj = decode(wave, t0);
        if (have_pre) {
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) cur.q[kg] = qpre[kg];
          cur.id = idpre;
          have_pre = false;
        } else {
          load_q(cj);
        }
        load_w(cj);
        for (int job = wave; job < njobs; job += NW) {
          Job cn;
          const int tgt_id = cur.id;
          run_job(cj, cur.q, cur.w, cur.b, tgt_id, [&]() {
            const int nj = job + NW;
            cn = decode(nj < njobs ? nj : job, t0);  // (the last job re-requests its own operands: no load under a branch)
            load_q(cn);
            load_w(cn);
          });
          cj = cn;
        }
      }
    } else {
      int job = wave;
      if (have_pre) {
        run_job(STAGE ? job0 : decode(job, t0), qpre, nullptr, nullptr, idpre, []() {});
        have_pre = false;
        job += NW;
      }
      for (; job < nt * NH; job += NW) {
        const Job c = (STAGE && t0 == t_lo && job == wave) ? job0 : decode(job, t0);
        f32x4 qfrag[G::NKG];
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) {
          if constexpr (STAGE) qfrag[kg] = lds4(Ot + (c.tl * G::NKG + kg) * 256 + 4 * lane);
          else qfrag[kg] = gload4s(c.o, (a.dbg & 4) ? 0 : c.lrow * a.ldo + 4 * mq, 16 * kg);
        }
        run_job(c, qfrag, nullptr, nullptr, STAGE ? Ids[c.tl * 16 + ln] : c.ids[c.lrow], []() {});
      }
    }
    __syncthreads();
    CF_STAMP(11);
    if (tid < nt * 16) {
      const int tl = tid >> 4, l16 = tid & 15;
      const Grp g = group_of(t0 + tl);
      const int n = 16 * (t0 + tl - g.ts) + l16;
      if (n < g.N) {
        float logit = ffn_b;
#pragma unroll
        for (int h = 0; h < NH; ++h) logit += Yp[(tl * NH + h) * 16 + l16];
        g.y[(size_t)u * g.ldy + n] = 1.0f / (1.0f + expf(-logit));
      }
    }
    if (t0 + TPR < t_hi) __syncthreads();
  }
  CF_STAMP(4);
#undef CF_STAMP
}

template <int DPI, int DHP, int NH, bool STAGE>
constexpr size_t fold_lds_bytes() {
  using G = AttGeom<DPI, DHP, NH>;
  size_t f = ATT_LMAX * G::SI + ATT_LMAX * G::SO + FOLD_UQ * NH * ATT_SK + ATT_LMAX + 4 + CROSS_TPR * NH * 16 + FOLD_WU_FLOATS(DPI);
  if (STAGE) f += FOLD_TPR_S * G::NKG * 256 + 512 + FOLD_TPR_S * 16 + (DPI <= 96 ? G::DPO * DPI : 0);
  return sizeof(float) * f;
}

template <int DPI, int DHP, int NH, int NW, bool STAGE>
int launch_fold_nw(const FoldArgs& fa, int B, hipStream_t stream) {
  constexpr size_t lds_bytes = fold_lds_bytes<DPI, DHP, NH, STAGE>();
  static_assert(lds_bytes <= 160 * 1024, "the staged variant must fit one CU's LDS");
  auto kern = cross_fold_kernel<DPI, DHP, NH, NW, STAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("cross_score_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))  // (timing events bound to this dispatch: carca_forward's ev[2], ev[3])
    hipExtLaunchKernelGGL(kern, dim3(B * fa.nparts), dim3(NW * 64), lds_bytes, stream, e0, e1, 0, fa);
  else
    hipLaunchKernelGGL(kern, dim3(B * fa.nparts), dim3(NW * 64), lds_bytes, stream, fa);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

// Variant choice (tuning key 1 as for the kernel above): two 16-wave workgroups per user while 2 B <= #CUs (latency
// regime: each takes half the target tiles), one per user up to #CUs -- both with LDS staging; beyond #CUs one 8-wave
// workgroup per user, two resident per CU.  Tuning key 7 = 1: 16-wave workgroups without staging (A/B).
template <int DPI, int DHP, int NH>
int launch_fold(FoldArgs& fa, int B, hipStream_t stream) {
  const int num_cus = carca_num_cus();
  const int tune = carca_tuning(CARCA_TUNE_ATTN_VARIANT);
  const int all_tiles = fa.tile_start[fa.ngroups];
  fa.nparts = (all_tiles > 1 && tune != 1 && tune != 3 && (tune == 2 || 2 * B <= num_cus)) ? 2 : 1;
  // (Measured and dropped: delaying the first occupant of every CU's second slot by 8-32 k cycles, against the idea that
  // two workgroups started together stay in lockstep -- the launch got longer by exactly the delay.)
  if (fa.nparts == 1 && (tune == 3 || (tune == 0 && B > num_cus)))
    return launch_fold_nw<DPI, DHP, NH, 8, false>(fa, B, stream);
  if (carca_tuning(7) == 1) return launch_fold_nw<DPI, DHP, NH, 16, false>(fa, B, stream);
  return launch_fold_nw<DPI, DHP, NH, 16, true>(fa, B, stream);
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
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))  // (timing events bound to this dispatch: carca_forward's ev[2], ev[3])
    hipExtLaunchKernelGGL(kern, dim3(B * nparts), dim3(NW * 64), lds_bytes, stream, e0, e1, 0, p_raw, ldp, p_ids, p_normed,
                          groups, ldo, L, d, d / NH, w, residual, training, sv, dc, site, nparts, carca_debug_buffer());
  else
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
    CARCA_CHECK_ARG(groups[i].o && groups[i].ids && groups[i].y && groups[i].N >= 1 &&
                        (groups[i].ldy == 0 || groups[i].ldy >= groups[i].N),
                    "cross_score_fwd: group %d malformed", i);
    gd.g[i] = groups[i];
    gd.tile_start[i] = t;
    t += (groups[i].N + 15) / 16;
  }
  gd.tile_start[ngroups] = t;
  gd.n = ngroups;
  CARCA_CHECK_ARG(!(drop && drop->p >= 1.0f), "cross_score_fwd: dropout p must be < 1");
  // eval mode (nothing saved, no dropout, no causal mask): the folded kernel (tuning key 6 = 1 forces the other one)
  if (!save && !training && w->wu && w->cu && carca_tuning(6) != 1 && ldp % 4 == 0 && ldp >= dpi) {
    FoldArgs fa{};
    fa.p_raw = p_raw; fa.p_ids = p_ids; fa.p_normed = p_normed;
    for (int i = 0; i < ngroups; ++i) fa.g[i] = gd.g[i];
    for (int i = 0; i <= ngroups; ++i) fa.tile_start[i] = gd.tile_start[i];
    fa.ngroups = ngroups; fa.ldp = ldp; fa.ldo = ldo; fa.L = L; fa.d = d; fa.residual = residual;
    fa.ln_w = w->ln_w; fa.ln_b = w->ln_b; fa.wq = w->wq; fa.bq = w->bq; fa.wk = w->wk; fa.bk = w->bk; fa.wu = w->wu;
    fa.cu = w->cu; fa.ffn_w = w->ffn_w; fa.ffn_b = w->ffn_b;
    fa.qscale = (float)(1.4426950408889634 / sqrt((double)(d / H)));
    fa.stamps = carca_debug_buffer();
    fa.dbg = carca_tuning(CARCA_TUNE_DIAG);
    fa.B = B;
    fa.opt = carca_tuning(CARCA_TUNE_XS_OPT);
    // Persistent workgroups that pipeline their units of work (cross_stream.hip): every batch size where they are not
    // slower -- tuning key 7: 2 = never, 3 = at every batch size, 0 = above #CUs users.  It has no p_normed output and
    // no instantiation above d = 96: the per-user kernel below then runs.
    const int t7 = carca_tuning(7);
    const int tune1 = carca_tuning(CARCA_TUNE_ATTN_VARIANT);
    // (... and from 48 target tiles per user on -- C5's 1 + 1000 candidates are 63 --: the per-user kernel fetches W_Q and the
    // target rows per (tile, head) job through the CU's load path, 53.4 us at C5 against 47.7 here; at 301 candidates 24.1
    // against 23.1, at C2's 101 14.3 against 14.7)
    if (!p_normed && t7 != 2 && (tune1 == 0 || tune1 == 2) &&
        (t7 == 3 || B > carca_num_cus() || gd.tile_start[ngroups] >= 48)) {
      fa.nparts = (gd.tile_start[ngroups] > 1 && (tune1 == 2 || 2 * B <= carca_num_cus())) ? 2 : 1;
      const int rc = carca_cross_stream_launch(fa, dpi, dhp, H, B, stream);
      if (rc != CARCA_ERR_UNSUPPORTED) return rc;
    }
    CARCA_ATT_DISPATCH(launch_fold, fa, B, stream);
    carca_set_error("cross_score_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
    return CARCA_ERR_UNSUPPORTED;
  }
  CarcaCaSave sv{};
  if (save) sv = *save;
  const DropCfg dc = make_drop(drop);
  CARCA_ATT_DISPATCH(launch_cross, p_raw, ldp, p_ids, p_normed, gd, ldo, B, L, d, *w, residual, training, sv, dc,
                     drop ? drop->site : 0u, stream);
  carca_set_error("cross_score_fwd: no kernel built for d=%d H=%d (padded %d / head %d)", d, H, dpi, dhp);
  return CARCA_ERR_UNSUPPORTED;
}
