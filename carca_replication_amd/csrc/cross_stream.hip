// K4 for batches of MANY users per CU (B > #CUs): final LayerNorm (carca.py:421) + CrossAttentionBlock.forward in eval mode
// (carca.py:338-349, causal = None, decoder.ffn folded into the value projection as in cross_fold_kernel) as PERSISTENT
// 16-wave workgroups, one per CU, that pipeline their users.
//
// Why another kernel.  Above #CUs users cross_fold_kernel ran as 8-wave workgroups, two per CU, each (target tile, head)
// job fetching its head's W_Q tile (12 KB at d = 90) and its target rows (6 KB) from L2: 21 jobs x 18 KB + W_K 36 KB +
// the profile = ~430 KB per user through a CU's load-return path, which delivers 12-20 B per cycle whatever the source
// (TUNING.md) -- ~27 k cycles per user against an MFMA floor of 18 k, and the phases of a user (LayerNorm, K
// projection, jobs) are separated by workgroup barriers that idle three of four SIMD slots in turn.  Here
//   * W_Q and W_K (fragment order, 2 x 36 KB at d = 90) are brought into LDS ONCE per workgroup and stay there for all of
//     its users: the only per-user global traffic is the profile rows (once) and the target rows;
//   * the waves have fixed roles and meet at ONE barrier per (user, round of 8 target tiles):
//       12 "C" waves run the (target tile, head) jobs of user k out of the K / u / mask images of buffer k & 1 --
//          Q^T projection (W_Q fragments from LDS), scores^T on top of the additive key mask, exp2 softmax, folded dot;
//          the next job's target rows are requested as soon as the projection has consumed the current ones;
//        4 "B" waves build user k + 1's images in buffer (k + 1) & 1 meanwhile: B wave st owns slot tile st of the
//          re-based profile -- its 16 rows come from HBM straight in the operand layout (lane (slot, mq) holds columns
//          16 kg + 4 mq ..), the final LayerNorm is a per-lane sum + two lane swaps, and the rows then feed all NF
//          feature tiles of the K projection as the Bt operand out of registers (NF independent accumulator chains,
//          W_K fragments from LDS), the folded value u = p . wu + cu and the additive mask;
//     two of the B waves also turn the previous round's per-head partial logits into sigmoid outputs.
//   Every SIMD then holds three C waves and one B wave whose MFMA work overlaps the others' softmax / LayerNorm VALU
//   work, and a user costs its MFMA time (C 1680 + B 576 MFMAs at C2 = 18 k cycles per CU) plus what the issue
//   order leaves idle, instead of the sum of its phases.
// Same arithmetic contract as cross_fold_kernel: identical products on the exact-fp32 MFMA; the LayerNorm row sums and
// the u dot products are grouped differently (per lane, then across the four lanes of a slot), so results agree to
// round-off (tests/test_hip_forward.py compares both with the oracle and with each other).
#include <hip/hip_ext.h>
#include <type_traits>
#include "attn_common.h"
#include "cross_fold.h"
#include "../../include/carca_hip.h"

namespace {

// A wave-uniform condition that must stay a BRANCH: hipcc turns a short conditional block into selects executed by
// everyone (the LayerNorm's column / row masks came out as 242 v_cndmask per tile), and every fp32 VALU instruction of a
// SIMD takes its issue slot from the MFMAs.  An (empty) asm statement keeps the block from being if-converted.
#define XS_RARE(cond) (__builtin_expect((cond), 0) && ([] { asm volatile(""); return true; }()))
#define XS_TPR 8   // target tiles per round
#define XS_NCW 12  // C waves (jobs)
#define XS_NBW 4   // B waves (one slot tile each)
#define XS_NW (XS_NCW + XS_NBW)

template <int DPI, int DHP, int NH>
struct XsLds {  // offsets in floats; every region starts on a 16-byte boundary
  using G = AttGeom<DPI, DHP, NH>;
  static constexpr int WK = 0;                                 // [DPO x DPI] fragment order
  static constexpr int WQ = WK + G::DPO * DPI;                 // [DPO x DPI] fragment order
  static constexpr int KS = WQ + G::DPO * DPI;                 // [2][64][SO]
  static constexpr int UT = KS + 2 * ATT_LMAX * G::SO;         // [2][NH][ATT_SK]
  static constexpr int KM = UT + 2 * NH * ATT_SK;              // [2][64] additive key mask
  static constexpr int HDR = KM + 2 * ATT_LMAX;                // [2][4] ints: key tiles of the user; [8..9]: job tickets
  static constexpr int YP = HDR + 16;                          // [2][XS_TPR][NH][16] per-head partial logits
  static constexpr int WU = YP + 2 * XS_TPR * NH * 16;         // [16 x DPI] fragment order (rows >= NH zero)
  static constexpr int LNW = WU + 16 * DPI;                    // final norm weight, bias
  static constexpr int LNB = LNW + DPI;
  static constexpr int BQ = LNB + DPI;
  static constexpr int BK = BQ + G::DPO;
  static constexpr int FW = BK + G::DPO;                       // decoder.ffn.weight, plain order
  static constexpr int CU = FW + DPI;                          // [16]
  static constexpr int XS = CU + 16;                           // [4][DPI] normalised rows of a nearly empty last slot tile
  static constexpr int TOTAL = XS + 4 * DPI;
};

template <int DPI, int DHP, int NH>
__global__ __launch_bounds__(XS_NW * 64) void cross_stream_kernel(const FoldArgs a) {
  using G = AttGeom<DPI, DHP, NH>;
  using M = XsLds<DPI, DHP, NH>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Wk = lds + M::WK;
  float* const Wq = lds + M::WQ;
  float* const Ks2 = lds + M::KS;
  float* const Ut2 = lds + M::UT;
  float* const Km2 = lds + M::KM;
  int* const Hdr = reinterpret_cast<int*>(lds + M::HDR);
  int* const Cnt = Hdr + 8;  // [2] next job ticket of the step, by step parity
  float* const Yp2 = lds + M::YP;
  float* const Wu = lds + M::WU;
  float* const Lnw = lds + M::LNW;
  float* const Lnb = lds + M::LNB;
  float* const Bq = lds + M::BQ;
  float* const Bk = lds + M::BK;
  float* const Fw = lds + M::FW;
  float* const Cu = lds + M::CU;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, mq = lane >> 4;
  const int L = a.L, d = a.d;
  // A workgroup's UNITS: unit w = (user w / nparts, part w % nparts of that user's target tiles), w = blockIdx.x,
  // blockIdx.x + gridDim.x, ...  nparts = 1 when users outnumber the CUs; 2 while 2 B <= #CUs (the latency regime: both
  // workgroups of a user build the same K / u images and score half of the tiles each, nothing passes between them).
  const int v0 = blockIdx.x, vs = gridDim.x;
  const int nparts = a.nparts;
  const int all_tiles = a.tile_start[a.ngroups];
  const int per_part = (all_tiles + nparts - 1) / nparts;
  const int nu = (a.B * nparts - v0 + vs - 1) / vs;  // units of this workgroup (>= 1: the grid never exceeds their number)
  const int R = (per_part + XS_TPR - 1) / XS_TPR;    // rounds per unit
  auto user_of = [&](int k) { return (v0 + k * vs) / nparts; };
  auto tlo_of = [&](int k) { return ((v0 + k * vs) % nparts) * per_part; };
  const bool ln_on = a.ln_w != nullptr;
#define XS_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (a.stamps && threadIdx.x == 0) a.stamps[blockIdx.x * 64 + 48 + (i)] = __builtin_readcyclecounter();  \
  } while (0)
  XS_STAMP(0);

  // ---- weights -> LDS, once per workgroup (LDS-DMA, 1 KB per wave instruction).  What the B waves need first -- W_K, wu, the
  // norm vectors -- is requested by all sixteen waves here; W_Q and the other C-side vectors by the C waves behind the
  // first barrier, so that they land under the B waves' first build instead of in front of it.
  constexpr int NCH = G::DPO * DPI / 256;
  {
    for (int c = wave; c < NCH; c += XS_NW) dma16(a.wk, 4 * lane, 256 * c, Wk + 256 * c);
    for (int c = wave; c < G::NKG; c += XS_NW) dma16(a.wu, 4 * lane, 256 * c, Wu + 256 * c);
    if (wave == 6 && ln_on && lane < DPI / 4) dma16(a.ln_w, 4 * lane, 0, Lnw);
    if (wave == 7 && ln_on && lane < DPI / 4) dma16(a.ln_b, 4 * lane, 0, Lnb);
    if (wave == 9 && lane < G::DPO / 4) dma16(a.bk, 4 * lane, 0, Bk);
    if (wave == 11 && lane < 4) dma16(a.cu, 4 * lane, 0, Cu);
  }

  // groups of target tiles (as in cross_fold_kernel): a wave-uniform tile's group through scalar reads of the arguments
  struct Job {
    int lrow;        // the lane's target row inside the user's block of the group (clamped into it)
    const float* o;  // the user's block of embedded targets of that group
    const int32_t* ids;
    bool in_range;
  };
  auto decode_tile = [&](int tile, int v, Job& c) {
    int gi = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
      if (i < a.ngroups && tile >= a.tile_start[i]) gi = i;
    const int n = 16 * (tile - a.tile_start[gi]) + ln;
    const int N = a.g[gi].N;
    c.in_range = n < N;
    c.lrow = c.in_range ? n : N - 1;
    c.o = a.g[gi].o + (size_t)v * N * a.ldo;
    c.ids = a.g[gi].ids + (size_t)v * N;
  };
  auto njobs_of = [&](int k, int rd) {
    const int t_lo = tlo_of(k);
    return max(0, min(XS_TPR, min(all_tiles, t_lo + per_part) - t_lo - rd * XS_TPR)) * NH;
  };
  // LDS-only barrier: global loads (the prefetched operands of the next job / user) stay in flight across it
#define XS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // diagnostic (debug buffer set, tools/k4_stream_check.py STAMPS=1): per-wave clocks of step a.dbg of this workgroup --
  // [wave] behind the barrier that opens it, [16 + wave] on arrival at the barrier that closes it, [32 + wave] job / build done
#define XS_WSTAMP(s_, slot)                                                                                         \
  do {                                                                                                              \
    if (a.stamps && (s_) == a.dbg && lane == 0) a.stamps[blockIdx.x * 64 + (slot) + wave] = __builtin_readcyclecounter(); \
  } while (0)

  if (wave >= XS_NCW) {
    // =================================================== B waves ==========================================================
    // This wave's slot tile of the re-based profile, dealt in reverse: wave 12 shares its SIMD with the C waves 0, 4, 8, the
    // three that always hold two jobs; it takes the last slot tile, the first one a short profile does not have.
    const int st = XS_NW - 1 - wave;
    // The SIMD serves its waves by priority, then age, and a wave issues in order: left at priority 0 the B wave -- the
    // youngest of its SIMD -- ran BEHIND the three C waves (per-wave stamps: the C waves reached the step's barrier after
    // 17-20 k cycles, the B waves after 26-29 k, their 144 MFMAs issued alone at the end).  Ahead of them its MFMAs and
    // LayerNorm arithmetic interleave with the C waves' and the step ends with the jobs.
    __builtin_amdgcn_s_setprio(3);
    const float ffn_b = a.ffn_b[0];
    f32x4 xr[G::NKG];              // the tile's rows as requested (raw encoder output)
    unsigned long long pmask;
    auto request_ids = [&](int k) {
      const int v = user_of(min(k, nu - 1));
      return gload1i(a.p_ids + (size_t)v * L, lane < L ? lane : L - 1);
    };
    // A ONE-TILE profile (5..16 slots: 29 % of BASELINE's draws) leaves three of the four B waves idle and its whole build --
    // 144 MFMAs + the LayerNorm -- on the SIMD of the fourth, beside five of the user's jobs: that SIMD sets the step (18.2 k
    // cycles against a mean of 14.2 k).  The waves of slot tiles 0, 1, 2 then share tile 0: each loads and normalises its rows
    // (cheap) and projects a THIRD of the K features; wave st = 0 alone writes u and the mask.  (NF % 3 == 0: d = 90 / 96;
    // a.opt bit 2 -- tuning key 14 -- switches it off.)  Measured (tools/k4_length_probe.py, B = 4096, every profile 16 / 8 slots):
    // 140.6 -> 134.9 / 125.6 us; with BASELINE's mixed lengths the launch reads the same (0.79): a short profile's jobs are
    // short, and their steps wait for the next job's target rows, not for the build.
    constexpr bool HELP = (G::NF % 3) == 0;
    auto shares_tile0 = [&](unsigned long long pm) {
      const int nk_ = pm ? L - (int)__builtin_ctzll(pm) : 0;
      return HELP && !(a.opt & 4) && st < 3 && nk_ >= 1 && nk_ <= 16;  // (1..4 slots too: a third of the tile each beats the 4 x 4 x 1 path on one SIMD, 126 against 142 us)
    };
    auto request_rows = [&](int k, unsigned long long pm) {
      const int v = user_of(min(k, nu - 1));
      const int s0 = pm ? (int)__builtin_ctzll(pm) : L;
      const int r = s0 + 16 * (shares_tile0(pm) ? 0 : st) + ln;
      const float* pu = a.p_raw + (size_t)v * L * a.ldp;
      const int ro = (r < L ? r : L - 1) * a.ldp + 4 * mq;
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) xr[kg] = gload4s(pu, ro, 16 * kg);
    };
    // user k's images into buffer k & 1 from xr / pmask; then the requests for user k + 1
    auto build = [&](int k) {
      const int buf = k & 1;
      const int id_next = request_ids(k + 1);
      const int s0 = pmask ? (int)__builtin_ctzll(pmask) : L;
      const int nk = L - s0;
      const int LTc = (nk + 15) >> 4;
      if (st == XS_NBW - 1 && lane == 0) Hdr[buf * 4] = LTc;  // (the wave that most often has no tile of its own)
      const bool third = shares_tile0(pmask);  // this wave projects a third of tile 0's K features
      const int st_own = st;
      const int st = third ? 0 : st_own;       // (shadows the wave's own tile inside the build)
      if (st < LTc) {  // (uniform) tiles beyond the profile are never read
        const int t = 16 * st + ln, r = s0 + t;
        const bool valid = r < L;
        // Masks cost a select per element and this wave's arithmetic is not hidden by anyone's MFMAs (fp32 MFMA and fp32
        // VALU work of a SIMD do not overlap: tools/mfma_valu_coexec.hip), so they are applied only where they can bite:
        // column groups that reach beyond d, tiles that reach beyond the profile (both wave-uniform conditions).
        const bool tile_full = s0 + 16 * st + 15 < L;
        f32x4 x[G::NKG];
        float s = 0.f;
#pragma unroll
        for (int kg = 0; kg < G::NKG; ++kg) {
          x[kg] = xr[kg];
          if (XS_RARE(16 * kg + 16 > d)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[kg][e] = (16 * kg + 4 * mq + e < d) ? x[kg][e] : 0.f;
          }
          if (XS_RARE(!tile_full)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[kg][e] = valid ? x[kg][e] : 0.f;
          }
          s += (x[kg][0] + x[kg][1]) + (x[kg][2] + x[kg][3]);
        }
        if (ln_on) {  // final LayerNorm (carca.py:421): biased variance, eps inside the root
          const float inv_d = 1.0f / (float)d;
          const float mean = quad4_sum(s) * inv_d;
          float q = 0.f;
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[kg][e] -= mean;
            if (XS_RARE(16 * kg + 16 > d)) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[kg][e] = (16 * kg + 4 * mq + e < d) ? x[kg][e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) q += x[kg][e] * x[kg][e];
          }
          const float rstd = 1.0f / sqrtf(quad4_sum(q) * inv_d + 1e-5f);
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            const f32x4 w = lds4(Lnw + 16 * kg + 4 * mq), b = lds4(Lnb + 16 * kg + 4 * mq);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[kg][e] = x[kg][e] * (rstd * w[e]) + b[e];
            if (XS_RARE(16 * kg + 16 > d)) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[kg][e] = (16 * kg + 4 * mq + e < d) ? x[kg][e] : 0.f;
            }
            if (XS_RARE(!tile_full)) {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[kg][e] = valid ? x[kg][e] : 0.f;
            }
          }
        }
        float* Ks = Ks2 + buf * ATT_LMAX * G::SO;
        const int nvalid = nk - 16 * st;  // slots of this tile that the profile holds (>= 1: st < LTc)
        if (nvalid <= 4 && !third && !(a.opt & 2)) {
          // A NEARLY EMPTY last slot tile (a full profile of L = 50 re-based: slots 48, 49): its K rows on the VALU.  The
          // MFMA tile below costs 144 MFMAs whatever it holds, and at full profiles it sits on the SIMD that also runs six
          // of the user's 21 jobs where the others run five: 49 instead of 48 slots cost 25 % of the kernel
          // (tools/k4_length_probe.py: 224 against 169 us at B = 4096).  Here: the <= 4 normalised rows through LDS, every
          // lane the dot products of features lane and lane + 64 with them (W_K fragments and rows as 16-byte reads; the
          // rows' reads are broadcasts), rows nvalid .. 15 of the image zero (masked keys: anything finite).  a.opt bit 1
          // (tuning key 14) switches back to the MFMA tile.
          float* Xs = lds + M::XS;
          if (ln < 4) {
#pragma unroll
            for (int kg = 0; kg < G::NKG; ++kg) *reinterpret_cast<f32x4*>(Xs + ln * DPI + 16 * kg + 4 * mq) = x[kg];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave writes and reads: its LDS operations are in order)
          // v_mfma_f32_4x4x1 (16 blocks of 4 x 4, one k per instruction): block b = features 4 b .. 4 b + 3 (A: lane l supplies
          // W_K[feature l][k]), columns = the four slots (B: lane l supplies x[slot l & 3][k]), D: lane l holds
          // K[slot l & 3][features 4 (l >> 2) .. + 3] -- 64 features x 4 slots per instruction at 8 cycles, 2 DPO / 64 x DPI of
          // them = 1.5 k cycles where the 16 x 16 tile costs 4.6 k.  (First tried on the VALU, 768 FMAs for four rows: fp32
          // VALU and MFMA share the SIMD's issue, the path cost as much as the tile it replaced -- 215 against 224 us.)
          constexpr int FPL = (G::DPO + 63) / 64;  // halves of 64 features
          const int fb = 4 * (lane >> 2);
          f32x4 dk[FPL];
          int fo[FPL];  // float offset of feature (lane + 64 q)'s fragment row inside a (feature tile, k group) block of W_K
#pragma unroll
          for (int q = 0; q < FPL; ++q) {
            const int f = min(lane + 64 * q, G::DPO - 1);
            fo[q] = ((f >> 4) * G::NKG * 64 + (f & 15)) * 4;
            dk[q] = lds4(Bk + min(64 * q + fb, G::DPO - 4));
          }
          const float* xrow = Xs + (lane & 3) * DPI;
#pragma unroll 2
          for (int kg = 0; kg < G::NKG; ++kg) {
#pragma unroll
            for (int m4 = 0; m4 < 4; ++m4) {
              const f32x4 xv = lds4(xrow + 16 * kg + 4 * m4);
              f32x4 w[FPL];
#pragma unroll
              for (int q = 0; q < FPL; ++q) w[q] = lds4(Wk + fo[q] + (kg * 64 + m4 * 16) * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < FPL; ++q) dk[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[q][e], xv[e], dk[q], 0, 0, 0);
            }
          }
          // rows 0..3 of the tile (rows at or past nvalid: x = 0, i.e. the bias -- masked keys, anything finite), rows 4..15 zeros
#pragma unroll
          for (int q = 0; q < FPL; ++q)
            if (64 * q + fb < G::DPO) *reinterpret_cast<f32x4*>(Ks + (16 * st + (lane & 3)) * G::SO + 64 * q + fb) = dk[q];
          constexpr int QPR = G::DPO / 4;  // 16-byte groups per row
          for (int u = lane; u < 12 * QPR; u += 64) {
            const int r = 4 + u / QPR, c4 = u - (u / QPR) * QPR;
            *reinterpret_cast<f32x4*>(Ks + (16 * st + r) * G::SO + 4 * c4) = zero4();
          }
        } else if (third) {
          constexpr int NT3 = HELP ? G::NF / 3 : 1;
          const int ft0 = st_own * NT3;
          f32x4 acc[NT3];
#pragma unroll
          for (int i = 0; i < NT3; ++i) acc[i] = lds4(Bk + 16 * (ft0 + i) + 4 * mq);
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            f32x4 wf[NT3];
#pragma unroll
            for (int i = 0; i < NT3; ++i) wf[i] = lds4(Wk + (((ft0 + i) * G::NKG + kg) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int i = 0; i < NT3; ++i) acc[i] = mfma16(wf[i][e], x[kg][e], acc[i]);
          }
#pragma unroll
          for (int i = 0; i < NT3; ++i) *reinterpret_cast<f32x4*>(Ks + t * G::SO + 16 * (ft0 + i) + 4 * mq) = acc[i];
        } else {
          // K^T tiles: Ks[16 st + ln][16 ft + 4 mq + r] = sum_k W_K[16 ft + 4 mq + r][k] x[16 st + ln][k] + b_K
          f32x4 acc[G::NF];
#pragma unroll
          for (int ft = 0; ft < G::NF; ++ft) acc[ft] = lds4(Bk + 16 * ft + 4 * mq);
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            f32x4 wf[G::NF];
#pragma unroll
            for (int ft = 0; ft < G::NF; ++ft) wf[ft] = lds4(Wk + ((ft * G::NKG + kg) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int ft = 0; ft < G::NF; ++ft) acc[ft] = mfma16(wf[ft][e], x[kg][e], acc[ft]);
          }
#pragma unroll
          for (int ft = 0; ft < G::NF; ++ft) *reinterpret_cast<f32x4*>(Ks + t * G::SO + 16 * ft + 4 * mq) = acc[ft];
        }
        // folded value u[h][slot] = x[slot] . wu[h] + cu[h]  (of a shared tile 0: by its own wave only)
        if (!third || st_own == 0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          float p = 0.f;
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            const f32x4 w = lds4(Wu + (kg * 64 + 16 * mq + h) * 4);
            p += (x[kg][0] * w[0] + x[kg][1] * w[1]) + (x[kg][2] * w[2] + x[kg][3] * w[3]);
          }
          p = quad4_sum(p) + Cu[h];
          if (mq == 0) Ut2[(buf * NH + h) * ATT_SK + t] = p;
        }
        if (mq == 0) Km2[buf * ATT_LMAX + t] = (t < nk && ((pmask >> (t + s0)) & 1ull)) ? 0.f : FOLD_NEG;
        }
      }
      pmask = __ballot(lane < L && id_next != 0);
      request_rows(k + 1, pmask);
    };
    // previous round's partial logits -> outputs (waves st = 2, 3: four tiles each)
    auto finish = [&](int k, int rd, int sp) {
      if (wave < XS_NW - 2) return;
      const int tl = 4 * (wave - (XS_NW - 2)) + mq;
      const int nt = njobs_of(k, rd) / NH;
      if (tl < nt) {
        const int tile = tlo_of(k) + rd * XS_TPR + tl;
        int gi = 0;
#pragma unroll
        for (int i = 1; i < CARCA_MAX_GROUPS; ++i)
          if (i < a.ngroups && tile >= a.tile_start[i]) gi = i;
        const int n = 16 * (tile - a.tile_start[gi]) + ln;
        const int N = a.g[gi].N, ldy = a.g[gi].ldy ? a.g[gi].ldy : N;
        if (n < N) {
          float logit = ffn_b;
#pragma unroll
          for (int h = 0; h < NH; ++h) logit += Yp2[((sp * XS_TPR + tl) * NH + h) * 16 + ln];
          const int v = user_of(k);
          a.g[gi].y[(size_t)v * ldy + n] = 1.0f / (1.0f + expf(-logit));
        }
      }
    };
    {
      const int id0 = request_ids(0);
      pmask = __ballot(lane < L && id0 != 0);
      request_rows(0, pmask);
    }
    if (wave == XS_NW - 1 && lane < 2) Cnt[lane] = XS_NCW;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // weights landed
    build(0);
    XS_BARRIER();
    int k = 0, rd = 0, sp = 0, pk = 0, prd = 0;
    for (int s = 0; s < nu * R; ++s) {
      XS_WSTAMP(s, 0);
      if (wave == XS_NW - 1 && lane == 0) Cnt[sp ^ 1] = XS_NCW;  // (last drawn from in step s - 1)
      if (s > 0) finish(pk, prd, sp ^ 1);
      XS_WSTAMP(s, 32);
      if (rd == 0 && k + 1 < nu) build(k + 1);
      XS_WSTAMP(s, 16);
      XS_BARRIER();
      pk = k;
      prd = rd;
      sp ^= 1;
      if (++rd == R) {
        rd = 0;
        ++k;
      }
    }
    finish(pk, prd, sp ^ 1);
  } else {
    // =================================================== C waves ==========================================================
    const int cw = wave;
    struct Ops {
      f32x4 q[G::NKG];
      int id;
    } cur;
    auto load_ops = [&](int k, int rd, int job) {
      Job c;
      decode_tile(tlo_of(k) + rd * XS_TPR + job / NH, user_of(k), c);
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) cur.q[kg] = gload4s(c.o, c.lrow * a.ldo + 4 * mq, 16 * kg);
      cur.id = gload1i(c.ids, c.lrow);
    };
    // the job this wave runs after (k, rd, job); false when there is none
    auto find_next = [&](int k, int rd, int cand, int& k2, int& rd2, int& job2) {
      if (cand < njobs_of(k, rd)) {
        k2 = k;
        rd2 = rd;
        job2 = cand;
        return true;
      }
      for (int i = 0; i < R; ++i) {
        if (++rd == R) {
          rd = 0;
          ++k;
        }
        if (k >= nu) return false;
        if (cw < njobs_of(k, rd)) {
          k2 = k;
          rd2 = rd;
          job2 = cw;
          return true;
        }
      }
      return false;
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the B side's weights landed
    for (int c = wave; c < NCH; c += XS_NCW) dma16(a.wq, 4 * lane, 256 * c, Wq + 256 * c);
    if (wave == 8 && lane < G::DPO / 4) dma16(a.bq, 4 * lane, 0, Bq);
    if (wave == 10 && lane < DPI / 4) dma16(a.ffn_w, 4 * lane, 0, Fw);
    {
      int k2 = 0, rd2 = 0, job2 = 0;
      if (!find_next(0, 0, cw, k2, rd2, job2)) {  // (a wave without any job still requests something: no branch)
        k2 = 0;
        rd2 = 0;
        job2 = 0;
      }
      load_ops(k2, rd2, job2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // W_Q landed (and the first job's operands)
    const float qs = a.qscale;
    int k = 0, rd = 0, sp = 0;
    for (int s = 0; s < nu * R; ++s) {
      const int buf = k & 1;
      const int LTc = __builtin_amdgcn_readfirstlane(Hdr[buf * 4]);
      const float* Ks = Ks2 + buf * ATT_LMAX * G::SO;
      const float* Km = Km2 + buf * ATT_LMAX;
      const float* Ut = Ut2 + buf * NH * ATT_SK;
      float* Yp = Yp2 + sp * XS_TPR * NH * 16;
      const int nj = njobs_of(k, rd);
      XS_WSTAMP(s, 0);
      for (int job = cw; job < nj;) {
        // Which job comes next: job + 12 (static dealing).  a.opt bit 0 (tuning key 3, A/B): a wave's first job of a step
        // stays fixed (its operands are requested across the step's barrier) and the others are drawn from a ticket
        // counter.  Measured SLOWER (B = 4096: 204 against 198 us): the ticket has to be drawn at the start of the previous
        // job for the prefetch, and the SIMD's oldest-first issue order already lets its early waves run ahead.
        int ticket = job + XS_NCW;
        if (a.opt & 1) {
          if (lane == 0) ticket = __hip_atomic_fetch_add(&Cnt[sp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const int tl = job / NH, h = job - tl * NH;
        const float* const w0 = Wq + (h * G::NFH * G::NKG) * 256 + 4 * lane;
        const float* const bq0 = Bq + h * DHP + 4 * mq;
        Job c;
        decode_tile(tlo_of(k) + rd * XS_TPR + tl, user_of(k), c);
        const bool q_ok = c.in_range && cur.id != 0;
        // Q^T tiles of the head: NFH interleaved accumulator chains that start from the bias, W_Q fragments from LDS.  The
        // softmax scale is applied inside the exponent below (one fma per score either way), not to the tile.
        f32x4 qt[G::NFH];
        {
#pragma unroll
          for (int ft = 0; ft < G::NFH; ++ft) qt[ft] = lds4(bq0 + 16 * ft);
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            f32x4 af[G::NFH];
#pragma unroll
            for (int ft = 0; ft < G::NFH; ++ft) af[ft] = lds4(w0 + (ft * G::NKG + kg) * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int ft = 0; ft < G::NFH; ++ft) qt[ft] = mfma16(af[ft][e], cur.q[kg][e], qt[ft]);
          }
        }
        // residual part of the logit (w . o, once per target)
        float ypart = 0.f;
        if (a.residual && h == 0) {
#pragma unroll
          for (int kg = 0; kg < G::NKG; ++kg) {
            const f32x4 wv = lds4(Fw + 16 * kg + 4 * mq);
#pragma unroll
            for (int e = 0; e < 4; ++e) ypart += wv[e] * cur.q[kg][e];
          }
          ypart = quad4_sum(ypart);
        }
        // the next job's operands into the same registers: they land under this job's scores and softmax
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        {
          int k2 = k, rd2 = rd, job2 = job;
          if (!find_next(k, rd, ticket, k2, rd2, job2)) {
            k2 = k;
            rd2 = rd;
            job2 = job;
          }
          load_ops(k2, rd2, job2);
        }
        // scores^T tiles (rows = keys, cols = targets) on top of the additive mask.  Exactly the LTc key tiles that hold a
        // slot of the re-based profile: whole pairs as two accumulator chains (one per tile), an odd last tile as two
        // chains over alternate feature tiles that are added afterwards (BASELINE's profile lengths U{3..50}: 2.1 key tiles
        // on average where whole pairs scored 2.75).
        f32x4 sc[ATT_LT];
        float mx = FOLD_NEG;
#pragma unroll
        for (int kp = 0; kp < ATT_LT / 2; ++kp) {
          const float* krow = Ks + (32 * kp + ln) * G::SO + h * DHP + 4 * mq;
          if (2 * kp + 1 < LTc) {
            f32x4 acc0 = lds4(Km + 32 * kp + 4 * mq), acc1 = lds4(Km + 32 * kp + 16 + 4 * mq);
#pragma unroll
            for (int ft = 0; ft < G::NFH; ++ft) {
              const f32x4 k0 = lds4(krow + 16 * ft), k1 = lds4(krow + 16 * G::SO + 16 * ft);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc0 = mfma16(k0[e], qt[ft][e], acc0);
                acc1 = mfma16(k1[e], qt[ft][e], acc1);
              }
            }
            sc[2 * kp] = acc0;
            sc[2 * kp + 1] = acc1;
            mx = fmaxf(fmaxf(mx, fmaxf(acc0[0], acc0[1])), fmaxf(acc0[2], acc0[3]));
            mx = fmaxf(fmaxf(mx, fmaxf(acc1[0], acc1[1])), fmaxf(acc1[2], acc1[3]));
          } else if (2 * kp < LTc) {
            f32x4 acc0 = lds4(Km + 32 * kp + 4 * mq), acc1 = zero4();
#pragma unroll
            for (int ft = 0; ft < G::NFH; ++ft) {
              const f32x4 k0 = lds4(krow + 16 * ft);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (ft & 1) acc1 = mfma16(k0[e], qt[ft][e], acc1);
                else acc0 = mfma16(k0[e], qt[ft][e], acc0);
              }
            }
            if constexpr (G::NFH > 1) acc0 = acc0 + acc1;
            sc[2 * kp] = acc0;
            mx = fmaxf(fmaxf(mx, fmaxf(acc0[0], acc0[1])), fmaxf(acc0[2], acc0[3]));
          }
        }
        mx = quad4_max(mx);
        const float mxs = mx * qs;  // exp2((score - max) * log2(e) / sqrt(dh)), the scale being positive
        float sum = 0.f, dot = 0.f;
#pragma unroll
        for (int kt = 0; kt < ATT_LT; ++kt) {
          if (kt < LTc) {
            const f32x4 uv = lds4(Ut + h * ATT_SK + 16 * kt + 4 * mq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float ex = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][e], qs, -mxs));
              sum += ex;
              dot += ex * uv[e];
            }
          }
        }
        sum = quad4_sum(sum);
        dot = quad4_sum(dot);
        // a target with no allowed key (pad target, all-pad profile) attends nothing: exact 0
        const float attn = (q_ok && mx > 0.5f * FOLD_NEG) ? dot * __builtin_amdgcn_rcpf(sum) : 0.f;  // (1 ulp; a division is ten instructions)
        if (mq == 0) Yp[(tl * NH + h) * 16 + ln] = attn + ypart;
        if (job == cw) XS_WSTAMP(s, 32);
        job = ticket;
      }
      XS_WSTAMP(s, 16);
      XS_BARRIER();
      sp ^= 1;
      if (++rd == R) {
        rd = 0;
        ++k;
      }
    }
  }
  XS_STAMP(4);
#undef XS_STAMP
#undef XS_BARRIER
#undef XS_WSTAMP
}

template <int DPI, int DHP, int NH>
int launch_stream(const FoldArgs& fa, int B, hipStream_t stream) {
  using M = XsLds<DPI, DHP, NH>;
  constexpr size_t lds_bytes = sizeof(float) * M::TOTAL;
  if constexpr (lds_bytes > 160 * 1024) {
    return CARCA_ERR_UNSUPPORTED;  // (d > 96: two weight matrices + two K images do not fit a CU's LDS)
  } else {
    auto kern = cross_stream_kernel<DPI, DHP, NH>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) {
        carca_set_error("cross_score_fwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
        return (int)e;
      }
      attr_set = true;
    }
    const int grid = min(B * fa.nparts, carca_num_cus());
    hipEvent_t e0, e1;
    if (carca_take_launch_events(&e0, &e1))
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(XS_NW * 64), lds_bytes, stream, e0, e1, 0, fa);
    else
      hipLaunchKernelGGL(kern, dim3(grid), dim3(XS_NW * 64), lds_bytes, stream, fa);
    CARCA_LAUNCH_CHECK();
    return CARCA_OK;
  }
}

}  // namespace

int carca_cross_stream_launch(const FoldArgs& fa, int dpi, int dhp, int H, int B, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_ATT_DISPATCH(launch_stream, fa, B, stream);
  return CARCA_ERR_UNSUPPORTED;
}
