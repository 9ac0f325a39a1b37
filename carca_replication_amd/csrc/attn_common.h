// Building blocks shared by the self-attention block (K2) and cross-attention scoring (K4) kernels.
//
// One workgroup owns one user; the user's profile (L <= 64 slots, padded to 16-slot tiles) lives
// in LDS.  All products are 16x16x4 fp32 MFMA tiles D[m][n] = sum_k A[m][k] Bt[n][k] (carca_common.h).
// A tile's result sits in the D layout: lane (n = l&15, mq = l>>4) holds rows m = 4*mq + r, r = 0..3.
// A product that contracts over the PREVIOUS product's m index takes that result straight from
// registers as its Bt operand (step r of group g contracts index 16g + 4*mq + r on both sides), so
// Q^T -> scores^T -> P^T -> O^T -> (FFN1 -> FFN2) chain without touching LDS.  That is why
//   - heads are padded to DHP = round_up(dh, 16) features: head boundaries fall on tile boundaries;
//   - K is kept [key][head-padded feature], V transposed [head-padded feature][key].
#pragma once
#include "carca_common.h"
#include "../../include/carca_hip.h"

#define ATT_LMAX 64
#define ATT_LT 4      // 16-slot tiles in ATT_LMAX
#define ATT_SK 72     // key stride of Vt rows: 64 + 8 (conflict-free ds_read_b128, brute-forced)

template <int DPI, int DHP, int NH>
struct AttGeom {
  static constexpr int DPO = DHP * NH;
  static constexpr int SI = DPI + 8;  // row stride of [slot][input feature] images
  static constexpr int SO = DPO + 8;  // row stride of [slot][head-padded feature] images
  static constexpr int NKG = DPI / 16;
  static constexpr int NFH = DHP / 16;  // feature tiles per head
  static constexpr int NF = DPO / 16;
};

// counter-based dropout: keep element `idx` of `site` iff a 24-bit hash of (seed, site, idx) >= p * 2^24
struct DropCfg {
  unsigned thresh;  // 0 = dropout off
  unsigned s0, s1;  // seed halves
  float scale;      // 1 / (1 - p)
  const unsigned long long* offset;  // device, or null: added to the seed when the kernel starts (hipGraph replays)
};
__host__ inline DropCfg make_drop(const CarcaDropout* d) {
  DropCfg c{0u, 0u, 0u, 1.0f, nullptr};
  if (d && d->p > 0.f) {
    double t = (double)d->p * 16777216.0;
    c.thresh = t >= 16777215.0 ? 16777215u : (unsigned)t;
    if (c.thresh == 0) c.thresh = 1;
    c.s0 = (unsigned)d->seed;
    c.s1 = (unsigned)(d->seed >> 32);
    c.scale = (float)(1.0 / (1.0 - (double)d->p));
    c.offset = (const unsigned long long*)d->seed_offset;
  }
  return c;
}
// the launch's DropCfg with the device-side seed offset applied (a kernel's first statement about dropout)
__device__ __forceinline__ DropCfg drop_resolve(DropCfg c) {
  if (c.thresh && c.offset) {
    const unsigned long long sd = (((unsigned long long)c.s1 << 32) | c.s0) + *c.offset;
    c.s0 = (unsigned)sd;
    c.s1 = (unsigned)(sd >> 32);
  }
  return c;
}
__device__ __forceinline__ bool drop_keep(const DropCfg& c, unsigned site, unsigned idx) {
  unsigned x = idx * 0x9E3779B1u + c.s0;
  x ^= (site + 1u) * 0x85EBCA77u + c.s1;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return (x >> 8) >= c.thresh;
}

// Weight matrices reach these kernels in FRAGMENT ORDER (CarcaPackDesc.frag16): element offset of the four k values
// lane (ln, mq) needs from the 16x16 tile (row tile rt, k group kg) of a matrix with NKG k groups per row.  One wave
// load = 1 KB contiguous (8 full lines); row-major it was 16 half-used lines and the loads cost 15 k of the 21 k
// cycles of the K/V phase (measured with the loads / the MFMAs switched off in turn).
__device__ __forceinline__ int wfrag_off(int rt, int kg, int nkg, int lane) { return ((rt * nkg + kg) * 64 + lane) * 4; }

// Left alone, hipcc (128-VGPR budget of the 1024-thread kernels) sinks most of a tile's weight-fragment loads between
// its MFMA groups, so that their L2 latencies chain (3 exposed latencies per 24-MFMA tile in the ISA).  A scheduling
// barrier right after the loads keeps all of them in flight before the first MFMA waits.
#define CARCA_PIN_LOADS() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 glb4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}

// LayerNorm of one row held two elements per lane (columns lane and lane+64), torch semantics
// (biased variance, eps inside the sqrt; carca.py:279,283,408).  Columns >= d must hold 0 on entry.
__device__ __forceinline__ void row_layernorm(float& v0, float& v1, int lane, int d, const float* __restrict__ w,
                                              const float* __restrict__ b) {
  const float inv_d = 1.0f / (float)d;
  const float mean = wave_sum(v0 + v1) * inv_d;
  const float d0 = lane < d ? v0 - mean : 0.f;
  const float d1 = lane + 64 < d ? v1 - mean : 0.f;
  const float var = wave_sum(d0 * d0 + d1 * d1) * inv_d;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  v0 = lane < d ? d0 * rstd * w[lane] + b[lane] : 0.f;
  v1 = lane + 64 < d ? d1 * rstd * w[lane + 64] + b[lane + 64] : 0.f;
}

// The same LayerNorm with the affine pair already in registers (rows_in_flight below holds several rows per wave).
__device__ __forceinline__ void row_layernorm_regs(float& v0, float& v1, int lane, int d, float w0, float w1, float b0,
                                                   float b1) {
  const float inv_d = 1.0f / (float)d;
  const float mean = wave_sum(v0 + v1) * inv_d;
  const float d0 = lane < d ? v0 - mean : 0.f;
  const float d1 = lane + 64 < d ? v1 - mean : 0.f;
  const float var = wave_sum(d0 * d0 + d1 * d1) * inv_d;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  v0 = lane < d ? d0 * rstd * w0 + b0 : 0.f;
  v1 = lane + 64 < d ? d1 * rstd * w1 + b1 : 0.f;
}

// K^T-style projection tile: out[slot][16ft + 4mq + r] = sum_k W[16ft + 4mq + r][k] X[slot][k] + bias
// (A = packed weight rows from global, Bt = slot rows from LDS); written as one 16-B LDS store per lane.
template <int DPI>
__device__ __forceinline__ void proj_tile_feat_major(const float* __restrict__ Wp, const float* __restrict__ bp,
                                                     const float* xs, int si, float* out, int so, int ft, int st,
                                                     int lane, float* __restrict__ save = nullptr, int dpo = 0,
                                                     int L = 0) {
  const int ln = lane & 15, mq = lane >> 4;
  const float* xrow = xs + (16 * st + ln) * si + 4 * mq;
  f32x4 wf[DPI / 16];  // all weight fragments first: their latencies overlap instead of chaining
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) wf[kg] = gload4(Wp, wfrag_off(ft, kg, DPI / 16, lane));
  const f32x4 bias = gload4(bp, 16 * ft + 4 * mq);
  CARCA_PIN_LOADS();
  f32x4 acc = zero4();
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) acc = mfma16_group(wf[kg], lds4(xrow + 16 * kg), acc);
  acc = acc + bias;
  *reinterpret_cast<f32x4*>(out + (16 * st + ln) * so + 16 * ft + 4 * mq) = acc;
  if (save && 16 * st + ln < L)  // [slot][head-padded feature], saved for backward
    *reinterpret_cast<f32x4*>(save + (size_t)(16 * st + ln) * dpo + 16 * ft + 4 * mq) = acc;
}

// V^T-style projection tile: out[16ft + n][16st + 4mq + r] = sum_k X[16st + 4mq + r][k] W[16ft + n][k] + bias[16ft+n]
// (A = slot rows from LDS, Bt = packed weight rows from global).
template <int DPI>
__device__ __forceinline__ void proj_tile_slot_major(const float* __restrict__ Wp, const float* __restrict__ bp,
                                                     const float* xs, int si, float* out, int sk, int ft, int st,
                                                     int lane, float* __restrict__ save = nullptr, int dpo = 0,
                                                     int L = 0) {
  const int ln = lane & 15, mq = lane >> 4;
  const float* xrow = xs + (16 * st + ln) * si + 4 * mq;
  f32x4 wf[DPI / 16];
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) wf[kg] = gload4(Wp, wfrag_off(ft, kg, DPI / 16, lane));
  const float bias = gload1(bp, 16 * ft + ln);
  CARCA_PIN_LOADS();
  f32x4 acc = zero4();
#pragma unroll
  for (int kg = 0; kg < DPI / 16; ++kg) acc = mfma16_group(lds4(xrow + 16 * kg), wf[kg], acc);
  f32x4 o = {acc[0] + bias, acc[1] + bias, acc[2] + bias, acc[3] + bias};
  *reinterpret_cast<f32x4*>(out + (16 * ft + ln) * sk + 16 * st + 4 * mq) = o;
  if (save) {  // [slot][head-padded feature]
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * st + 4 * mq + r < L) save[(size_t)(16 * st + 4 * mq + r) * dpo + 16 * ft + ln] = o[r];
  }
}

// One head of attention for a 16-query tile, everything in registers.
//   qfrag[kg]   : the queries' input rows as Bt fragments (lane (query, mq) holds k = 16kg + 4mq + 0..3)
//   Ks, Vt      : LDS images of this user's keys / values
//   key_ok(key, r-th) supplied by the caller through `okbits`: bit (4*kt + r) of okbits = this lane's
//                 query may attend key 16kt + 4mq + r
//   returns o[ft] = O^T tile rows (head-padded features 16ft + 4mq + r of head h) for the lane's query,
//   and (optionally) leaves the probabilities in p[kt].
template <int DPI, int DHP, int NH, bool HOIST_WQ = false>
__device__ __forceinline__ void attend_head(const f32x4 (&qfrag)[DPI / 16], const float* __restrict__ wq,
                                            const float* __restrict__ bq, const float* Ks, const float* Vt, int h,
                                            int nkt, unsigned okbits, float sqrt_dh,
                                            f32x4 (&o)[DHP / 16], f32x4 (&p)[ATT_LT], int lane,
                                            float* __restrict__ qh_row = nullptr, const DropCfg* dc = nullptr,
                                            unsigned site = 0, unsigned midx = 0, uint8_t* mrow = nullptr,
                                            int nkeys = 0) {
  using G = AttGeom<DPI, DHP, NH>;
  const int ln = lane & 15, mq = lane >> 4;
  // Q^T tiles of this head.  HOIST_WQ (callers with registers to spare: the self-attention block; the scoring kernel
  // spills with it) and few enough fragments (<= 12 x 16 B per lane): ALL of the head's weight loads are issued before
  // the first chain, so one L2 latency is exposed per head instead of one per feature tile.
  f32x4 qt[G::NFH];
  if constexpr (HOIST_WQ && G::NFH * G::NKG <= 12) {
    f32x4 wf[G::NFH][G::NKG], bias[G::NFH];
#pragma unroll
    for (int ft = 0; ft < G::NFH; ++ft) {
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) wf[ft][kg] = gload4(wq, wfrag_off(h * (DHP / 16) + ft, kg, G::NKG, lane));
      bias[ft] = gload4(bq, h * DHP + 16 * ft + 4 * mq);
    }
    CARCA_PIN_LOADS();
#pragma unroll
    for (int ft = 0; ft < G::NFH; ++ft) {
      f32x4 acc = zero4();
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(wf[ft][kg], qfrag[kg], acc);
      qt[ft] = acc + bias[ft];
      if (qh_row) *reinterpret_cast<f32x4*>(qh_row + h * DHP + 16 * ft + 4 * mq) = qt[ft];  // saved for backward
    }
  } else {
#pragma unroll
    for (int ft = 0; ft < G::NFH; ++ft) {
      f32x4 wf[G::NKG];
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) wf[kg] = gload4(wq, wfrag_off(h * (DHP / 16) + ft, kg, G::NKG, lane));
      const f32x4 bias = gload4(bq, h * DHP + 16 * ft + 4 * mq);
      CARCA_PIN_LOADS();
      f32x4 acc = zero4();
#pragma unroll
      for (int kg = 0; kg < G::NKG; ++kg) acc = mfma16_group(wf[kg], qfrag[kg], acc);
      qt[ft] = acc + bias;
      if (qh_row) *reinterpret_cast<f32x4*>(qh_row + h * DHP + 16 * ft + 4 * mq) = qt[ft];  // saved for backward
    }
  }
  // scores^T tiles: rows = keys, cols = queries
#pragma unroll
  for (int kt = 0; kt < ATT_LT; ++kt) {
    p[kt] = zero4();
    if (kt < nkt) {
      const float* krow = Ks + (16 * kt + ln) * G::SO + h * DHP + 4 * mq;
      f32x4 acc = zero4();
#pragma unroll
      for (int ft = 0; ft < G::NFH; ++ft) acc = mfma16_group(lds4(krow + 16 * ft), qt[ft], acc);
      p[kt] = acc;
    }
  }
  // masked softmax over keys: (mask + QK^T) / sqrt(dh) -> softmax -> * mask   (carca.py:251-256).
  // A masked score is -2^32/sqrt(dh) in the reference and underflows to an exact 0 weight next to
  // any unmasked score; a row with no unmasked key becomes all zeros.  Same thing, said directly:
  // (the scale is applied as a multiplication by 1/sqrt(dh): an fp32 division is ~10 instructions per score, and this
  // wave's instruction stream IS the critical path; 1 ulp on the score, exact for dh = 16 and 64)
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
  // dropout on the attention weights (carca.py:258): element (user, head, query, key) = midx + key
  if (dc && dc->thresh) {
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * mq + r;
        const bool keep = drop_keep(*dc, site, midx + (unsigned)key);
        p[kt][r] = keep ? p[kt][r] * dc->scale : 0.f;
        if (mrow && key < nkeys) mrow[key] = keep ? 1 : 0;
      }
  }
  // O^T tiles: rows = head features, cols = queries, contracting over keys
#pragma unroll
  for (int ft = 0; ft < G::NFH; ++ft) {
    const float* vrow = Vt + (h * DHP + 16 * ft + ln) * ATT_SK + 4 * mq;
    f32x4 acc = zero4();
#pragma unroll
    for (int kt = 0; kt < ATT_LT; ++kt)
      if (kt < nkt) acc = mfma16_group(lds4(vrow + 16 * kt), p[kt], acc);
    o[ft] = acc;
  }
}

// (DPI, DHP, H) combinations with a kernel instantiation; anything else is CARCA_ERR_UNSUPPORTED
#define CARCA_ATT_DISPATCH(FN, ...)                                               \
  do {                                                                            \
    if (dpi == 64 && dhp == 16 && H == 4) return FN<64, 16, 4>(__VA_ARGS__);      \
    if (dpi == 64 && dhp == 32 && H == 2) return FN<64, 32, 2>(__VA_ARGS__);      \
    if (dpi == 64 && dhp == 64 && H == 1) return FN<64, 64, 1>(__VA_ARGS__);      \
    if (dpi == 96 && dhp == 32 && H == 3) return FN<96, 32, 3>(__VA_ARGS__);      \
    if (dpi == 96 && dhp == 48 && H == 2) return FN<96, 48, 2>(__VA_ARGS__);      \
    if (dpi == 96 && dhp == 96 && H == 1) return FN<96, 96, 1>(__VA_ARGS__);      \
    if (dpi == 128 && dhp == 32 && H == 4) return FN<128, 32, 4>(__VA_ARGS__);    \
    if (dpi == 128 && dhp == 64 && H == 2) return FN<128, 64, 2>(__VA_ARGS__);    \
    if (dpi == 128 && dhp == 128 && H == 1) return FN<128, 128, 1>(__VA_ARGS__);  \
  } while (0)
