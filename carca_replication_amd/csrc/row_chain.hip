// Row chains of the SelfAttentionBlock backward (autograd of carca.py:297-318) that used to be three launches each:
//   FFN side   dh1pre = (dy W_2) * LeakyReLU'(h1)          ->  ds = dh1pre W_1 (+ dy)  ->  dr = LayerNorm2'(ds; r)
//   input side dqn = dQ W_Q (+ dr),  dx|kv = dK W_K + dV W_V                            ->  dx = LayerNorm1'(dqn; x) + dx|kv
// Every step is local to a row, the weights are d x d: ONE workgroup takes 64 rows through the whole chain -- operand
// tiles in LDS, 16x16x4 fp32 MFMAs (wave (ct, rh): column tile ct of row tiles 2 rh, 2 rh + 1: two accumulator chains),
// weight fragments straight from L2, the LayerNorm backward on the accumulators (row sums over the column-tile waves
// meet in LDS: three rounds -- mean, variance, the two gradient means -- as in layernorm_bwd_kernel), gamma / beta
// gradients as one atomic per column and wave.  Outputs the weight-gradient products need (dh1pre) are written on the way.
// No dropout (p > 0 keeps the launch-per-step path of block_bwd.hip).
#include <hip/hip_ext.h>
#include "attn_common.h"
#include "../../include/carca_hip.h"

namespace {

struct RowChainArgs {
  int rows, d, ld;        // every [rows, ld] activation buffer has row stride ld (= DPI); weights: Bt[n][k], row stride ld
  int mode;               // 0 = FFN side, 1 = input side
  int residual;
  int diag;               // timing experiments of a stamped run (tuning key 9 = mode + 16 * diag): 1 drain vmcnt before the result
                          // goes to LDS, 2 touch the later weight matrices at the kernel's start, 8 slot 7 = LDS writes retired (before the barrier)
  // FFN side: a0 = dy, w0 = w2_t, gate = h1, out1 = dh1pre, w1 = w1_t;  x = r;  out = dr
  // input side: a0 = dQ, w0 = wq_t (T2 = a0 w0 (+ add)), a1 = dK, w1 = wk_t, a2 = dV, w2 = wv_t (U = a1 w1 + a2 w2); x = x_in; out = dx
  const float *a0, *a1, *a2, *w0, *w1, *w2, *gate, *add, *x, *ln_w;
  float *out1, *out, *g_ln_w, *g_ln_b;
  unsigned long long* stamps;  // diagnostic runs (carca_set_debug_buffer): 8 clocks per wave, 16 wave slots per workgroup
};

template <int DPI, int MODE>
__global__ __launch_bounds__(DPI / 16 * 2 * 64) void row_chain_bwd_kernel(const RowChainArgs a) {
  constexpr int NCT = DPI / 16, NW = 2 * NCT, NT = NW * 64, BM = 64, LS = DPI + 4, NKG = DPI / 16;
  constexpr int C4 = DPI / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float (*As)[BM * LS] = reinterpret_cast<float (*)[BM * LS]>(lds);                       // [3] operand tiles (FFN side: dy, then dh1pre)
  float (*Ex)[BM][8] = reinterpret_cast<float (*)[BM][8]>(lds + 3 * BM * LS);             // [2] per (row, column tile) partial sums of a round
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (diagnostic runs: 8 clocks per WAVE -- [workgroup][16 waves][8])
#define RC_STAMP(i)                                                                                                  \
  do {                                                                                                               \
    if (a.stamps && lane == 0) a.stamps[(blockIdx.x * 16 + wave) * 8 + (i)] = __builtin_readcyclecounter();          \
  } while (0)
  RC_STAMP(0);
  const int ln = lane & 15, mq = lane >> 4;
  const int ct = wave % NCT, rh = wave / NCT;
  const int row0 = blockIdx.x * BM;
  const int d = a.d, ld = a.ld;
  const int n = 16 * ct + ln;  // this lane's output column
  const bool n_ok = n < d;
  const int nn = n_ok ? n : d - 1;

  // ---- operand tiles into LDS (rows beyond the end: zeros) -------------------------------------------------------
  // (every request of the kernel goes out before anything waits: unconditional buffer loads with clamped rows -- a load
  // under a branch is waited for right there -- and the weight fragments of all products with them)
  constexpr int SPT = BM * C4 / NT;  // 16-byte slots per thread and tile: 2 in every geometry
  static_assert(SPT * NT == BM * C4, "tile slots divide evenly over the threads");
  constexpr int NIN = MODE == 1 ? 3 : 1;  // operand tiles staged
  f32x4 st[NIN][SPT];
#pragma unroll
  for (int t = 0; t < NIN; ++t) {
    const float* src = t == 0 ? a.a0 : (t == 1 ? a.a1 : a.a2);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int s = tid + j * NT, r = s / C4, c4 = s - r * C4;
      st[t][j] = gload4(src, min(row0 + r, a.rows - 1) * ld + 4 * c4);
    }
  }
  // what the epilogues read per element, requested now: acc[i][r] = C[row0 + 16 (2 rh + i) + 4 mq + r][n]
  float xv[2][4], gv[2][4], av[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = min(row0 + 16 * (2 * rh + i) + 4 * mq + r, a.rows - 1);
      xv[i][r] = gload1(a.x, row * ld + nn);
      gv[i][r] = gload1(a.gate ? a.gate : a.x, row * ld + nn);
      av[i][r] = gload1(a.add ? a.add : a.x, row * ld + nn);
    }
  const float gamma = gload1(a.ln_w, nn);
#pragma unroll
  for (int t = 0; t < NIN; ++t)
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int s = tid + j * NT, r = s / C4, c4 = s - r * C4;
      *reinterpret_cast<f32x4*>(&As[t][r * LS + 4 * c4]) = row0 + r < a.rows ? st[t][j] : zero4();
    }
  // C (2 row tiles x this wave's column tile) = A[slot] Bt^T
  // (the weight fragments -- Bt row = this lane's output column -- are requested ONE product ahead: all of them at the
  // kernel's start, beside the tiles and the per-element operands, measured 10 % slower; 32-row workgroups, i.e. twice
  // the weight traffic, 20 % slower: every wave re-reads its column tile's 6 KB per matrix out of the L2 -- but 32-row
  // workgroups with the matrices staged ONCE per workgroup in LDS, i.e. the same traffic on twice the CUs, were slower too:
  // 28.5 against 24.5 us, so it is not the weight traffic)
  auto load_w = [&](const float* bt, f32x4 (&wf)[NKG]) {
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) wf[kg] = gload4(bt, nn * ld + 16 * kg + 4 * mq);
  };
  f32x4 wfA[NKG], wfB[NKG];
  if (a.diag & 2) {  // one dword of each later matrix per wave: translation + L2 line warm before they are needed
    const float t1 = gload1(a.w1, nn * ld), t2 = gload1(a.w2 ? a.w2 : a.w1, nn * ld);
    asm volatile("" ::"v"(t1), "v"(t2));
  }
  load_w(a.w0, wfA);
  auto gemm = [&](int slot, const f32x4 (&wf)[NKG], f32x4 (&acc)[2]) {
    const float* a0p = &As[slot][(16 * (2 * rh) + ln) * LS + 4 * mq];
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
      const f32x4 x0 = lds4(a0p + 16 * kg), x1 = lds4(a0p + 16 * LS + 16 * kg);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = mfma16(x0[e], wf[kg][e], acc[0]);
        acc[1] = mfma16(x1[e], wf[kg][e], acc[1]);
      }
    }
  };
  __syncthreads();
  RC_STAMP(1);
  f32x4 t2[2] = {zero4(), zero4()}, u[2] = {zero4(), zero4()};
  if constexpr (MODE == 0) {
    f32x4 t1[2] = {zero4(), zero4()};
    gemm(0, wfA, t1);
    load_w(a.w1, wfB);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * (2 * rh + i) + 4 * mq + r;
        // LeakyReLU'(h1): slope 0.01 at and below 0 (no dropout on this path)
        const float v = n_ok ? t1[i][r] * (gv[i][r] > 0.f ? 1.0f : 0.01f) : 0.f;
        As[1][lrow * LS + n] = v;
        t1[i][r] = v;
      }
    if (a.diag & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RC_STAMP(2);
    if (a.diag & 8) {  // (slot 7 = LDS writes retired, before the barrier)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RC_STAMP(7);
    }
    __syncthreads();
    RC_STAMP(3);
    gemm(1, wfB, t2);
    // dh1pre goes to memory only now: vmcnt retires in order, so a store issued before the product's last wait on its
    // weight fragments made that wait a wait for the stores (measured: the product took 14 k cycles instead of 6 k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * (2 * rh + i) + 4 * mq + r;
        if (row0 + lrow < a.rows) a.out1[(size_t)(row0 + lrow) * ld + n] = t1[i][r];
      }
  } else {
    load_w(a.w1, wfB);
    gemm(0, wfA, t2);
    load_w(a.w2, wfA);
    gemm(1, wfB, u);
    gemm(2, wfA, u);
  }
  RC_STAMP(4);
  // ---- LayerNorm backward of t2 (+ add) against x, plus u -------------------------------------------------------------
  const float inv_d = 1.0f / (float)d;
  float dyv[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dyv[i][r] = n_ok ? t2[i][r] + ((a.add && a.residual) ? av[i][r] : 0.f) : 0.f;
      xv[i][r] = n_ok ? xv[i][r] : 0.f;
    }
  // one exchange round: every row's sum of `v` (and `w`) over all columns
  auto row_sums = [&](float (&v)[2][4], float (&w)[2][4], bool two) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * (2 * rh + i) + 4 * mq + r;
        const float s0 = row16_sum(v[i][r]);
        const float s1 = two ? row16_sum(w[i][r]) : 0.f;
        if (ln == 0) {
          Ex[0][lrow][ct] = s0;
          if (two) Ex[1][lrow][ct] = s1;
        }
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * (2 * rh + i) + 4 * mq + r;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          s0 += Ex[0][lrow][c];
          if (two) s1 += Ex[1][lrow][c];
        }
        v[i][r] = s0;
        w[i][r] = s1;
      }
    __syncthreads();
  };
  float s0[2][4], s1[2][4], hv[2][4], rstd[2][4], aa[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) s0[i][r] = xv[i][r];
  row_sums(s0, s1, false);  // mean
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hv[i][r] = n_ok ? xv[i][r] - s0[i][r] * inv_d : 0.f;
      s0[i][r] = hv[i][r] * hv[i][r];
    }
  row_sums(s0, s1, false);  // variance of the centred row
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rstd[i][r] = 1.0f / sqrtf(s0[i][r] * inv_d + 1e-5f);
      hv[i][r] *= rstd[i][r];
      aa[i][r] = dyv[i][r] * gamma;
      s0[i][r] = aa[i][r];
      s1[i][r] = aa[i][r] * hv[i][r];
    }
  row_sums(s0, s1, true);  // the two means of the gradient
  RC_STAMP(5);
  float dg = 0.f, db = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + 16 * (2 * rh + i) + 4 * mq + r;
      if (row >= a.rows) continue;
      float o = 0.f;
      if (n_ok) {
        o = rstd[i][r] * (aa[i][r] - s0[i][r] * inv_d - hv[i][r] * s1[i][r] * inv_d) + u[i][r];
        dg += dyv[i][r] * hv[i][r];
        db += dyv[i][r];
      }
      a.out[(size_t)row * ld + n] = o;  // (pad columns: zeros)
    }
  RC_STAMP(6);
  dg = quad4_sum(dg);
  db = quad4_sum(db);
  if (mq == 0 && n_ok) {
    grad_add(&a.g_ln_w[n], dg);
    grad_add(&a.g_ln_b[n], db);
  }
  if (!(a.diag & 8)) RC_STAMP(7);
#undef RC_STAMP
}

template <int DPI, int MODE>
int launch_row_chain(const RowChainArgs& a, hipStream_t stream) {
  const int blocks = (a.rows + 63) / 64;
  constexpr size_t lds_bytes = sizeof(float) * (3 * 64 * (DPI + 4) + 2 * 64 * 8);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)(row_chain_bwd_kernel<DPI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("sa_block_bwd: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((row_chain_bwd_kernel<DPI, MODE>), dim3(blocks), dim3(DPI / 16 * 2 * 64), lds_bytes, stream, a);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

// the two fused chains of carca_sa_block_bwd (block_bwd.hip); dpi = dpo in every supported geometry
int carca_sa_ffn_chain_bwd(const float* dy, const float* h1, const float* r, const float* w2_t, const float* w1_t,
                           const float* ln2_w, int rows, int d, int dpi, int residual, float* dh1pre, float* dr,
                           float* g_ln2_w, float* g_ln2_b, void* stream) {
  RowChainArgs a{};
  a.rows = rows; a.d = d; a.ld = dpi; a.mode = 0; a.residual = residual;
  a.a0 = dy; a.w0 = w2_t; a.w1 = w1_t; a.gate = h1; a.add = dy; a.x = r; a.ln_w = ln2_w;
  a.out1 = dh1pre; a.out = dr; a.g_ln_w = g_ln2_w; a.g_ln_b = g_ln2_b;
  a.stamps = (carca_tuning(CARCA_TUNE_STAMPS) & 3) == 1 ? carca_debug_buffer() : nullptr;
  a.diag = carca_tuning(CARCA_TUNE_STAMPS) >> 4;
  switch (dpi) {
    case 64: return launch_row_chain<64, 0>(a, (hipStream_t)stream);
    case 96: return launch_row_chain<96, 0>(a, (hipStream_t)stream);
    case 128: return launch_row_chain<128, 0>(a, (hipStream_t)stream);
  }
  carca_set_error("sa_block_bwd: unsupported padded width %d", dpi);
  return CARCA_ERR_UNSUPPORTED;
}

int carca_sa_input_chain_bwd(const float* dqh, const float* dkh, const float* dvh, const float* dr, const float* x_in,
                             const float* wq_t, const float* wk_t, const float* wv_t, const float* ln1_w, int rows, int d,
                             int dpi, int residual, float* dx, float* g_ln1_w, float* g_ln1_b, void* stream) {
  RowChainArgs a{};
  a.rows = rows; a.d = d; a.ld = dpi; a.mode = 1; a.residual = residual;
  a.a0 = dqh; a.a1 = dkh; a.a2 = dvh; a.w0 = wq_t; a.w1 = wk_t; a.w2 = wv_t; a.add = dr; a.x = x_in; a.ln_w = ln1_w;
  a.out = dx; a.g_ln_w = g_ln1_w; a.g_ln_b = g_ln1_b;
  a.stamps = (carca_tuning(CARCA_TUNE_STAMPS) & 3) == 2 ? carca_debug_buffer() : nullptr;
  a.diag = carca_tuning(CARCA_TUNE_STAMPS) >> 4;
  switch (dpi) {
    case 64: return launch_row_chain<64, 1>(a, (hipStream_t)stream);
    case 96: return launch_row_chain<96, 1>(a, (hipStream_t)stream);
    case 128: return launch_row_chain<128, 1>(a, (hipStream_t)stream);
  }
  carca_set_error("sa_block_bwd: unsupported padded width %d", dpi);
  return CARCA_ERR_UNSUPPORTED;
}
