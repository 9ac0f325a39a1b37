// Split-precision feature GEMM (opt-in, tuning key 16): AllEmbedding.feats_embed (carca.py:86), q = [a ; c] W_f^T + b_f,
// on the 16-bit MFMA pipe -- 16x the rate of the exact-fp32 MFMA the default path is bound by (MI355X_MICROARCH.md,
// Matrix cores) -- with fp32-class accuracy (hard part 1: "fp32 MFMA, or a bf16x3 split-accumulate scheme").
//
// Every fp32 operand x is written as a short sum of 16-bit parts, round-to-nearest each time:
//   mode 1, bf16 x 3:  x = p0 + p1 + p2 (+ <= 2^-24 |x|),  p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1);
//                      a.b ~= a0 b0 + [a0 b1 + a1 b0 + a0 b2 + a1 b1 + a2 b0]   six products (what is dropped is 2^-24 relative),
//                      the leading product in one fp32 accumulator, the five corrections in a second one;
//   mode 2, fp16 x 2:  x = h0 + 2^-11 h1 (+ <= 2^-22 |x|), h0 = fp16(x), h1 = fp16((x - h0) 2^11) -- the residual is
//                      scaled so that it stays a NORMAL fp16 number wherever x is one;
//                      a.b ~= a0 b0 + 2^-11 [a0 b1 + a1 b0]                     three products, two accumulators
//                      (|x| < 65504 required; dropped terms 2^-22 relative: within 4x of fp32's own product rounding).
// A product of two 16-bit parts is exact in fp32, so what remains is the fp32 accumulation the default path has as well.
//
// The kernel has gemm_rows_cu_kernel's grid (gemm.hip): ONE workgroup per CU, tile 384 x 96, K step 32 -- but four waves,
// one per SIMD, each with a 96 x 96 sub-tile (v_mfma_f32_32x32x16_{bf16,f16}: lane (r = l & 31, h = l >> 5) supplies
// A[row r][k = 8 h .. 8 h + 7] and Bt[col r][k = 8 h .. 8 h + 7]; C / D as the fp32 MFMA's).
//   A stays fp32 in HBM (the caller's tensor) and in LDS; a wave splits the eight values of its fragment in registers
//   right behind the fragment read -- once per (32 rows x 16 k), reused over its three column tiles.
//   W_f is split ONCE per weight version into packed planes (carca_split_pack) laid out in the order a workgroup stages
//   them: [column block][K step][part][96 rows][32 k], 6 KB contiguous per (step, part) -- fully coalesced staging loads.
// Data, not MFMA, bounds mode 2: a K step is 60 KB through the CU's load path against 1728 cycles of MFMA.
#include <hip/hip_ext.h>
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
struct Split;
template <>
struct Split<1> {  // bf16 x 3
  static constexpr int NP = 3;
  typedef b16x8 vec;
  static __device__ __forceinline__ void split(const f32x4 x0, const f32x4 x1, vec (&p)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = j < 4 ? x0[j & 3] : x1[j & 3];
      const __bf16 b0 = (__bf16)v;
      const float r1 = v - (float)b0;
      const __bf16 b1 = (__bf16)r1;
      p[0][j] = b0;
      p[1][j] = b1;
      p[2][j] = (__bf16)(r1 - (float)b1);
    }
  }
  static __device__ __forceinline__ f32x16 mfma(const vec a, const vec b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void products(const vec (&a)[3], const vec (&b)[3], f32x16& lead, f32x16& corr) {
    lead = mfma(a[0], b[0], lead);
    corr = mfma(a[0], b[2], corr);  // (smallest first)
    corr = mfma(a[1], b[1], corr);
    corr = mfma(a[2], b[0], corr);
    corr = mfma(a[0], b[1], corr);
    corr = mfma(a[1], b[0], corr);
  }
  static __device__ __forceinline__ float combine(float lead, float corr) { return lead + corr; }
};
template <>
struct Split<2> {  // fp16 x 2, second part scaled by 2^11
  static constexpr int NP = 2;
  typedef h16x8 vec;
  static __device__ __forceinline__ void split(const f32x4 x0, const f32x4 x1, vec (&p)[2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = j < 4 ? x0[j & 3] : x1[j & 3];
      const _Float16 h0 = (_Float16)v;
      p[0][j] = h0;
      p[1][j] = (_Float16)((v - (float)h0) * 2048.f);
    }
  }
  static __device__ __forceinline__ f32x16 mfma(const vec a, const vec b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void products(const vec (&a)[2], const vec (&b)[2], f32x16& lead, f32x16& corr) {
    lead = mfma(a[0], b[0], lead);
    corr = mfma(a[0], b[1], corr);
    corr = mfma(a[1], b[0], corr);
  }
  static __device__ __forceinline__ float combine(float lead, float corr) { return fmaf(corr, 1.0f / 2048.f, lead); }
};

constexpr int SP_BM = 384, SP_BN = 96, SP_BK = 32;
constexpr int SP_LSA = SP_BK + 4;   // floats per A row in LDS: the 16-byte fragment reads of 16 rows cover all 64 banks once
constexpr int SP_LSB = SP_BK + 8;   // 16-bit values per Bt row in LDS (80 bytes): the same for the 16-byte B reads
constexpr int SP_PLANE = SP_BN * SP_BK;  // 16-bit values of one packed (column block, K step, part): 6 KB

constexpr int SPLIT_TM = 1;  // row tiles per wave: 1 = twelve waves (three per SIMD), 3 = four waves (one per SIMD)
struct SplitDev {
  CarcaGemmDesc d;
  int rb_start[CARCA_MAX_SEGS + 1];
  int nrb, ncb;
  const uint16_t* wp;  // packed planes [ncb][ntiles][NP][96][32]
};

// W (fp32, [N, ldb0], the K0 columns of k-source 0) -> packed planes.  One thread = 8 consecutive k of one row.
template <int MODE>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ bt0, int ldb0, int K0, int N, int ncb,
                                                         int ntiles, uint16_t* __restrict__ out) {
  using S = Split<MODE>;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)ncb * ntiles * (SP_BN * 4)) return;
  const int c = (int)(idx & 3), n = (int)((idx >> 2) % SP_BN);
  const long ct = idx / (SP_BN * 4);
  const int t = (int)(ct % ntiles), cb = (int)(ct / ntiles);
  const int gn = cb * SP_BN + n;
  const int k0 = t * SP_BK + c * 8;
  f32x4 x0, x1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (gn < N && k0 + j < K0) ? bt0[(size_t)gn * ldb0 + k0 + j] : 0.f;
    if (j < 4) x0[j & 3] = v; else x1[j & 3] = v;
  }
  typename S::vec p[S::NP];
  S::split(x0, x1, p);
#pragma unroll
  for (int q = 0; q < S::NP; ++q)
    *reinterpret_cast<typename S::vec*>(out + ((size_t)(ct * S::NP + q) * SP_PLANE + (n * 4 + c) * 8)) = p[q];
}

// ONE wave per SIMD (256 threads, up to 512 registers per lane): wave w owns rows 96 w .. 96 w + 95 x all 96 columns, 3 x 3
// tiles of 32 x 32 in two accumulator sets (288 registers).  Every fragment it reads from LDS serves three tiles (0.33 KB of
// LDS reads per MFMA), every A element is split exactly once per workgroup, and nothing but its own instruction stream has
// to hide a wave's VALU work: 24 cycles of issue in the shadow of each 32-cycle MFMA.
// DIAG (tuning key 15, timing experiments with WRONG results; 0 in every shipped launch): 1 no global loads behind the
// prologue, 2 no LDS writes behind the prologue, 4 no MFMAs, 8 no split arithmetic (the fragment's bits reinterpreted),
// 16 no barriers in the K loop, 32 no B fragment reads (one set reused).
template <int MODE, int TM, int DIAG = 0>
__global__ __launch_bounds__(768 / TM) void gemm_rows_split_kernel(const SplitDev args) {
  using S = Split<MODE>;
  constexpr int NP = S::NP, TN = 3, C4 = SP_BK / 4, SP_NT = 768 / TM, RPI = SP_NT / 8;  // (RPI: A rows staged per slot index)
  constexpr int A_PER = SP_BM * C4 / SP_NT;                          // 12 sixteen-byte slots of the A tile per thread
  constexpr int B_SLOTS = NP * SP_BN * 4, B_PER = (B_SLOTS + SP_NT - 1) / SP_NT;  // 768 / 1152 slots: 3 or 5 per thread
  constexpr int A_BUF = SP_BM * SP_LSA, B_BUF = NP * SP_BN * SP_LSB;
  __shared__ __attribute__((aligned(16))) float As[2 * A_BUF];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2 * B_BUF];

  const CarcaGemmDesc& D = args.d;
  const int id = blockIdx.x, total = args.nrb * args.ncb;
  const int xcd = id & 7, q8 = total >> 3, r8 = total & 7;  // (column blocks of a row block on one XCD: gemm.hip)
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int rb = wg / args.ncb, cb = wg - rb * args.ncb;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = (rb - args.rb_start[s]) * SP_BM;
  const int n0 = cb * SP_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nfast = D.K0 / SP_BK;             // whole K steps: the pipelined loop
  const int ntiles = (D.K0 + SP_BK - 1) / SP_BK;  // (+ one ragged step when K0 % 32 != 0; k-source 1 joins in the epilogue)

  // ---- staging slots (loop invariants): slot tid + SP_NT i = row (tid >> 3) + RPI i, 16-byte column tid & 7 -------------
  const int a_r = tid >> 3, a_c4 = tid & 7;
  unsigned a_byte[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int gr = min(row0 + a_r + RPI * i, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t off = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
                       : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                                       : (size_t)gr * D.lda0;
    a_byte[i] = (unsigned)((off + a_c4 * 4) * sizeof(float));
  }
  const int a_lds = a_r * SP_LSA + a_c4 * 4;  // (+ 32 i rows: an immediate)
  int b_lds[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int slot = min(tid + i * SP_NT, B_SLOTS - 1);
    const int part = slot / (SP_BN * 4), rem = slot - part * (SP_BN * 4);
    b_lds[i] = (part * SP_BN + (rem >> 2)) * SP_LSB + (rem & 3) * 8;
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = carca_rsrc(sg.a0);
  // (the packed planes of this column block; the pointer goes through readfirstlane: left to the compiler the resource ended up
  // in vector registers and every B load became a waterfall loop)
  const unsigned long long wp_u = (unsigned long long)(args.wp + (size_t)cb * ntiles * NP * SP_PLANE);
  const unsigned wp_lo = __builtin_amdgcn_readfirstlane((unsigned)wp_u), wp_hi = __builtin_amdgcn_readfirstlane((unsigned)(wp_u >> 32));
  const __amdgpu_buffer_rsrc_t b_rsrc = carca_rsrc((const void*)(((unsigned long long)wp_hi << 32) | wp_lo));
  constexpr int B_STEP_BYTES = NP * SP_PLANE * 2;

  f32x4 ra[A_PER];
  u32x4 rbv[B_PER];
  auto load_a_fast = [&](int t) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_byte[i], t * SP_BK * (int)sizeof(float), 0);
      ra[i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
  };
  auto load_b = [&](int t) {
#pragma unroll
    for (int i = 0; i < B_PER; ++i)  // (a slot past the end repeats the last one: same load, same store)
      rbv[i] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (unsigned)(min(tid + i * SP_NT, B_SLOTS - 1) * 16),
                                                     t * B_STEP_BYTES, 0);
  };
  auto load_a_ragged = [&](int t) {
    // the step that holds the end of K0 (a multiple of 4): a 16-byte group is all inside K0 or all outside; the ones outside
    // re-read the row's last group -- finite data against the zeros the packed W holds there
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int kc = min(t * SP_BK + a_c4 * 4, D.K0 - 4);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_byte[i] + (unsigned)((kc - a_c4 * 4) * (int)sizeof(float)), 0, 0);
      ra[i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) *reinterpret_cast<f32x4*>(&As[buf * A_BUF + a_lds + i * RPI * SP_LSA]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) *reinterpret_cast<u32x4*>(&Bs[buf * B_BUF + b_lds[i]]) = rbv[i];
  };

  f32x16 lead[TM][TN], corr[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) lead[tm][tn][r] = corr[tm][tn][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * (32 * TM) + lr) * SP_LSA + 8 * lh];
  const uint16_t* b_frag = &Bs[lr * SP_LSB + 8 * lh];
  auto compute = [&](int buf) {
#pragma unroll
    for (int kg = 0; kg < SP_BK / 16; ++kg) {
      typename S::vec bp[TN][NP];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int q = 0; q < NP; ++q)
          bp[tn][q] = *reinterpret_cast<const typename S::vec*>(
              b_frag + buf * B_BUF + ((DIAG & 32) ? 0 : (q * SP_BN + tn * 32) * SP_LSB + kg * 16));
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + tm * 32 * SP_LSA + kg * 16);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + tm * 32 * SP_LSA + kg * 16 + 4);
        typename S::vec ap[NP];
        if constexpr (DIAG & 8) {
#pragma unroll
          for (int q = 0; q < NP; ++q) ap[q] = __builtin_bit_cast(typename S::vec, q & 1 ? x1 : x0);
        } else {
          S::split(x0, x1, ap);
        }
        if constexpr (DIAG & 4) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
              const f32x4 u = __builtin_bit_cast(f32x4, ap[q]), w = __builtin_bit_cast(f32x4, bp[tn][q]);
              lead[tm][tn][q] += u[0] * w[1] + u[2] * w[3];  // (keeps the operands alive)
            }
        } else {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) S::products(ap, bp[tn], lead[tm][tn], corr[tm][tn]);
        }
      }
    }
  };

  // ---- K loop: LDS double-buffered, the next step's operands in registers while this step multiplies ------------------
  if (nfast > 0) {
    load_a_fast(0);
    load_b(0);
    store_stage(0);
    if (nfast > 1) {
      load_a_fast(1);
      load_b(1);
    }
    __syncthreads();
    for (int t = 0; t < nfast; ++t) {
      const int cur = t & 1;
      compute(cur);
      if constexpr (!(DIAG & 2))
        if (t + 1 < nfast) store_stage(cur ^ 1);  // (last read in step t - 1, behind that step's barrier)
      if constexpr (!(DIAG & 1))
        if (t + 2 < nfast) {
          load_a_fast(t + 2);
          load_b(t + 2);
        }
      if constexpr (!(DIAG & 16)) __syncthreads();
    }
  }
  for (int t = nfast; t < ntiles; ++t) {
    load_a_ragged(t);
    load_b(t);
    __syncthreads();  // (every wave is done with both LDS buffers)
    store_stage(0);
    __syncthreads();
    compute(0);
  }

  // ---- epilogue: the launcher admits the plain one only (alpha, bias, row mask -- what feats_embed needs) -----------------
  // D row = (reg & 3) + 8 (reg >> 2) + 4 lh, col = lr of each 32 x 32 tile.  The few columns of k-source 1 (context, K1 <= 8)
  // are added to the finished sums as plain fp32 fused multiply-adds, exact like the default path's.
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row_w = row0 + wave * (32 * TM) + tm * 32;
    int rid[16];
    const float* a1p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gr = min(row_w + (r & 3) + 8 * (r >> 2) + 4 * lh, sg.rows - 1);
      rid[r] = D.mask_rows ? sg.ids[gr] : 1;
      const int ub = gr / sg.T, ut = gr - ub * sg.T;
      a1p[r] = D.K1 > 0 ? sg.a1 + (sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1)
                        : nullptr;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + tn * 32 + lr;
      const int nc = min(n, D.N - 1);
      const float bias = D.bias ? D.bias[nc] : 0.f;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = S::combine(lead[tm][tn][r], corr[tm][tn][r]);
      for (int k = 0; k < D.K1; ++k) {
        const float wv = D.bt1[(size_t)nc * D.ldb1 + k];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaf(a1p[r][k], wv, v[r]);
      }
      if (n < D.ncols_out) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row >= sg.rows) continue;
          float o = 0.f;
          if (n < D.N) {
            o = (D.alpha != 0.f ? D.alpha * v[r] : v[r]) + bias;
            if (D.mask_rows) o = rid[r] != 0 ? o : 0.f;  // e * mask (carca.py:94): exact zeros
          }
          sg.c[(size_t)row * D.ldc + n] = o;
        }
      }
    }
  }
}

// the packed planes a caller prepared for a weight matrix (carca_split_bind): thread-local, one binding
struct SplitBinding {
  const float* w;
  const void* planes;
  int mode, N, K0;
};
thread_local SplitBinding g_split_bound = {nullptr, nullptr, 0, 0, 0};
long long g_split_launches = 0;  // launches of the split-precision kernel so far (tests: the path was really taken)

template <int MODE>
int launch_pack(const float* bt0, int ldb0, int K0, int N, void* out, hipStream_t stream) {
  const int ncb = (N + SP_BN - 1) / SP_BN, ntiles = (K0 + SP_BK - 1) / SP_BK;
  const long threads = (long)ncb * ntiles * (SP_BN * 4);
  hipLaunchKernelGGL(split_pack_kernel<MODE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, bt0, ldb0, K0, N, ncb,
                     ntiles, (uint16_t*)out);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

extern "C" long long carca_split_launch_count(void) { return g_split_launches; }

extern "C" long long carca_split_bytes(int N, int K0, int mode) {
  if (N < 1 || K0 < 1 || (mode != 1 && mode != 2)) return 0;
  const long long ncb = (N + SP_BN - 1) / SP_BN, ntiles = (K0 + SP_BK - 1) / SP_BK;
  return ncb * ntiles * (mode == 1 ? 3 : 2) * SP_PLANE * 2;
}

extern "C" int carca_split_pack(const float* w, int ldw, int K0, int N, int mode, void* out, void* stream) {
  CARCA_CHECK_ARG(w && out && N >= 1 && K0 >= 1 && ldw >= K0, "split_pack: bad geometry");
  CARCA_CHECK_ARG(mode == 1 || mode == 2, "split_pack: mode %d is neither 1 (bf16 x 3) nor 2 (fp16 x 2)", mode);
  return mode == 1 ? launch_pack<1>(w, ldw, K0, N, out, (hipStream_t)stream) : launch_pack<2>(w, ldw, K0, N, out, (hipStream_t)stream);
}

extern "C" int carca_split_bind(const float* w, const void* planes, int mode, int N, int K0) {
  CARCA_CHECK_ARG((w && planes && (mode == 1 || mode == 2)) || (!w && !planes), "split_bind: weight, planes and mode go together");
  g_split_bound = SplitBinding{w, planes, mode, N, K0};
  return CARCA_OK;
}

// CARCA_OK: launched.  1: not this kernel's product (the caller launches the fp32 kernels).  gemm_rows_choose (gemm.hip)
// has already admitted the one-workgroup-per-CU kernel: 32-bit operand offsets, K0 >= 64, a grid that suits 384 x 96 tiles.
int carca_gemm_rows_split_try(const CarcaGemmDesc* desc, hipStream_t stream) {
  const int mode = carca_tuning(CARCA_TUNE_SPLIT_GEMM) & 15;
  if (mode != 1 && mode != 2) return 1;
  // k-source 0 on the MFMA pipe in 32-wide K steps (16-byte groups: K0 a multiple of 4); k-source 1 (a handful of
  // context columns) as fp32 FMAs
  if (desc->K0 % 4 != 0 || desc->K0 < 4 || desc->K1 > 8) return 1;
  if (desc->colvec || desc->pos || desc->gate_scale != 0.f) return 1;  // (the plain epilogue only: alpha, bias, row mask)
  for (int s = 0; s < desc->nseg; ++s)
    if (desc->seg[s].add || desc->seg[s].gate || desc->seg[s].rowscale || desc->seg[s].add_pos) return 1;
  SplitDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (desc->seg[s].rows + SP_BM - 1) / SP_BM;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = (desc->ncols_out + SP_BN - 1) / SP_BN;
  if (desc->ncols_out != desc->N && (desc->ncols_out + SP_BN - 1) / SP_BN != (desc->N + SP_BN - 1) / SP_BN) return 1;
  const SplitBinding& b = g_split_bound;
  if (b.w == desc->bt0 && b.mode == mode && b.N == desc->N && b.K0 == desc->K0) {
    g.wp = (const uint16_t*)b.planes;
  } else {
    // nobody prepared this matrix: split it here, into stream scratch (7.9 / 11.9 MB at C2, ~8 us)
    const size_t bytes = (size_t)carca_split_bytes(desc->N, desc->K0, mode);
    void* buf = carca_stream_capturing(stream) ? carca_capture_alloc(stream, bytes, false, nullptr)
                                               : carca_stream_scratch(stream, CARCA_SCRATCH_SPLITW, bytes);
    if (!buf) return (int)hipErrorOutOfMemory;
    if (int rc = carca_split_pack(desc->bt0, desc->ldb0, desc->K0, desc->N, mode, buf, stream))
      return rc;
    g.wp = (const uint16_t*)buf;
  }
  const int grid = rb * g.ncb;
  hipEvent_t e0, e1;
  const bool ev = carca_take_launch_events(&e0, &e1);
  auto launch = [&](auto kernel) {
    if (ev) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(768 / SPLIT_TM), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(768 / SPLIT_TM), 0, stream, g);
  };
  const int diag = carca_tuning(CARCA_TUNE_DIAG);
#define SPLIT_CASE(D_) \
  case D_: mode == 1 ? launch(gemm_rows_split_kernel<1, SPLIT_TM, D_>) : launch(gemm_rows_split_kernel<2, SPLIT_TM, D_>); break;
  switch (diag) {
    SPLIT_CASE(1) SPLIT_CASE(2) SPLIT_CASE(3) SPLIT_CASE(4) SPLIT_CASE(8) SPLIT_CASE(12) SPLIT_CASE(16) SPLIT_CASE(19) SPLIT_CASE(32)
    SPLIT_CASE(7) SPLIT_CASE(15)
    default: mode == 1 ? launch(gemm_rows_split_kernel<1, SPLIT_TM, 0>) : launch(gemm_rows_split_kernel<2, SPLIT_TM, 0>);
  }
#undef SPLIT_CASE
  CARCA_LAUNCH_CHECK();
  ++g_split_launches;
  return CARCA_OK;
}
