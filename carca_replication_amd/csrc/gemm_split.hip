// Split-precision feature GEMM (opt-in, tuning key 16): AllEmbedding.feats_embed (carca.py:86), q = [a ; c] W_f^T + b_f,
// on the 16-bit MFMA pipe -- 16x the rate of the exact-fp32 MFMA the default path is bound by (MI355X_MICROARCH.md,
// Matrix cores) -- with fp32-class accuracy (hard part 1: "fp32 MFMA, or a bf16x3 split-accumulate scheme").
//
// Every fp32 operand x is written as a short sum of 16-bit parts, round-to-nearest each time, and the product of two
// operands as the sum of the part products that matter, all into ONE fp32 accumulator (a product of two 16-bit parts is
// exact in fp32; what remains is the fp32 accumulation the default path has as well, in 16-term groups instead of a chain):
//   mode 1, bf16 x 3:  x = p0 + p1 + p2 (+ <= 2^-24 |x|),  p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1);
//                      a.b ~= a0 b2 + a1 b1 + a2 b0 + a0 b1 + a1 b0 + a0 b0       six products, dropped terms 2^-24 relative;
//   mode 2, fp16 x 2:  x = h0 + h1 (+ <= 2^-22 |x|),       h0 = fp16(x), h1 = fp16(x - h0);
//                      a.b ~= a0 b1 + a1 b0 + a0 b0                               three products, dropped terms 2^-22 relative.
//                      fp16 has five exponent bits: |x| < 65504 is required, and a residual below 2^-14 is a DENORMAL fp16
//                      number, i.e. carried with absolute precision 2^-25.  W is therefore scaled by a power of two when it
//                      is packed (its largest entry to [2^14, 2^15): every residual that matters is a normal number; the
//                      epilogue scales back), A is taken as it is: attribute values of order 1 (BASELINE.md draws U[0, 1))
//                      keep 2^-25 absolute = fp32's own resolution at 0.5; data of a much smaller scale should use mode 1.
//
// Two kernels on gemm_rows_cu_kernel's grid (gemm.hip: ONE workgroup per CU, tile 384 x 96, K step 32, twelve waves, wave w
// owns rows 32 w .. 32 w + 31 x 96 columns = three 32 x 32 tiles of v_mfma_f32_32x32x16_{bf16,f16}: lane (r = l & 31,
// h = l >> 5) supplies A[row r][k = 8 h .. 8 h + 7] and Bt[col r][k = 8 h .. 8 h + 7]; C / D as the fp32 MFMA's):
//   gemm_rows_split_dma_kernel  both operands staged by LDS-DMA, fragment traffic software-pipelined: the fast path;
//   gemm_rows_split_kernel      register staging, any K0 % 4 == 0 and any alignment: ragged shapes (and the A/B baseline).
// A stays fp32 in HBM (the caller's tensor) and in LDS; a wave splits the eight values of its fragment in registers right
// behind the fragment read, once per (32 rows x 16 k), reused over its three column tiles.  W_f is split ONCE per weight
// version into packed planes (carca_split_pack) laid out in the order a workgroup stages them.
#include <hip/hip_ext.h>
#include "carca_common.h"
#include "../../include/carca_hip.h"
#include <type_traits>

namespace {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));

// The split of a fragment is written instruction by instruction: the SIMD's vector issue is what bounds these kernels
// (PMC: 5 VALU instructions per MFMA in the compiler's version of mode 2, 15 % of the MFMA cycles co-executing with VALU,
// VALU + MFMA issue = the kernel's length), and hipcc packed half of the arithmetic into v_pk_* f32 instructions, which
// cost several plain ones beside MFMAs (MI355X_MICROARCH.md, price of one filler).
template <int MODE>
struct Split;
template <>
struct Split<1> {  // bf16 x 3
  static constexpr int NP = 3;
  typedef b16x8 vec;
  // device hot path: two values -> three packed pairs (11 VALU instructions: 5.5 per value).  The conversions are the
  // compiler's (v_cvt_pk_bf16_f32 from the vector cast): written as asm too, an MFMA that read a converted pair right behind
  // it saw stale registers in the ragged-K tail of the register-staged kernel (garbage of 1e36 in rows that changed from run
  // to run) -- hipcc's hazard recognizer pads its own instructions, nothing inside or behind an asm statement.  The
  // subtractions stay asm: as C they become v_pk_add_f32.
  static __device__ __forceinline__ void split_pair(float v0, float v1, unsigned& p0, unsigned& p1, unsigned& p2) {
    typedef __bf16 b16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float r0, r1, s0, s1;
    const b16x2 a = __builtin_convertvector(f32x2{v0, v1}, b16x2);
    p0 = __builtin_bit_cast(unsigned, a);
    const float t0 = __uint_as_float(p0 << 16), t1 = __uint_as_float(p0 & 0xffff0000u);
    asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(v0), "v"(t0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(v1), "v"(t1));
    const b16x2 b = __builtin_convertvector(f32x2{r0, r1}, b16x2);
    p1 = __builtin_bit_cast(unsigned, b);
    const float u0 = __uint_as_float(p1 << 16), u1 = __uint_as_float(p1 & 0xffff0000u);
    asm("v_sub_f32 %0, %1, %2" : "=v"(s0) : "v"(r0), "v"(u0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(s1) : "v"(r1), "v"(u1));
    const b16x2 c = __builtin_convertvector(f32x2{s0, s1}, b16x2);
    p2 = __builtin_bit_cast(unsigned, c);
  }
  // the same arithmetic in C (the pack kernel; every conversion rounds to nearest even, as v_cvt_pk_bf16_f32 does)
  static __device__ __forceinline__ void split_value(float v, float, uint16_t (&p)[3]) {
    const __bf16 b0 = (__bf16)v;
    const float r1 = v - (float)b0;
    const __bf16 b1 = (__bf16)r1;
    const __bf16 b2 = (__bf16)(r1 - (float)b1);
    p[0] = __builtin_bit_cast(uint16_t, b0);
    p[1] = __builtin_bit_cast(uint16_t, b1);
    p[2] = __builtin_bit_cast(uint16_t, b2);
  }
  static __device__ __forceinline__ f32x16 mfma(const vec a, const vec b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void products(const vec (&a)[3], const vec (&b)[3], f32x16& acc) {
    acc = mfma(a[0], b[2], acc);  // (smallest first)
    acc = mfma(a[1], b[1], acc);
    acc = mfma(a[2], b[0], acc);
    acc = mfma(a[0], b[1], acc);
    acc = mfma(a[1], b[0], acc);
    acc = mfma(a[0], b[0], acc);
  }
};
template <>
struct Split<2> {  // fp16 x 2
  static constexpr int NP = 2;
  typedef h16x8 vec;
  // two values -> two packed pairs, 4 VALU instructions: the residual x - h0 is ONE v_fma_mix_f32 (the f16 half as an operand;
  // asm, hipcc does not select it), the conversions are the compiler's (see Split<1>)
  static __device__ __forceinline__ void split_pair(float v0, float v1, unsigned& p0, unsigned& p1, unsigned&) {
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float r0, r1;
    const h16x2 a = __builtin_convertvector(f32x2{v0, v1}, h16x2);
    p0 = __builtin_bit_cast(unsigned, a);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p0), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p0), "v"(v1));
    const h16x2 b = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    p1 = __builtin_bit_cast(unsigned, b);
  }
  static __device__ __forceinline__ void split_value(float v, float scale, uint16_t (&p)[2]) {
    const float s = v * scale;  // (a power of two: exact)
    const _Float16 h0 = (_Float16)s;
    const _Float16 h1 = (_Float16)(s - (float)h0);
    p[0] = __builtin_bit_cast(uint16_t, h0);
    p[1] = __builtin_bit_cast(uint16_t, h1);
  }
  static __device__ __forceinline__ f32x16 mfma(const vec a, const vec b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void products(const vec (&a)[2], const vec (&b)[2], f32x16& acc) {
    acc = mfma(a[0], b[1], acc);
    acc = mfma(a[1], b[0], acc);
    acc = mfma(a[0], b[0], acc);
  }
};
// the eight fp32 values of a lane's fragment -> NP vectors of eight 16-bit parts
template <int MODE, int DIAG>
__device__ __forceinline__ void split_fragment(const f32x4 x0, const f32x4 x1, typename Split<MODE>::vec (&p)[Split<MODE>::NP]) {
  using S = Split<MODE>;
  if constexpr (DIAG & 8) {  // (timing experiment: no arithmetic, the bits reinterpreted)
#pragma unroll
    for (int q = 0; q < S::NP; ++q) p[q] = __builtin_bit_cast(typename S::vec, q & 1 ? x1 : x0);
  } else {
    u32x4 w[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v0 = j < 2 ? x0[2 * j] : x1[2 * j - 4], v1 = j < 2 ? x0[2 * j + 1] : x1[2 * j - 3];
      unsigned a = 0, b = 0, c = 0;
      S::split_pair(v0, v1, a, b, c);
      w[0][j] = a;
      w[1][j] = b;
      w[2][j] = c;
    }
#pragma unroll
    for (int q = 0; q < S::NP; ++q) p[q] = __builtin_bit_cast(typename S::vec, w[q]);
  }
}

constexpr int SP_BM = 384, SP_BN = 96, SP_BK = 32, SP_NT = 768;
constexpr int SP_LSA = SP_BK + 4;   // register-staged kernel: floats per A row in LDS (conflict-free 16-byte fragment reads)
constexpr int SP_LSB = SP_BK + 8;   // ... 16-bit values per Bt row in LDS (80 bytes)
constexpr int SP_PLANE = SP_BN * SP_BK;  // 16-bit values of one packed (column block, K step, part): 6 KB
constexpr int SP_HEADER = 256;      // bytes in front of the packed planes: [0] max |w| (bits), [1] scale, [2] 1 / scale

struct SplitDev {
  CarcaGemmDesc d;
  int rb_start[CARCA_MAX_SEGS + 1];
  int nrb, ncb;
  const uint16_t* wp;    // packed planes [ncb][ntiles][NP][96][32]
  const float* header;   // the packed buffer's header (scale of the planes)
};

// ---- packing: W (fp32, [N, ldw], the K0 columns of k-source 0) -> [header | planes] -------------------------------------
__global__ __launch_bounds__(256) void split_absmax_kernel(const float* __restrict__ w, int ldw, int K0, int N, unsigned* header) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)N * K0; i += (long)gridDim.x * 256)
    m = fmaxf(m, fabsf(w[(size_t)(i / K0) * ldw + i % K0]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(header, __float_as_uint(m));  // (non-negative floats order like their bits)
}
// One thread = 8 consecutive k of one row.  Inside a row's 64 bytes the four 16-byte groups sit XOR-swizzled, group c at
// position c ^ ((n >> 2) & 3): a (step, part) plane copied LINEARLY into LDS (the LDS-DMA kernel) is then read conflict-free
// by the B fragment reads (16 lanes = 16 rows, same group: bank (n & 3) 16 + 4 position -- distinct over a lane group's rows).
template <int MODE>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, int ldw, int K0, int N, int ncb,
                                                         int ntiles, float* __restrict__ header, uint16_t* __restrict__ out) {
  using S = Split<MODE>;
  float scale = 1.0f;
  if (MODE == 2) {
    // the largest |w| to [2^14, 2^15): frexp gives m in [0.5, 1) x 2^e
    const float amax = __uint_as_float(reinterpret_cast<const unsigned*>(header)[0]);
    int e = 0;
    if (amax > 0.f && amax < 3.0e38f) (void)frexpf(amax, &e);
    scale = amax > 0.f ? ldexpf(1.0f, max(-100, min(100, 15 - e))) : 1.0f;
  }
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx == 0) {
    header[1] = scale;
    header[2] = 1.0f / scale;
  }
  if (idx >= (long)ncb * ntiles * (SP_BN * 4)) return;
  const int c = (int)(idx & 3), n = (int)((idx >> 2) % SP_BN);
  const long ct = idx / (SP_BN * 4);
  const int t = (int)(ct % ntiles), cb = (int)(ct / ntiles);
  const int gn = cb * SP_BN + n;
  const int k0 = t * SP_BK + c * 8;
  uint16_t parts[S::NP][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (gn < N && k0 + j < K0) ? w[(size_t)gn * ldw + k0 + j] : 0.f;
    uint16_t p[S::NP];
    S::split_value(v, scale, p);
#pragma unroll
    for (int q = 0; q < S::NP; ++q) parts[q][j] = p[q];
  }
#pragma unroll
  for (int q = 0; q < S::NP; ++q) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (unsigned)parts[q][2 * j] | ((unsigned)parts[q][2 * j + 1] << 16);
    *reinterpret_cast<u32x4*>(out + ((size_t)(ct * S::NP + q) * SP_PLANE + (n * 4 + (c ^ ((n >> 2) & 3))) * 8)) = v;
  }
}

// ---- the epilogue both kernels share: the launcher admits the plain one only (alpha, bias, row mask -- what feats_embed needs).
// D row = (reg & 3) + 8 (reg >> 2) + 4 lh, col = lr of each 32 x 32 tile.  The few columns of k-source 1 (context, K1 <= 8)
// are added to the finished sums as plain fp32 fused multiply-adds, exact like the default path's.
__device__ __forceinline__ void split_epilogue(const CarcaGemmDesc& D, const CarcaGemmSeg& sg, const f32x16 (&acc)[3], float inv_scale,
                                               int n0, int row_w, int lr, int lh) {
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
  for (int tn = 0; tn < 3; ++tn) {
    const int n = n0 + tn * 32 + lr;
    const int nc = min(n, D.N - 1);
    const float bias = D.bias ? D.bias[nc] : 0.f;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[tn][r] * inv_scale;
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

struct SplitWhere {  // block id -> (row block, column block, segment): column blocks of a row block on one XCD (gemm.hip)
  int rb, cb, s, row0, n0;
};
__device__ __forceinline__ SplitWhere split_where(const SplitDev& args) {
  const int id = blockIdx.x, total = args.nrb * args.ncb;
  const int xcd = id & 7, q8 = total >> 3, r8 = total & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  SplitWhere w;
  w.rb = wg / args.ncb;
  w.cb = wg - w.rb * args.ncb;
  w.s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < args.d.nseg && w.rb >= args.rb_start[i]) w.s = i;
  w.row0 = (w.rb - args.rb_start[w.s]) * SP_BM;
  w.n0 = w.cb * SP_BN;
  return w;
}

// DIAG (tuning key 15, timing experiments with WRONG results; 0 in every shipped launch): 1 no global loads / DMA behind the
// prologue, 4 no MFMAs, 8 no split arithmetic (the fragment's bits reinterpreted), 16 no barriers in the K loop, 32 no B
// fragment addresses (one group re-read), 64 the DMA spread over the step (A/B), 128 no wait for the DMA.
// -------------------------------------------------------------------------------------------------------------------------
// gemm_rows_split_kernel: register staging (buffer loads -> registers -> ds_write), LDS double-buffered, padded rows.
template <int MODE, int DIAG = 0>
__global__ __launch_bounds__(768) void gemm_rows_split_kernel(const SplitDev args) {
  using S = Split<MODE>;
  typedef typename S::vec vec;
  constexpr int NP = S::NP, TN = 3, C4 = SP_BK / 4, RPI = SP_NT / 8;
  constexpr int A_PER = SP_BM * C4 / SP_NT;                                        // 4 sixteen-byte slots of the A tile per thread
  constexpr int B_SLOTS = NP * SP_BN * 4, B_PER = (B_SLOTS + SP_NT - 1) / SP_NT;  // 768 / 1152 slots: 1 or 2 per thread
  constexpr int A_BUF = SP_BM * SP_LSA, B_BUF = NP * SP_BN * SP_LSB;
  __shared__ __attribute__((aligned(16))) float As[2 * A_BUF];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2 * B_BUF];

  const CarcaGemmDesc& D = args.d;
  const SplitWhere W = split_where(args);
  const CarcaGemmSeg sg = D.seg[W.s];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nfast = D.K0 / SP_BK;                 // whole K steps: the pipelined loop
  const int ntiles = (D.K0 + SP_BK - 1) / SP_BK;  // (+ one ragged step when K0 % 32 != 0; k-source 1 joins in the epilogue)

  // ---- staging slots (loop invariants): slot tid + 768 i = row (tid >> 3) + 96 i, 16-byte column tid & 7 ---------------------
  const int a_r = tid >> 3, a_c4 = tid & 7;
  unsigned a_byte[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int gr = min(W.row0 + a_r + RPI * i, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t off = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
                       : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                                       : (size_t)gr * D.lda0;
    a_byte[i] = (unsigned)((off + a_c4 * 4) * sizeof(float));
  }
  const int a_lds = a_r * SP_LSA + a_c4 * 4;  // (+ 96 i rows: an immediate)
  int b_lds[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int slot = min(tid + i * SP_NT, B_SLOTS - 1);
    const int part = slot / (SP_BN * 4), rem = slot - part * (SP_BN * 4);
    b_lds[i] = (part * SP_BN + (rem >> 2)) * SP_LSB + ((rem & 3) ^ ((rem >> 4) & 3)) * 8;  // (the packed planes' swizzle undone)
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = carca_rsrc(sg.a0);
  // (the packed planes of this column block; the pointer goes through readfirstlane: left to the compiler the resource ended up
  // in vector registers and every B load became a waterfall loop)
  const unsigned long long wp_u = (unsigned long long)(args.wp + (size_t)W.cb * ntiles * NP * SP_PLANE);
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
    // are requested from a clamped address (the row's last group) and stored as zeros
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int k = t * SP_BK + a_c4 * 4;
      const int kc = min(k, D.K0 - 4);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_byte[i] + (unsigned)((kc - a_c4 * 4) * (int)sizeof(float)), 0, 0);
      ra[i] = k < D.K0 ? f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])}
                       : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) *reinterpret_cast<f32x4*>(&As[buf * A_BUF + a_lds + i * RPI * SP_LSA]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) *reinterpret_cast<u32x4*>(&Bs[buf * B_BUF + b_lds[i]]) = rbv[i];
  };

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * SP_LSA + 8 * lh];
  const uint16_t* b_frag = &Bs[lr * SP_LSB + 8 * lh];
  auto compute = [&](int buf) {
#pragma unroll
    for (int kg = 0; kg < SP_BK / 16; ++kg) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + kg * 16);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + kg * 16 + 4);
      vec ap[NP];
      split_fragment<MODE, DIAG>(x0, x1, ap);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        vec bp[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q)
          bp[q] = *reinterpret_cast<const vec*>(b_frag + buf * B_BUF + (q * SP_BN + tn * 32) * SP_LSB + kg * 16);
        if constexpr (DIAG & 4) {
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            const f32x4 u = __builtin_bit_cast(f32x4, ap[q]), w = __builtin_bit_cast(f32x4, bp[q]);
            acc[tn][q] += u[0] * w[1] + u[2] * w[3];  // (keeps the operands alive)
          }
        } else {
          S::products(ap, bp, acc[tn]);
        }
      }
    }
  };

  // ---- K loop: the next step's operands in registers while this step multiplies ------------------------------------------
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
  split_epilogue(D, sg, acc, args.header[2], W.n0, W.row0 + wave * 32, lr, lh);
}

// -------------------------------------------------------------------------------------------------------------------------
// gemm_rows_split_dma_kernel: the same product with the staging done by LDS-DMA and the fragment traffic software-pipelined.
// What the register-staged kernel measured (tools/split_probe.py, C2, fp16 x 2 with two accumulators: 262 us; its parts
// ablated): the LDS writes of the staging path 58 us (ds_write_b128 moves 79 B per clock and CU), the fragment reads 25-90 us,
// the split arithmetic 34 us, the MFMAs 95 us -- and they ADD UP: after every barrier the three waves of a SIMD read, split
// and multiply in the same phases, and VALU and MFMA instructions share the SIMD's issue.  Here
//  * both operands reach LDS by buffer_load ... lds (no registers, no ds_write): A as pieces of 8 rows x 128 bytes whose
//    16-byte groups are XOR-swizzled through the SOURCE address (group c of row r at position c ^ ((r >> 1) & 7): the
//    fragment reads of a lane group cover all 64 banks once: SQ_LDS_BANK_CONFLICT = 0), the packed W planes as straight
//    1 KB copies (swizzled when they were packed);
//  * a wave requests the operands of unit (kg, tn) + 1 before it multiplies unit (kg, tn), and splits the next k group's
//    fragment in the shadow of this one's MFMAs;
//  * the K loop is unrolled by two: every LDS address is a register plus an immediate;
//  * one barrier per K step; the step's DMA is issued behind the first fragment requests and has the whole step to land
//    (s_waitcnt vmcnt(0) in front of the barrier that publishes it).
template <int MODE, int DIAG = 0>
__global__ __launch_bounds__(768) void gemm_rows_split_dma_kernel(const SplitDev args) {
  using S = Split<MODE>;
  typedef typename S::vec vec;
  constexpr int NP = S::NP, TN = 3;
  constexpr int A_BYTES = SP_BM * SP_BK * 4, B_BYTES = NP * SP_PLANE * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PIECES = A_BYTES / 1024 / 12;        // 4 per wave
  constexpr int B_PIECES = B_BYTES / 1024;             // 12 / 18: one per wave (+ one more for the first six)
  __shared__ __attribute__((aligned(1024))) char Sm[2 * STAGE];

  const CarcaGemmDesc& D = args.d;
  const SplitWhere W = split_where(args);
  const CarcaGemmSeg sg = D.seg[W.s];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsteps = D.K0 / SP_BK;  // (the launcher sends K0 % 32 != 0 to the register-staged kernel)

  // ---- DMA sources: A piece p = 4 wave + i holds rows 8 p .. 8 p + 7; lane l fills position (l & 7) of row (l >> 3) --------
  unsigned a_src[A_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int r = 8 * (A_PIECES * wave + i) + (lane >> 3);  // row inside the tile
    const int gr = min(W.row0 + r, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t off = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
                       : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                                       : (size_t)gr * D.lda0;
    a_src[i] = (unsigned)(off * sizeof(float)) + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = carca_rsrc(sg.a0);
  const unsigned long long wp_u = (unsigned long long)(args.wp + (size_t)W.cb * nsteps * NP * SP_PLANE);
  const unsigned wp_lo = __builtin_amdgcn_readfirstlane((unsigned)wp_u), wp_hi = __builtin_amdgcn_readfirstlane((unsigned)(wp_u >> 32));
  const __amdgpu_buffer_rsrc_t b_rsrc = carca_rsrc((const void*)(((unsigned long long)wp_hi << 32) | wp_lo));
  const unsigned b_src = (unsigned)(lane * 16);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // piece i of this wave's share of step t: 0 .. 3 its four A pieces, 4 its B piece, 5 the second B piece of the first six
  // waves (bf16 x 3: 18 pieces)
  auto stage_piece = [&](int i, int t, int buf) {
    char* base = Sm + buf * STAGE;
    if (i < A_PIECES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr)(base + (A_PIECES * wave + i) * 1024), 16, a_src[i < A_PIECES ? i : 0],
                                               t * (SP_BK * 4), 0, 0);
    else if (i == A_PIECES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr)(base + A_BYTES + wave * 1024), 16, b_src,
                                               t * B_BYTES + wave * 1024, 0, 0);
    else if (B_PIECES > 12 && wave < B_PIECES - 12)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr)(base + A_BYTES + (12 + wave) * 1024), 16, b_src,
                                               t * B_BYTES + (12 + wave) * 1024, 0, 0);
  };
  auto stage = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < A_PIECES + 2; ++i) stage_piece(i, t, buf);
  };

  // ---- fragment addresses (LDS pointers into stage 0; stage 1 = + STAGE, an immediate): lane (lr, lh) -----------------------
  const int lr = lane & 31, lh = lane >> 5;
  const int arow = wave * 32 + lr, asw = (arow >> 1) & 7;
  const char* a_at[2][2];  // [kg][half of the 32-byte fragment]
#pragma unroll
  for (int kg = 0; kg < 2; ++kg)
#pragma unroll
    for (int e = 0; e < 2; ++e) a_at[kg][e] = Sm + arow * 128 + (((4 * kg + 2 * lh + e) ^ asw) * 16);
  const char* b_at[2];     // [kg]: row lr of part 0, tile 0; (part q, tile tn) = + (q 96 + tn 32) 64 bytes -- (n >> 2) & 3 does not depend on tn
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) b_at[kg] = Sm + A_BYTES + lr * 64 + (((2 * kg + lh) ^ ((lr >> 2) & 3)) * 16);

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  auto mul = [&](const vec (&a)[NP], const vec (&b)[NP], int tn) {
    if constexpr (DIAG & 4) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const f32x4 u = __builtin_bit_cast(f32x4, a[q]), w = __builtin_bit_cast(f32x4, b[q]);
        acc[tn][q] += u[0] * w[1] + u[2] * w[3];
      }
    } else {
      S::products(a, b, acc[tn]);
    }
  };
#define SPLIT_PIN() __builtin_amdgcn_sched_barrier(0)
  // one K step on LDS stage CUR (compile-time): step t multiplies, step t + 1 is requested
  auto step = [&](auto cur_tag, int t) {
    constexpr int CUR = decltype(cur_tag)::value, OFF = CUR * STAGE;
    auto read_a = [&](int kg, f32x4& x0, f32x4& x1) {
      x0 = *reinterpret_cast<const f32x4*>(a_at[kg][0] + OFF);
      x1 = *reinterpret_cast<const f32x4*>(a_at[kg][1] + OFF);
    };
    auto read_b = [&](int kg, int tn, vec (&b)[NP]) {
#pragma unroll
      for (int q = 0; q < NP; ++q)
        b[q] = *reinterpret_cast<const vec*>(b_at[kg] + OFF + ((DIAG & 32) ? 0 : (q * SP_BN + tn * 32) * 64));
    };
    f32x4 x0, x1, y0, y1;
    vec ap0[NP], ap1[NP], b0[NP], b1[NP];
    // the step's first operands, then the NEXT step's DMA (issued while those reads are in flight)
    read_a(0, x0, x1);
    read_b(0, 0, b0);
    SPLIT_PIN();
    const bool more = !(DIAG & 1) && t + 1 < nsteps;
    if constexpr (!(DIAG & 64))
      if (more) stage(t + 1, CUR ^ 1);  // (that stage was last read in step t - 1, behind its barrier)
    SPLIT_PIN();
    read_b(0, 1, b1);
    split_fragment<MODE, DIAG>(x0, x1, ap0);
    read_a(1, y0, y1);
    SPLIT_PIN();
    // (DIAG 64, A/B: the DMA spread over the step, two pieces behind each of the first three units, instead of all up front)
    mul(ap0, b0, 0);            // unit (0, 0)
    read_b(0, 2, b0);
    if constexpr (DIAG & 64) if (more) { stage_piece(0, t + 1, CUR ^ 1); stage_piece(1, t + 1, CUR ^ 1); }
    SPLIT_PIN();
    mul(ap0, b1, 1);            // unit (0, 1)
    read_b(1, 0, b1);
    split_fragment<MODE, DIAG>(y0, y1, ap1);  // (its arithmetic in the shadow of this unit's MFMAs)
    if constexpr (DIAG & 64) if (more) { stage_piece(2, t + 1, CUR ^ 1); stage_piece(3, t + 1, CUR ^ 1); }
    SPLIT_PIN();
    mul(ap0, b0, 2);            // unit (0, 2)
    read_b(1, 1, b0);
    if constexpr (DIAG & 64) if (more) { stage_piece(4, t + 1, CUR ^ 1); stage_piece(5, t + 1, CUR ^ 1); }
    SPLIT_PIN();
    mul(ap1, b1, 0);            // unit (1, 0)
    read_b(1, 2, b1);
    SPLIT_PIN();
    mul(ap1, b0, 1);            // unit (1, 1)
    SPLIT_PIN();
    mul(ap1, b1, 2);            // unit (1, 2)
    SPLIT_PIN();
    if constexpr (!(DIAG & 128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of step t + 1 has landed
    if constexpr (!(DIAG & 16)) __builtin_amdgcn_s_barrier();
    SPLIT_PIN();
  };

  if (nsteps > 0) stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int t = 0;
  for (; t + 1 < nsteps; t += 2) {
    step(std::integral_constant<int, 0>{}, t);
    step(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t < nsteps) step(std::integral_constant<int, 0>{}, t);
#undef SPLIT_PIN
  split_epilogue(D, sg, acc, args.header[2], W.n0, W.row0 + wave * 32, lr, lh);
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
int launch_pack(const float* w, int ldw, int K0, int N, void* out, hipStream_t stream) {
  const int ncb = (N + SP_BN - 1) / SP_BN, ntiles = (K0 + SP_BK - 1) / SP_BK;
  const long threads = (long)ncb * ntiles * (SP_BN * 4);
  float* header = (float*)out;
  if (hipMemsetAsync(out, 0, SP_HEADER, stream) != hipSuccess) return CARCA_ERR_BADARG;
  if (MODE == 2) hipLaunchKernelGGL(split_absmax_kernel, dim3(256), dim3(256), 0, stream, w, ldw, K0, N, (unsigned*)header);
  hipLaunchKernelGGL(split_pack_kernel<MODE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, w, ldw, K0, N, ncb,
                     ntiles, header, (uint16_t*)((char*)out + SP_HEADER));
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

}  // namespace

extern "C" long long carca_split_launch_count(void) { return g_split_launches; }

extern "C" long long carca_split_bytes(int N, int K0, int mode) {
  if (N < 1 || K0 < 1 || (mode != 1 && mode != 2)) return 0;
  const long long ncb = (N + SP_BN - 1) / SP_BN, ntiles = (K0 + SP_BK - 1) / SP_BK;
  return SP_HEADER + ncb * ntiles * (mode == 1 ? 3 : 2) * SP_PLANE * 2;
}

extern "C" int carca_split_pack(const float* w, int ldw, int K0, int N, int mode, void* out, void* stream) {
  CARCA_CHECK_ARG(w && out && N >= 1 && K0 >= 1 && ldw >= K0, "split_pack: bad geometry");
  CARCA_CHECK_ARG(mode == 1 || mode == 2, "split_pack: mode %d is neither 1 (bf16 x 3) nor 2 (fp16 x 2)", mode);
  CARCA_CHECK_ARG(((uintptr_t)out & 255) == 0, "split_pack: the packed buffer must be 256-byte aligned");
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
  if (desc->colvec || desc->pos || desc->add_table) return 1;  // (the plain epilogue only: alpha, bias, row mask; gate_scale acts through seg.gate)
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
  const char* packed;
  if (b.w == desc->bt0 && b.mode == mode && b.N == desc->N && b.K0 == desc->K0) {
    packed = (const char*)b.planes;
  } else {
    // nobody prepared this matrix: split it here, into stream scratch (7.9 / 11.9 MB at C2, ~8 us)
    const size_t bytes = (size_t)carca_split_bytes(desc->N, desc->K0, mode);
    void* buf = carca_stream_capturing(stream) ? carca_capture_alloc(stream, bytes, false, nullptr)
                                               : carca_stream_scratch(stream, CARCA_SCRATCH_SPLITW, bytes);
    if (!buf) return (int)hipErrorOutOfMemory;
    if (int rc = carca_split_pack(desc->bt0, desc->ldb0, desc->K0, desc->N, mode, buf, stream)) return rc;
    packed = (const char*)buf;
  }
  g.header = (const float*)packed;
  g.wp = (const uint16_t*)(packed + SP_HEADER);
  const int grid = rb * g.ncb;
  hipEvent_t e0, e1;
  const bool ev = carca_take_launch_events(&e0, &e1);
  // the LDS-DMA kernel where every row of k-source 0 is whole 16-byte groups at 16-byte addresses and K0 whole K steps;
  // tuning variant 21 = the register-staged kernel everywhere (A/B)
  const int diag = carca_tuning(CARCA_TUNE_DIAG);
  bool dma = desc->K0 % SP_BK == 0 && desc->lda0 % 4 == 0 && carca_tuning(CARCA_TUNE_GEMM_VARIANT) != 21;
  for (int s = 0; s < desc->nseg && dma; ++s)
    dma = ((uintptr_t)desc->seg[s].a0 & 15) == 0 && desc->seg[s].a0_bstride % 4 == 0;
  carca_rows_log(dma ? (mode == 1 ? "gemm_rows_split_dma_kernel<bf16x3>" : "gemm_rows_split_dma_kernel<fp16x2>")
                     : (mode == 1 ? "gemm_rows_split_kernel<bf16x3>" : "gemm_rows_split_kernel<fp16x2>"), desc, grid);
  auto launch = [&](auto kernel) {
    if (ev) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(768), 0, stream, g);
  };
#define SPLIT_CASE(D_)                                                                                          \
  case D_:                                                                                                      \
    if (dma) mode == 1 ? launch(gemm_rows_split_dma_kernel<1, D_>) : launch(gemm_rows_split_dma_kernel<2, D_>); \
    else mode == 1 ? launch(gemm_rows_split_kernel<1, D_>) : launch(gemm_rows_split_kernel<2, D_>);             \
    break;
  switch (diag) {
    SPLIT_CASE(1) SPLIT_CASE(4) SPLIT_CASE(5) SPLIT_CASE(8) SPLIT_CASE(12) SPLIT_CASE(13) SPLIT_CASE(16) SPLIT_CASE(32)
    SPLIT_CASE(64) SPLIT_CASE(68) SPLIT_CASE(128) SPLIT_CASE(192)
    default: SPLIT_CASE(0)
  }
#undef SPLIT_CASE
  CARCA_LAUNCH_CHECK();
  ++g_split_launches;
  return CARCA_OK;
}
