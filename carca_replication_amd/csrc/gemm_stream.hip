// gemm_rows_cus_kernel: carca_gemm_rows (gemm.hip) for a SHORT K with MANY tiles per CU -- AllEmbedding.feats_embed
// (carca.py:86) at BASELINE's C5 (n_attrs = 512: K0 = 16 K steps of 32; B (L + 1001) = 134,528 rows: 1,755 tiles of
// 384 x 96 for 256 CUs).  gemm_rows_cu_kernel runs one tile per workgroup: every tile pays a workgroup start (~9 us to the
// first barrier: twelve 168-register waves), a pipeline fill (two round trips to HBM before the first MFMA), an unpipelined
// tail step for the K1 = 6 context columns (plain loads, two barriers, 48 MFMAs for 6 columns: ~7 us) and an epilogue,
// ~21 us on top of 65 us of MFMA work: 66 % of the fp32 MFMA peak.  Here ONE persistent workgroup per CU walks a column
// block down its share of the row blocks and treats its tiles' K steps as ONE stream of items:
//   * the loads run two items ahead of the MFMAs ACROSS tile boundaries -- while a tile's last steps multiply, the next
//     tile's first two K tiles are requested and stored, so a tile begins with its operands in LDS / registers;
//   * the context columns are one more item of the stream, 8 wide: each thread requests ONE 16-byte group of its row (the
//     last group clamped to end at K1 and shifted when stored, elements past K1 zeroed), the item costs one 8-k group of
//     MFMAs (12 instead of 48) and no barrier of its own;
//   * the epilogue is stores only (buffer stores: one lane-offset register, the row in the scalar offset; the pad mask of
//     a wave's 32 rows is ONE id per lane and a ballot), issued while the next tile's loads fly.
// Tiles and the K step itself are gemm_rows_cu_kernel's (384 x 96, twelve waves, wave w owns rows 32 w .. + 31 x 96
// columns, hand-scheduled step with LDS double buffer and one barrier); the narrow last column block (N = 450 = 4 x 96 +
// 66) is two MFMA column tiles + up to two VALU columns as in gemm_rows_sk_kernel; its (cheaper) tiles are dealt BEHIND the
// full ones -- fewer to the workgroups that hold one full tile more -- so that all workgroups end together.  Same products in the same order as gemm_rows_cu_kernel per output
// element (K steps in order, the context group last): results agree to the last bit where that kernel's tail tile adds its
// 32-wide step in the same grouping, and to round-off otherwise (tests/test_hip_gemm_stream.py: float64 products).
#include <hip/hip_ext.h>
#include <algorithm>
#include <type_traits>
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

struct StreamDev {
  CarcaGemmDesc d;
  int rb_start[CARCA_MAX_SEGS + 1];  // 384-row blocks in front of each segment
  int nrb, nfull;                    // row blocks of all segments; full (96-column) column blocks
  // x TEAMS of nfull workgroups walk the full column blocks: the first `rem` teams base + 1 row blocks each, the others base.
  // The narrow last column block's tiles (cheaper: two MFMA column tiles) are dealt BEHIND them so that everybody ends
  // together: n1 each to the workgroups of the long teams, n0 to those of the short ones, n2 to the workgroups beyond the
  // teams (which have nothing else); in workgroup order, the last ones clipped at nrb.  All 0 when there is no narrow block.
  int x, base, rem, n1, n0, n2;
};

enum { IT_FAST = 1, IT_CTX = 2 };

template <int TN, int XC>
__device__ __forceinline__ void stream_tiles(const StreamDev& args, float* __restrict__ As, float* __restrict__ Bs,
                                             const int n0, const int rbA, const int rbB) {
  constexpr int BM = 384, BNS = 32 * TN + XC, BK = 32, NT = 768, LS = BK + 4, C4 = BK / 4;
  constexpr int A_PER = BM * C4 / NT;  // 4
  constexpr int A_BUF = BM * LS, B_BUF = BNS * LS;
  constexpr int XCA = XC > 0 ? XC : 1;
  static_assert(A_PER == 4 && BNS * C4 <= NT && 2 * B_BUF + 1024 <= 2 * 96 * LS + 1024, "one B slot per thread, inside the kernel's Bs");
  const CarcaGemmDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nfast = D.K0 / BK;
  const int K1 = D.K1;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  auto seg_of = [&](int rb) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && rb >= args.rb_start[i]) s = i;
    return s;
  };

  // ---- per-thread invariants --------------------------------------------------------------------------------------
  const int r0 = tid >> 3, c4 = tid & 7;            // fast items: slot i is row r0 + 96 i, 16-byte group c4
  const int a_lds0 = r0 * LS + c4 * 4;
  const bool b_live = tid < BNS * C4;
  const int b_r = b_live ? r0 : 0;
  const unsigned b_v = (unsigned)(((size_t)min(n0 + b_r, D.N - 1) * D.ldb0 + c4 * 4) * sizeof(float));
  const int b_at0 = b_live ? b_r * LS + c4 * 4 : 2 * B_BUF + ((tid - BNS * C4) & 255) * 4;
  const int b_at1 = b_live ? B_BUF + b_r * LS + c4 * 4 : 2 * B_BUF + ((tid - BNS * C4) & 255) * 4;
  // context item: thread = (row tid >> 1, half tid & 1); the half's four columns start at cs = min(4 half, K1 - 4).  What
  // only this item needs is computed where it is used (once per tile), not kept in registers across the K steps.
  const int xr = tid >> 1, xh = tid & 1;
  auto cs_of = [&]() { return min(4 * xh, K1 - 4); };
  auto b1_v = [&]() {
    return (unsigned)(((size_t)min(n0 + (tid < 2 * BNS ? xr : 0), D.N - 1) * D.ldb1 + cs_of()) * sizeof(float));
  };
  auto bx_at = [&](int buf) { return tid < 2 * BNS ? buf * B_BUF + xr * LS + xh * 4 : 2 * B_BUF + ((tid - 2 * BNS) & 255) * 4; };
  const __amdgpu_buffer_rsrc_t b_rsrc = carca_rsrc(D.bt0);

  // ---- the load cursor: the next item to request -------------------------------------------------------------------
  const float* l_a0 = D.seg[0].a0;
  const float* l_a1 = D.seg[0].a1;
  unsigned a_v[A_PER], a1_v = 0;
  auto retarget = [&](int rb) {  // the cursor's tile: operand bases and this thread's row offsets (clamped into the segment)
    const int s = seg_of(rb);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = (rb - args.rb_start[s]) * BM, last = sg.rows - 1;
    l_a0 = sg.a0;
    l_a1 = sg.a1;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      a_v[i] = ((unsigned)min(row0 + r0 + 96 * i, last) * (unsigned)D.lda0 + c4 * 4) * 4u;  // (dense rows: the launcher refuses strided views)
    a1_v = ((unsigned)min(row0 + xr, last) * (unsigned)D.lda1 + cs_of()) * 4u;
  };
  f32x4 ra[A_PER], rbv;
  auto to_f = [](const u32x4 v) {
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };
  // slot i of the cursor's item (0..3: A, 4: B), of kind LK
  auto request_slot = [&](auto lk_tag, int i, const int so) {  // so: byte offset of the item's K tile inside a row
    constexpr int LK = decltype(lk_tag)::value;
    if constexpr (LK == IT_FAST) {
      if (i < A_PER)
        ra[i < A_PER ? i : 0] = to_f(__builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(l_a0), a_v[i < A_PER ? i : 0], so, 0));
      else
        rbv = to_f(__builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_v, so, 0));
    } else {
      if (i == 0) ra[0] = to_f(__builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(l_a1), a1_v, 0, 0));
      if (i == A_PER) rbv = to_f(__builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(D.bt1), b1_v(), 0, 0));
    }
  };
  auto ctx_fix = [&](const f32x4 v) {  // the clamped group moved to its columns; columns at or past K1 are zeros
    const int sh = 4 * xh - cs_of();
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = sh == 0 ? v[e] : (sh == 1 ? v[(e + 1) & 3] : (sh == 2 ? v[(e + 2) & 3] : v[(e + 3) & 3]));
      o[e] = (4 * xh + e < K1) ? x : 0.f;
    }
    return o;
  };
  auto store_slot = [&](auto rk_tag, int i, int buf) {
    constexpr int RK = decltype(rk_tag)::value;
    if constexpr (RK == IT_FAST) {
      if (i < A_PER)
        *reinterpret_cast<f32x4*>(&As[buf * A_BUF + a_lds0 + (i < A_PER ? i : 0) * 96 * LS]) = ra[i < A_PER ? i : 0];
      else
        *reinterpret_cast<f32x4*>(&Bs[buf ? b_at1 : b_at0]) = rbv;
    } else {
      if (i == 0) *reinterpret_cast<f32x4*>(&As[buf * A_BUF + xr * LS + xh * 4]) = ctx_fix(ra[0]);
      if (i == A_PER) *reinterpret_cast<f32x4*>(&Bs[bx_at(buf)]) = ctx_fix(rbv);
    }
  };

  // ---- accumulators, fragments -------------------------------------------------------------------------------------
  f32x16 acc[TN];
  f32x4 xacc[XCA];
  auto clear_acc = [&]() {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
#pragma unroll
    for (int c = 0; c < XCA; ++c) xacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * LS + 4 * lh];
  const float* b_frag = &Bs[lr * LS + 4 * lh];
  const float* x_frag = &Bs[(32 * TN) * LS + 4 * lh];
  f32x4 fa0, fa1, fb0[TN], fb1[TN], fx0[XCA], fx1[XCA];
#define CARCA_PIN() __builtin_amdgcn_sched_barrier(0)
  constexpr int NR = TN + 1, NRX = NR + XC, NSL = A_PER + 1;
  static_assert(NRX <= 4 * TN && NR + NSL <= 4 * TN, "a group's gaps hold its reads and the item's staging slots");
  auto read_frag = [&](int j, int buf, int kg, f32x4& fa, f32x4(&fb)[TN], f32x4(&fx)[XCA]) {
    if (j == 0)
      fa = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + kg * 8);
    else if (j < NR)
      fb[j - 1] = *reinterpret_cast<const f32x4*>(b_frag + buf * B_BUF + (j - 1) * 32 * LS + kg * 8);
    else
      fx[j - NR] = *reinterpret_cast<const f32x4*>(x_frag + buf * B_BUF + (j - NR) * LS + kg * 8);
  };
  auto mfma_group = [&](const f32x4& fa, const f32x4(&fb)[TN], const f32x4(&fx)[XCA], auto&& aux) {
#pragma unroll
    for (int i = 0; i < 4 * TN; ++i) {
      acc[i % TN] = mfma32(fa[i / TN], fb[i % TN][i / TN], acc[i % TN]);
      CARCA_PIN();
      if constexpr (XC > 0) {
        if (i < XC) {  // (two v_pk_fma_f32, written out: gemm.hip, cu_tile)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const int c = i < XC ? i : 0;
          f32x2 a0 = {fa[0], fa[1]}, a1 = {fa[2], fa[3]};
          f32x2 w0 = {fx[c][0], fx[c][1]}, w1 = {fx[c][2], fx[c][3]};
          f32x2 x0 = {xacc[c][0], xacc[c][1]}, x1 = {xacc[c][2], xacc[c][3]};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(a0), "v"(w0));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(a1), "v"(w1));
          xacc[c] = f32x4{x0[0], x0[1], x1[0], x1[1]};
        }
      }
      aux(i);
      CARCA_PIN();
    }
  };
  // one fast item in LDS buffer CUR: four 8-k groups; the item in registers (kind RK) goes to buffer NXT, the cursor's item
  // (kind LK) is requested, the next item's first fragments are read behind the barrier.  The kinds are COMPILE-TIME (a
  // branch per staging slot between the pinned MFMAs cut the groups into pieces and cost ~200 spilled registers).
  auto fast_step = [&](auto cur_tag, auto rk_tag, auto lk_tag, const int so) {
    constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
    mfma_group(fa0, fb0, fx0, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 1, fa1, fb1, fx1);
    });
    mfma_group(fa1, fb1, fx1, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 2, fa0, fb0, fx0);
      if (i >= NR && i < NR + NSL) store_slot(rk_tag, i - NR, NXT);
    });
    mfma_group(fa0, fb0, fx0, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 3, fa1, fb1, fx1);
      if (i >= NR && i < NR + NSL) request_slot(lk_tag, i - NR, so);
    });
    CARCA_PIN();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // raw: the requested item stays in flight across it
    CARCA_PIN();
    mfma_group(fa1, fb1, fx1, [&](int i) {
      if (i < NRX) read_frag(i, NXT, 0, fa0, fb0, fx0);
    });
  };
  // the context item in LDS buffer CUR, a tile's last: one 8-k group (its fragments were read behind the previous barrier);
  // the next tile's item 0 goes to buffer NXT, its item 1 is requested
  auto ctx_step = [&](auto cur_tag) {
    constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
    (void)CUR;
    constexpr std::integral_constant<int, IT_FAST> kf{};
    mfma_group(fa0, fb0, fx0, [&](int i) {
      if (i < NSL) store_slot(kf, i, NXT);
      if (i >= NSL && i < 2 * NSL) request_slot(kf, i - NSL, BK * 4);
    });
#pragma unroll
    for (int i = 4 * TN; i < 2 * NSL; ++i) request_slot(kf, i - NSL, BK * 4);  // (TN = 2: eight gaps for ten slots)
    CARCA_PIN();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CARCA_PIN();
  };
#undef CARCA_PIN

  // ---- the epilogue of row block rb (the plain one: alpha, bias, rows with id 0 as zeros) -----------------------------
  auto epilogue = [&](int rb) {
    const unsigned c_v = (unsigned)((((size_t)wave * 32 + 4 * lh) * D.ldc + n0 + lr) * sizeof(float));
    const int s = seg_of(rb);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = (rb - args.rb_start[s]) * BM, nrows = sg.rows;
    unsigned keep = 0xffffffffu;  // bit j: row row0 + 32 wave + j is kept
    if (D.mask_rows) {
      const int idv = gload1i(sg.ids, min(row0 + wave * 32 + lr, nrows - 1));
      keep = (unsigned)__ballot(idv != 0);
    }
    const unsigned keep_l = keep >> (4 * lh);
    const __amdgpu_buffer_rsrc_t c_rsrc = carca_rsrc(sg.c);
    const bool full = row0 + BM <= nrows;  // (uniform: every row of the tile exists)
    const float alpha = D.alpha != 0.f ? D.alpha : 1.0f;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + tn * 32 + lr;
      const bool n_ok = n < D.N;
      const float bias = (D.bias && n_ok) ? D.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        const int row = row0 + wave * 32 + rr + 4 * lh;
        float v = alpha * acc[tn][r] + bias;
        v = ((keep_l >> rr) & 1u) ? v : 0.f;
        const int so = ((row0 + rr) * D.ldc + tn * 32) * (int)sizeof(float);
        if (n_ok && (full || row < nrows)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, c_v, so, 0);
      }
    }
    if constexpr (XC > 0) {
      const int row = row0 + wave * 32 + lr;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const float mine = (xacc[c][0] + xacc[c][1]) + (xacc[c][2] + xacc[c][3]);
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
        const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        const int n = n0 + 32 * TN + c;
        if (lh == 0 && row < nrows && n < D.N) {
          float v = alpha * tot + (D.bias ? D.bias[n] : 0.f);
          v = ((keep >> lr) & 1u) ? v : 0.f;
          sg.c[(size_t)row * D.ldc + n] = v;
        }
      }
    }
  };

  // ---- the stream: per tile nfast fast items and the context item (the launcher: nfast even, 4 <= K1 <= 8).  A tile is
  // nfast + 1 items, an odd number: consecutive tiles start in alternate LDS buffers, and the tile's body exists once per
  // starting buffer P -- straight-line pairs of steps, the accumulators born and stored inside it (a loop over single steps
  // that picks the buffer at run time made the compiler keep a second copy of the 48 accumulator registers).
  if (rbA >= rbB) return;
  using std::integral_constant;
  constexpr integral_constant<int, IT_FAST> kF{};
  constexpr integral_constant<int, IT_CTX> kC{};
  retarget(rbA);
#pragma unroll
  for (int i = 0; i < NSL; ++i) request_slot(kF, i, 0);
#pragma unroll
  for (int i = 0; i < NSL; ++i) store_slot(kF, i, 0);
#pragma unroll
  for (int i = 0; i < NSL; ++i) request_slot(kF, i, BK * 4);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NRX; ++j) read_frag(j, 0, 0, fa0, fb0, fx0);
  // tile rb: its item 0 in LDS buffer P (first fragments read), item 1 in registers
  auto tile_body = [&](auto p_tag, const int rb) {
    constexpr int P = decltype(p_tag)::value;
    constexpr integral_constant<int, P> c0{};
    constexpr integral_constant<int, P ^ 1> c1{};
    clear_acc();
    int t = 0;
    for (; t + 2 < nfast; t += 2) {  // items t, t + 1 multiply; items t + 2, t + 3 are requested
      fast_step(c0, kF, kF, (t + 2) * (BK * 4));
      fast_step(c1, kF, kF, (t + 3) * (BK * 4));
    }
    // t = nfast - 2: the context item is requested; then the NEXT tile's first items (behind the last tile: the last tile's
    // again, never used -- every tile's steps then see the same kinds of items around them, all compile-time)
    fast_step(c0, kF, kC, 0);
    retarget(min(rb + 1, rbB - 1));
    fast_step(c1, kC, kF, 0);
    ctx_step(c0);
    epilogue(rb);
    if (rb + 1 < rbB) {  // the next tile starts in buffer P ^ 1
#pragma unroll
      for (int j = 0; j < NRX; ++j) read_frag(j, P ^ 1, 0, fa0, fb0, fx0);
    }
  };
  for (int rb = rbA; rb < rbB; rb += 2) {
    tile_body(integral_constant<int, 0>{}, rb);
    if (rb + 1 < rbB) tile_body(integral_constant<int, 1>{}, rb + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the stream's last requests are never stored: let them land)
}

template <int XC>
__global__ __launch_bounds__(768) void gemm_rows_cus_kernel(const StreamDev args) {
  __shared__ __attribute__((aligned(16))) float As[2 * 384 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[2 * 96 * 36 + 1024];
  carca_warm_kernargs<sizeof(StreamDev)>();
  // consecutive workgroup numbers w on ONE XCD: the nfull workgroups of a team stream the same A rows through one L2
  const int id = blockIdx.x, nw = gridDim.x;
  const int xcd = id & 7, q8 = nw >> 3, r8 = nw & 7;
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int nfull = args.nfull, x = args.x, rem = args.rem;
  if (w < x * nfull) {
    const int j = w / nfull, cb = w - j * nfull;
    const int rbA = j * args.base + min(j, rem), rbB = rbA + args.base + (j < rem ? 1 : 0);
    stream_tiles<3, 0>(args, As, Bs, cb * 96, rbA, rbB);
  }
  // its share of the narrow column block's tiles
  const int c1 = rem * nfull, c0 = (x - rem) * nfull;
  int nA, nB;
  if (w < c1) {
    nA = w * args.n1;
    nB = nA + args.n1;
  } else if (w < c1 + c0) {
    nA = c1 * args.n1 + (w - c1) * args.n0;
    nB = nA + args.n0;
  } else {
    nA = c1 * args.n1 + c0 * args.n0 + (w - c1 - c0) * args.n2;
    nB = nA + args.n2;
  }
  nA = min(nA, args.nrb);
  nB = min(nB, args.nrb);
  if (nA < nB) {
    __syncthreads();  // (everybody is done with the LDS buffers of the full tiles)
    stream_tiles<2, XC>(args, As, Bs, nfull * 96, nA, nB);
  }
}

}  // namespace

// CARCA_OK = launched; 1 = not this kernel's product (the caller goes on to gemm_rows_cu_kernel)
int carca_gemm_rows_stream_try(const CarcaGemmDesc* desc, bool fits32, hipStream_t stream) {
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  if (variant == 24 || !fits32) return 1;  // (24: never -- A/B switch)
  if (desc->K0 % 64 != 0 || desc->K0 < 128) return 1;  // (an even number of K steps, at least four)
  if (desc->K1 < 4 || desc->K1 > 8 || !desc->bt1) return 1;  // (the context item: one 8-k group, two 16-byte groups per row)
  if (desc->ncols_out != desc->N || desc->N <= 96) return 1;
  if (desc->colvec || desc->pos || desc->add_table) return 1;
  const int ncb = (desc->N + 95) / 96;
  const int rem = desc->N - 96 * (ncb - 1);  // columns of the last block: 96 = a full one
  const int xc = rem == 96 ? 0 : (rem > 64 ? rem - 64 : 0);
  if (rem != 96 && (rem <= 32 || xc > 2)) return 1;
  const int nfull = rem == 96 ? ncb : ncb - 1;
  StreamDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    if (sg.add || sg.gate || sg.rowscale || sg.add_pos || sg.a0_gather || (desc->mask_rows && !sg.ids)) return 1;
    if (sg.a0_bstride || sg.a1_bstride) return 1;  // (dense rows only: a row's offset is row x lda)
    if ((uint64_t)sg.rows * (uint64_t)desc->ldc * 4ull >= (1ull << 31)) return 1;  // (the epilogue's store offsets are signed 32-bit scalars)
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (sg.rows + 383) / 384;
  }
  for (int s = desc->nseg; s <= CARCA_MAX_SEGS; ++s) g.rb_start[s] = rb;
  g.nrb = rb;
  g.nfull = nfull;
  const int ncu = carca_num_cus();
  const int nfast = desc->K0 / 32;
  // worth it: several tiles per workgroup (a single round belongs to the one-tile kernels and their stream-K relatives)
  // and a K short enough that the per-tile costs matter (tuning variant 25 forces the kernel wherever it is correct)
  // (B = 256 at C5's other dimensions, 505 tiles: 152 against 168 us on the one-tile kernel; the CLI's default shape, 303 tiles:
  // 139 against 113 -- the bound sits between them)
  if (variant != 25 && ((long)rb * ncb * 4 < 7l * ncu || nfast >= 64)) return 1;
  // Everybody in teams on the full column blocks, the narrow block's tiles behind them (StreamDev): the smallest common end T
  // (in 1/100 of a full tile's time) for which the narrow tiles all find a place.  C5: 64 teams of four, 31 of them six row
  // blocks and 33 five; T = 6.74 tiles -- one narrow tile behind six full ones, two behind five -- where separate workgroups
  // for the narrow block made it 7.
  const long cheap = rem == 96 ? 0 : (xc == 2 ? 74 : (xc == 1 ? 71 : 68));
  const int grid = (int)std::min<long>(ncu, (long)rb * nfull + (cheap ? rb : 0));
  int x = std::min(rb, grid / nfull);
  if (x < 1) return 1;
  g.x = x;
  g.base = rb / x;
  g.rem = rb - g.base * x;
  g.n1 = g.n0 = g.n2 = 0;
  if (cheap) {
    const long c1 = (long)g.rem * nfull, c0 = (long)(x - g.rem) * nfull, ce = grid - (long)x * nfull;
    for (long T = 100l * g.base;; ++T) {
      const long n1 = std::max(0l, (T - 100l * (g.base + 1)) / cheap), n0 = std::max(0l, (T - 100l * g.base) / cheap), n2 = T / cheap;
      if (c1 * n1 + c0 * n0 + ce * n2 >= rb) {
        g.n1 = (int)n1;
        g.n0 = (int)n0;
        g.n2 = (int)n2;
        break;
      }
    }
  }
  carca_rows_log(xc == 0 ? "gemm_rows_cus_kernel<0>" : (xc == 1 ? "gemm_rows_cus_kernel<1>" : "gemm_rows_cus_kernel<2>"), desc, grid);
  hipEvent_t e0, e1;
  const bool ev = carca_take_launch_events(&e0, &e1);
  if (xc == 0) {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_cus_kernel<0>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_cus_kernel<0>), dim3(grid), dim3(768), 0, stream, g);
  } else if (xc == 1) {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_cus_kernel<1>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_cus_kernel<1>), dim3(grid), dim3(768), 0, stream, g);
  } else {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_cus_kernel<2>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_cus_kernel<2>), dim3(grid), dim3(768), 0, stream, g);
  }
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// gemm_rows_n96s_kernel: the NARROW-output row product (64 < N <= 96: AllEmbedding.joint_embed, carca.py:89, e = [z ; q]
// W_j^T + b_j -- in inference over q's g columns with the item term from the projected table, CarcaGemmDesc.add_table) for
// MANY rows per CU: BASELINE's C5 has 134,528 rows, C3 77,312.  gemm_rows_n96_kernel is one 80-row block per CU and is only
// chosen for a single round (C2: 242 blocks); beyond it the tiled kernel ran (C5: 177 us, 44 % of the fp32 MFMA peak), every
// 128-row block paying its own start, fill and epilogue.  Here ONE persistent workgroup per CU takes an EQUAL share of the
// rows (cut at 16-row granularity, not at block granularity: 841 blocks of 160 rows over 256 CUs would be 3 or 4 each) and
// walks it in blocks of 160 rows x all 96 columns:
//   * wave (ct = wave % 6, rh = wave / 6) owns column tile ct for the row tiles 2 j + rh, j = 0..4, over the WHOLE K -- no
//     K halves to add up in LDS at the end (gemm_rows_n96_kernel's 80-row blocks need that exchange to occupy twelve waves),
//     80 MFMAs (16x16x4) per 64-wide K stage and wave in five independent chains, one barrier per stage;
//   * the K stages of consecutive blocks are ONE stream: a stage is requested two stages ahead of its MFMAs into registers
//     and stored to the LDS double buffer one stage ahead, across block boundaries;
//   * a block's row ids are requested when it starts, its table rows / positional rows two stages before its end: the
//     epilogue is arithmetic and buffer stores (row in the scalar offset);
//   * a partial block (the share's last, a segment's last) skips the row tiles it does not have, wave by wave.
// Same K order per output element as the other row kernels (stages of 64 in order); agrees with them to round-off.
namespace n96s {
constexpr int BM = 160, BN = 96, BK = 64, LS = BK + 4, NT = 768, C4 = BK / 4;
constexpr int A_BUF = BM * LS, B_BUF = BN * LS;  // floats per stage
constexpr int NA = 4, NB = 2;                    // staging slots per thread and stage: A rows r0 + 48 i (i = 3: threads < 256), B rows r0 + 48 j
constexpr size_t LDS_BYTES = sizeof(float) * 2 * (A_BUF + B_BUF) + sizeof(int) * 2 * BM;  // + the row ids of two blocks
static_assert(BM * C4 == 3 * NT + 256 && BN * C4 == NB * NT, "slot map");
}  // namespace n96s

struct N96sDev {
  CarcaGemmDesc d;
  int row_start[CARCA_MAX_SEGS + 1];  // global row (all segments laid end to end) of each segment's first row
  unsigned long long* dbg;            // tools/stamp_n96s.py: wall-clock stamps (100 MHz) of a diagnostic run, or null
};

// DIAG (tuning key 5, timing experiments with WRONG results): 1 no global loads, 2 no MFMAs, 4 no LDS stores, 8 no addends /
// output stores, 16 no fragment reads
template <int DIAG>
__global__ __launch_bounds__(768) void gemm_rows_n96s_kernel(const N96sDev args) {
  using namespace n96s;
  extern __shared__ __attribute__((aligned(16))) float Sm[];  // [2][A stage | B stage]
  carca_warm_kernargs<sizeof(N96sDev)>();
  const CarcaGemmDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, mq = lane >> 4;
  const int ct = wave % 6, rh = wave / 6;
  const int K0 = D.K0, nst = (K0 + BK - 1) / BK;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define N96S_STAMP(k) do { if (args.dbg && tid == 0 && (k) < 64) args.dbg[65536 + blockIdx.x * 64 + (k)] = wall_clock64(); } while (0)
  N96S_STAMP(0);
  // this workgroup's share of the rows, in 16-row units
  const int R = args.row_start[CARCA_MAX_SEGS];
  const int units = (R + 15) / 16;
  const int g_lo = (int)((long)units * blockIdx.x / gridDim.x) * 16, g_hi = min(R, (int)((long)units * (blockIdx.x + 1) / gridDim.x) * 16);
  if (g_lo >= g_hi) return;

  // ---- staging ----------------------------------------------------------------------------------------------------
  const int r0 = tid >> 4, c = tid & 15;  // slot i: A row r0 + 48 i (i < 4), B row r0 + 48 j; 16-byte group c of the stage
  const int lds0 = r0 * LS + c * 4;
  const int i3 = tid < 256 ? 3 : 2;  // (A has 3 1/3 slots per thread: the others repeat their third one)
  unsigned offb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) offb[j] = (unsigned)min(r0 + 48 * j, D.N - 1) * (unsigned)D.ldb0 * 4u;
  const __amdgpu_buffer_rsrc_t r_b = carca_rsrc(D.bt0);
  // the cursor's block (the block whose stages are being requested)
  int l_g = g_lo;            // its first global row
  const float* l_a0 = D.seg[0].a0;
  unsigned offa[NA];
  auto block_end = [&](int g) {  // one past the block that starts at global row g: 160 rows, the segment's end or the share's
    int e = min(g + BM, g_hi);
#pragma unroll
    for (int s = 1; s <= CARCA_MAX_SEGS; ++s)
      if (g < args.row_start[s]) e = min(e, args.row_start[s]);
    return e;
  };
  auto seg_of = [&](int g) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && g >= args.row_start[i]) s = i;
    return s;
  };
  auto retarget = [&](int g) {
    const int s = seg_of(g);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = g - args.row_start[s], last = sg.rows - 1;
    l_a0 = sg.a0;
#pragma unroll
    for (int i = 0; i < NA; ++i) offa[i] = (unsigned)min(row0 + r0 + 48 * (i < 3 ? i : i3), last) * (unsigned)D.lda0 * 4u;
  };
  f32x4 rg[NA + NB];
  auto to_f = [](const u32x4 v) {
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  };
  auto load_stage = [&](int st) {  // (a group past K0 is clamped to end at K0 and moved / zeroed when stored: gemm_rows_n96_kernel)
    if constexpr (DIAG & 1) return;
    const unsigned kc = 4u * (unsigned)min(st * BK + c * 4, K0 - 4);
    const __amdgpu_buffer_rsrc_t r_a = carca_rsrc(l_a0);
#pragma unroll
    for (int i = 0; i < NA; ++i) rg[i] = to_f(__builtin_amdgcn_raw_buffer_load_b128(r_a, offa[i] + kc, 0, 0));
#pragma unroll
    for (int j = 0; j < NB; ++j) rg[NA + j] = to_f(__builtin_amdgcn_raw_buffer_load_b128(r_b, offb[j] + kc, 0, 0));
  };
  auto store_stage = [&](int st, int buf) {
    if constexpr (DIAG & 4) return;
    float* As = Sm + buf * (A_BUF + B_BUF);
    float* Bs = As + A_BUF;
    const int kb = st * BK;
    if (kb + BK <= K0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[lds0 + 48 * LS * (i < 3 ? i : i3)]) = rg[i];
#pragma unroll
      for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(&Bs[lds0 + 48 * LS * j]) = rg[NA + j];
      return;
    }
    const int kcol = kb + c * 4, sh = kcol - min(kcol, K0 - 4);
    auto fix = [&](const f32x4 r) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = sh == 0 ? r[e] : (sh == 1 ? r[(e + 1) & 3] : (sh == 2 ? r[(e + 2) & 3] : r[(e + 3) & 3]));
        v[e] = (kcol + e < K0) ? x : 0.f;
      }
      return v;
    };
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(&As[lds0 + 48 * LS * (i < 3 ? i : i3)]) = fix(rg[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(&Bs[lds0 + 48 * LS * j]) = fix(rg[NA + j]);
  };

  // ---- MFMAs: acc[j][r] = C[block row 16 (2 j + rh) + 4 mq + r][16 ct + ln] --------------------------------------------
  f32x4 acc[5];
  const int a_frag = (16 * rh + ln) * LS + 4 * mq, b_frag = A_BUF + (16 * ct + ln) * LS + 4 * mq;
  auto compute = [&](int buf, int kleft, int nj) {  // kleft: columns of K0 from this stage's first one on; nj: the wave's row tiles
    if constexpr (DIAG & 2) return;
    const float* S = Sm + buf * (A_BUF + B_BUF);
    if (nj == 5) {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (16 * ch >= kleft) continue;  // (the ragged last stage: a chunk past K0 is zeros in both operands)
        const f32x4 b = *reinterpret_cast<const f32x4*>(S + b_frag + 16 * ch);
        f32x4 a[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] = *reinterpret_cast<const f32x4*>(S + a_frag + 32 * j * LS + 16 * ch);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[j] = mfma16(a[j][e], b[e], acc[j]);
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (16 * ch >= kleft) continue;
        const f32x4 b = *reinterpret_cast<const f32x4*>(S + b_frag + 16 * ch);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          if (j < nj) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(S + a_frag + 32 * j * LS + 16 * ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j] = mfma16(a[e], b[e], acc[j]);
          }
        }
      }
    }
  };

  // The hot step -- a full 64-wide stage of a full block -- hand-scheduled as gemm_rows_cu_kernel's: four groups of twenty
  // MFMAs (one 16-k chunk each: five accumulator chains), every gap between two MFMAs pinned; the gaps of a group carry
  // the six fragment reads of the NEXT chunk (two fragment sets alternate), the gaps of group 1 the six LDS stores of stage
  // st + 1, those of group 2 the six requests of stage st + 2; ONE barrier behind group 2, the next stage's first
  // fragments read under group 3.  Left to the compiler (store, request, multiply, barrier: the generic step below, kept
  // for the ragged last stage and for partial blocks) a stage took 5.5 us against 3.3 us of MFMA time.
  f32x4 fa0[5], fa1[5], fb0, fb1;
  const int lds3 = lds0 + 48 * LS * i3;
#define N96S_PIN() __builtin_amdgcn_sched_barrier(0)
  auto read_slot = [&](int i, const float* S, int ch, f32x4(&fa)[5], f32x4& fb) {
    if constexpr (DIAG & 16) return;
    if (i == 0)
      fb = *reinterpret_cast<const f32x4*>(S + b_frag + 16 * ch);
    else
      fa[i > 0 ? i - 1 : 0] = *reinterpret_cast<const f32x4*>(S + a_frag + 32 * (i > 0 ? i - 1 : 0) * LS + 16 * ch);
  };
  auto mfma_group = [&](const f32x4(&fa)[5], const f32x4& fb, auto&& aux) {
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      if constexpr (!(DIAG & 2)) acc[i % 5] = mfma16(fa[i % 5][i / 5], fb[i / 5], acc[i % 5]);
      N96S_PIN();
      aux(i);
      N96S_PIN();
    }
  };
  auto store_slot = [&](auto rag_tag, int i, float* So, const int kb1) {  // rag: the stored stage (first column kb1) ends past K0
    if constexpr (DIAG & 4) return;
    f32x4 v = rg[i];
    if constexpr (decltype(rag_tag)::value) {
      const int kcol = kb1 + c * 4, sh = kcol - min(kcol, K0 - 4);
      const f32x4 r = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = sh == 0 ? r[e] : (sh == 1 ? r[(e + 1) & 3] : (sh == 2 ? r[(e + 2) & 3] : r[(e + 3) & 3]));
        v[e] = (kcol + e < K0) ? x : 0.f;
      }
    }
    if (i < NA)
      *reinterpret_cast<f32x4*>(So + (i < 3 ? lds0 + 48 * LS * i : lds3)) = v;
    else
      *reinterpret_cast<f32x4*>(So + A_BUF + lds0 + 48 * LS * (i - NA)) = v;
  };
  auto load_slot = [&](int i, const unsigned kcv) {
    if constexpr (DIAG & 1) return;
    if (i < NA)
      rg[i] = to_f(__builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(l_a0), offa[i < NA ? i : 0] + kcv, 0, 0));
    else
      rg[i] = to_f(__builtin_amdgcn_raw_buffer_load_b128(r_b, offb[i >= NA ? i - NA : 0] + kcv, 0, 0));
  };
  auto pre_read = [&](int buf) {
    const float* S = Sm + buf * (A_BUF + B_BUF);
#pragma unroll
    for (int i = 0; i < 6; ++i) read_slot(i, S, 0, fa0, fb0);
  };
  // (taken only while two more stages follow: a branch per staging slot between the pinned MFMAs cuts the groups to pieces)
  auto step_full = [&](auto rag_tag, int buf, const unsigned kcv, const int kb1) __attribute__((always_inline)) {
    const float* S = Sm + buf * (A_BUF + B_BUF);
    float* So = Sm + (buf ^ 1) * (A_BUF + B_BUF);
    mfma_group(fa0, fb0, [&](int i) {
      if (i < 6) read_slot(i, S, 1, fa1, fb1);
    });
    mfma_group(fa1, fb1, [&](int i) {
      if (i < 6) read_slot(i, S, 2, fa0, fb0);
      if (i >= 6 && i < 6 + NA + NB) store_slot(rag_tag, i - 6, So, kb1);
    });
    mfma_group(fa0, fb0, [&](int i) {
      if (i < 6) read_slot(i, S, 3, fa1, fb1);
      if (i >= 6 && i < 6 + NA + NB) load_slot(i - 6, kcv);
    });
    N96S_PIN();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // raw: the requested stage stays in flight across it
    N96S_PIN();
    mfma_group(fa1, fb1, [&](int i) {
      if (i < 6) read_slot(i, So, 0, fa0, fb0);
    });
  };
#undef N96S_PIN

  // ---- epilogue operands -------------------------------------------------------------------------------------------
  // (Measured and dropped, round 5: the raw sums through the free stage buffer as a [160][100] tile and 16-byte row-wise
  // table reads / stores -- 153 us against 140 at C5; what the epilogue costs, ~6 us per block by the kernel's stamps, is the
  // WAIT for the table rows behind the next block's freshly requested stages, not the width of its accesses:
  // tools/stamp_n96s.py, DIAG 32 / 64.)
  const int n = 16 * ct + ln;
  const bool n_ok = n < D.N;
  const int nn = min(n, D.N - 1);
  const float bias = D.bias ? gload1(D.bias, nn) : 0.f;
  const bool use_tab = D.add_table != nullptr;
  float addv[5][4];  // the table's rows + the positional rows of the lane's twenty outputs
  // A block's row ids wait in LDS (requested one block ahead by threads 0..159, one each: twenty per lane kept in registers
  // across the K stages were what the hand-scheduled step spilled): Ids[parity of the block][row of the block]
  int* const Ids = reinterpret_cast<int*>(Sm + 2 * (A_BUF + B_BUF));
  int id_mine = 0;
  auto request_ids = [&](int g) {  // thread t < 160: the id of row t of the block at global row g (clamped into the segment)
    const int s = seg_of(g);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = g - args.row_start[s];
    const int32_t* idp = sg.ids ? sg.ids : reinterpret_cast<const int32_t*>(D.bt0);
    id_mine = gload1i(idp, sg.ids ? min(row0 + min(tid, BM - 1), sg.rows - 1) : 0);
  };
  auto park_ids = [&](int par) {
    if (tid < BM) Ids[par * BM + tid] = id_mine;
  };
  auto request_addends = [&](int g, int par) {
    if constexpr (DIAG & 8) return;
    const int s = seg_of(g);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = g - args.row_start[s];
    const bool use_pos = sg.add_pos && D.pos;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) addv[j][r] = 0.f;
    if (use_tab) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          addv[j][r] = gload1(D.add_table, Ids[par * BM + 16 * (2 * j + rh) + 4 * mq + r] * D.ld_add_table + nn);
    }
    if (use_pos) {  // (positional rows: row % T, twenty divisions -- only where an encoding is configured)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(row0 + 16 * (2 * j + rh) + 4 * mq + r, sg.rows - 1);
          addv[j][r] += gload1(D.pos, (row % sg.T) * D.N + nn);
        }
    }
  };
  auto epilogue = [&](int g, int g_end, int par) __attribute__((always_inline)) {
    const int s = seg_of(g);
    const CarcaGemmSeg sg = D.seg[s];
    const int row0 = g - args.row_start[s], nrows = g_end - g;
    const __amdgpu_buffer_rsrc_t c_rsrc = carca_rsrc(sg.c);
    const unsigned c_v = (unsigned)((4 * mq * D.ldc + n) * 4);
    const float alpha = D.alpha != 0.f ? D.alpha : 1.0f;
    if (n >= D.ncols_out) return;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int br = 16 * (2 * j + rh) + 4 * mq + r;  // row inside the block
        float v = alpha * acc[j][r] + bias + addv[j][r];
        if (D.mask_rows) v = Ids[par * BM + br] != 0 ? v : 0.f;
        v = n_ok ? v : 0.f;
        const int so = (row0 + 16 * (2 * j + rh) + r) * D.ldc * 4;
        if (br < nrows && (!(DIAG & 8) || v == 123.456f)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rsrc, c_v, so, 0);
      }
  };
  auto tiles_of = [&](int nrows) {  // the wave's row tiles 2 j + rh that start inside a block of nrows rows
    const int nt = (nrows + 15) >> 4;
    return max(0, min(5, (nt - rh + 1) >> 1));
  };

  // ---- the stream ------------------------------------------------------------------------------------------------------
  retarget(g_lo);
  load_stage(0);
  request_ids(g_lo);
  store_stage(0, 0);
  park_ids(0);
  load_stage(1);
  __syncthreads();
  pre_read(0);
  N96S_STAMP(1);
  int buf = 0, par = 0, nblk_done = 0;
  // stages st of a block whose step can be the pinned one: stage st full, stage st + 1 full (it is stored without shifting),
  // two more stages behind it INSIDE the block (the block's last two steps request the next block's stages: generic)
  const int npin_full = max(0, min(K0 / BK - 1, nst - 2));  // (= nst - 2 whenever at most the last stage is ragged)
  auto generic_step = [&](int st, int g, int g_end, bool more, int nj) __attribute__((always_inline)) {
    const bool has1 = st + 1 < nst || more, has2 = st + 2 < nst || more;
    const int st1 = st + 1 < nst ? st + 1 : 0, st2 = st + 2 < nst ? st + 2 : st + 2 - nst;
    if (st + 2 == nst && more) retarget(g_end);
    if (st + 2 == nst) request_addends(g, par);
    if (has1) store_stage(st1, buf ^ 1);
    if (has2) load_stage(st2);
    compute(buf, K0 - st * BK, nj);
    __syncthreads();
    if (has1) pre_read(buf ^ 1);
    buf ^= 1;
  };
  constexpr std::integral_constant<bool, false> kPlain{};
  for (int g = g_lo; g < g_hi;) {
    const int g_end = block_end(g), nj = tiles_of(g_end - g);
    const bool more = g_end < g_hi;
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // stage st multiplies out of LDS buffer `buf`; stage st + 1 (in registers) goes to the other buffer; stage st + 2 is
    // requested -- behind a block's last stages the NEXT block's first ones (behind the share's last block: nothing).
    // The NEXT block's ids: requested here, parked in LDS behind the pinned steps (a few stages later).
    if (more) request_ids(g_end);
    const int sb = 2 + 14 * nblk_done;  // stamps of this block: start, behind the pinned steps, behind each later step, around the epilogue
    N96S_STAMP(sb);
    int st = 0;
    if (nj == 5) {
      // (a loop of nothing but pinned steps: with the generic step as an alternative inside the same loop the compiler kept
      // the accumulators in scratch between steps; pinned steps for the block's last two stages as well -- three more places
      // where the two kinds of step meet -- cost 150-280 spilled registers: they stay generic)
      for (; st < npin_full; ++st) {
        step_full(kPlain, buf, 4u * (unsigned)min((st + 2) * BK + c * 4, K0 - 4), 0);
        buf ^= 1;
      }
    }
    N96S_STAMP(sb + 1);
    if (st == 0) generic_step(st++, g, g_end, more, nj);
    if (more) park_ids(par ^ 1);
    for (; st < nst; ++st) {
      generic_step(st, g, g_end, more, nj);
      N96S_STAMP(sb + 2 + min(max(st - (nst - 2), 0), 9));
    }
    N96S_STAMP(sb + 12);
    epilogue(g, g_end, par);
    N96S_STAMP(sb + 13);
    ++nblk_done;
    par ^= 1;
    g = g_end;
  }
}

}  // namespace

int carca_gemm_rows_n96s_try(const CarcaGemmDesc* desc, bool fits32, hipStream_t stream) {
  using namespace n96s;
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  if (variant == 26 || !fits32) return 1;  // (26: never -- A/B switch)
  if (desc->K1 != 0 || desc->N <= 64 || desc->N > 96 || desc->ncols_out > 96 || desc->K0 < 4 * BK) return 1;
  if (desc->colvec) return 1;
  N96sDev g{};
  g.d = *desc;
  long rows = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    if (sg.add || sg.gate || sg.rowscale || sg.a0_gather || sg.a0_bstride) return 1;
    if ((desc->mask_rows || desc->add_table) && !sg.ids) return 1;
    if (sg.add_pos && (!desc->pos || sg.T < 1)) return 1;
    if ((uint64_t)sg.rows * (uint64_t)desc->ldc * 4ull >= (1ull << 31)) return 1;  // (signed 32-bit store offsets)
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.row_start[s] = (int)rows;
    rows += sg.rows;
  }
  for (int s = desc->nseg; s <= CARCA_MAX_SEGS; ++s) g.row_start[s] = (int)rows;
  if (rows >= (1l << 30)) return 1;
  const int ncu = carca_num_cus();
  // worth it: from 2.5 blocks per CU on (tuning variant 27 forces the kernel wherever it is correct).  Measured, interleaved A/B
  // of the joint product: C5 (134,528 rows, 3.3 blocks per CU) 142 against 174 us on the tiled kernel; C3 (77,312 rows, 1.9
  // blocks per CU) 97 against 88 us -- a share's last, partial block costs its seven latency-bound steps whatever it holds
  if (variant != 27 && rows < (long)ncu * BM * 5 / 2) return 1;
  const int grid = (int)std::min<long>(ncu, (rows + BM - 1) / BM);
  const int diag = carca_tuning(5);
  static bool attr_set[128] = {false};
  auto launch = [&](auto kern) -> int {
    if (!attr_set[diag & 127]) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return 1;
      }
      attr_set[diag & 127] = true;
    }
    hipEvent_t e0, e1;
    if (carca_take_launch_events(&e0, &e1))
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(768), LDS_BYTES, stream, e0, e1, 0, g);
    else
      hipLaunchKernelGGL(kern, dim3(grid), dim3(768), LDS_BYTES, stream, g);
    return 0;
  };
  g.dbg = carca_debug_buffer();
  carca_rows_log("gemm_rows_n96s_kernel", desc, grid);
  int lrc;
  switch (diag) {
    case 1: lrc = launch(gemm_rows_n96s_kernel<1>); break;
    case 2: lrc = launch(gemm_rows_n96s_kernel<2>); break;
    case 4: lrc = launch(gemm_rows_n96s_kernel<4>); break;
    case 8: lrc = launch(gemm_rows_n96s_kernel<8>); break;
    case 16: lrc = launch(gemm_rows_n96s_kernel<16>); break;
    case 5: lrc = launch(gemm_rows_n96s_kernel<5>); break;
    case 21: lrc = launch(gemm_rows_n96s_kernel<21>); break;
    default: lrc = launch(gemm_rows_n96s_kernel<0>); break;
  }
  if (lrc) return 1;
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}
