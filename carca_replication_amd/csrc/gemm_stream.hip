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
// 66) is two MFMA column tiles + up to two VALU columns as in gemm_rows_sk_kernel, run by workgroups of their own, their
// number chosen so that both kinds finish together.  Same products in the same order as gemm_rows_cu_kernel per output
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
  int x, y;                          // workgroups per full column block; workgroups of the narrow block (0: there is none)
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
  const int nfull = args.nfull, x = args.x;
  if (w < x * nfull) {
    const int j = w / nfull, cb = w - j * nfull;
    const int rbA = (int)((long)args.nrb * j / x), rbB = (int)((long)args.nrb * (j + 1) / x);
    stream_tiles<3, 0>(args, As, Bs, cb * 96, rbA, rbB);
  } else {
    const int j = w - x * nfull, y = args.y;
    const int rbA = (int)((long)args.nrb * j / y), rbB = (int)((long)args.nrb * (j + 1) / y);
    stream_tiles<2, XC>(args, As, Bs, nfull * 96, rbA, rbB);
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
    if ((uint64_t)sg.rows * (uint64_t)desc->ldc * 4ull >= (1ull << 32)) return 1;  // (the epilogue's 32-bit store offsets)
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
  if (variant != 25 && ((long)rb * ncb < 2l * ncu || nfast >= 64)) return 1;
  // x workgroups per full column block, y for the narrow one: the smaller maximum of their tile counts, in 1/100 tiles
  const long cheap = rem == 96 ? 0 : (xc == 2 ? 74 : (xc == 1 ? 71 : 68));
  int bx = 1, by = cheap ? 1 : 0;
  long best = -1;
  for (int x = 1; x * nfull + (cheap ? 1 : 0) <= ncu && x <= rb; ++x) {
    const int y = cheap ? std::min(rb, ncu - x * nfull) : 0;
    const long tf = (long)((rb + x - 1) / x) * 100, tn = cheap ? (long)((rb + y - 1) / y) * cheap : 0;
    const long t = std::max(tf, tn);
    if (best < 0 || t < best) {
      best = t;
      bx = x;
      by = y;
    }
  }
  if (cheap) {  // (no more narrow workgroups than it takes to stay under the full ones' time)
    while (by > 1 && (long)((rb + by - 2) / (by - 1)) * cheap <= (long)((rb + bx - 1) / bx) * 100) --by;
  }
  g.x = bx;
  g.y = by;
  const int grid = bx * nfull + by;
  if (grid < 1 || grid > ncu) return 1;
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
