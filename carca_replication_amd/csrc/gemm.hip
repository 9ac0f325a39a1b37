// Dense building blocks on v_mfma_f32_32x32x2_f32 (exact fp32):
//
//   carca_gemm_rows   C[m][n]  = sum_k A[m][k] * Bt[n][k] (+ epilogue)     rows = users x slots
//   carca_gemm_wgrad  dW[n][k] += sum_r dY[r][n] * X[r][k]                  contraction over rows
//
// gemm_rows is the kernel behind AllEmbedding's feats_embed / joint_embed (carca.py:86,89; 97% of
// the model's flops at n_attrs = 4096) and every input-gradient product of the backward pass;
// gemm_wgrad produces every weight gradient (the feats_embed one is as large as the forward GEMM).
//
// Kernels in this file (the launchers at the bottom pick one from the shape):
//   gemm_rows_cu_kernel      ONE 384 x 96 block per CU, hand-scheduled K step: the feature GEMM (see its own comment)
//   gemm_rows_kernel<128,96> block tile 128 x 96, K step 32, 4 waves, wave w owns rows 32w..32w+31 x all 96 columns
//                            (3 accumulator tiles), 3 blocks per CU: large products whose grid does not suit the
//                            one-block-per-CU kernel
//   gemm_rows_kernel<128,32> 32-column blocks + a 4-deep register prefetch ring: narrow outputs (joint embedding,
//                            every d-wide product of the backward pass)
//   gemm_wgrad_kernel        block tile 96 (n) x 128 (k), 32 rows per step; both operands are read from their
//                            row-major LDS tiles TRANSPOSED (lane = n resp. k), so neither dY nor X is ever transposed
//                            in memory; row splits combine through fp32 atomics.  gemm_wgrad_group_kernel runs several
//                            such products in one launch.  (The feats_embed weight gradient: wgrad_cu.hip.)
// Common: operands staged global -> registers -> LDS with rows padded to 36 floats (conflict-free 16-byte fragment
// reads), buffer loads with 32-bit offsets wherever they provably fit, blocks that share an A row block numbered so
// that they land on the same XCD (the A tile is fetched from HBM once and re-read from that XCD's L2).
#include <hip/hip_ext.h>
#include "carca_common.h"
#include <type_traits>
#include <vector>
#include "../../include/carca_hip.h"
#include "gemm_epilogue.h"
#include <cstring>
#include <string>

// Which kernel a row product took (tools/bench_configs.py names each configuration's dominant kernel with it): while
// switched on, every row-GEMM launch of this thread appends "kernel rows=.. N=.. K=.. grid=..;".
static thread_local std::string g_rows_log;
static thread_local bool g_rows_log_on = false;
void carca_rows_log(const char* kernel, const CarcaGemmDesc* d, int grid) {
  if (!g_rows_log_on || g_rows_log.size() > 60000) return;
  long rows = 0;
  for (int s = 0; s < d->nseg; ++s) rows += d->seg[s].rows;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s rows=%ld N=%d K=%d grid=%d;", kernel, rows, d->N, d->K0 + d->K1, grid);
  g_rows_log += buf;
}
extern "C" int carca_gemm_rows_log(char* out, int cap) {
  if (!out) {  // clear and switch on (cap = 0 switches off)
    g_rows_log.clear();
    g_rows_log_on = cap != 0;
    return 0;
  }
  const int n = (int)std::min<size_t>(g_rows_log.size(), cap > 0 ? (size_t)cap - 1 : 0);
  memcpy(out, g_rows_log.data(), n);
  if (cap > 0) out[n] = 0;
  return n;
}

namespace {

struct GemmDev {
  CarcaGemmDesc d;
  int rb_start[CARCA_MAX_SEGS + 1];
  int nrb, ncb;
  unsigned long long* dbg;  // phase-stamp buffer of a diagnostic run (DBG instantiation only)
  int diag;                 // timing experiments of gemm_rows_n96_kernel (tuning key 5; wrong results): 1 no loads, 2 no MFMAs, 4 plain epilogue, 8 no stores
  int has_pas;              // one-block-per-CU kernel: block nrb * ncb (one past the tiles) runs the item-row gather
  CarcaGatherArgs pas;
  // gemm_rows_sk_kernel: K steps of every full tile that the row block's cheap workgroup computes, its partial tiles
  // [nrb][ncb - 1][384 x 96] (register order) and their flags (0 = empty, 1 = ready; the taker resets its flag)
  int sk_don;
  float* sk_part;
  int* sk_flag;
  int* sk_err;  // host-visible word: a taker's bounded wait expired
  unsigned sk_spin;  // the bound: sleeps of ~0.4 us a taker spends on one flag before it gives up (launcher: 2^23, ~5 s)
  int sk_withhold;   // TEST ONLY (tuning key 13): 1 + index of the one flag its giver does not raise; 0 = none
  int skc_cheap;     // gemm_rows_skc_kernel: cost of a K step of the narrow column block's tile in 1/100 of a full tile's step
  int skc_ov, skc_ov_lone;  // ... and what OWNING a row block costs (tail tile, partials taken, epilogue) in K steps of the
                            // workgroup's kind: the stretches are cut equal in steps + that overhead (step 3 of the kernel)
};


// The item-row gather (carca.py:87-88) as the PASSENGER workgroup of gemm_rows_sk_kernel, through LDS-DMA.  carca_gather_rows
// keeps its rows in registers and the compiler serialises them beyond ~16 in flight (a lone workgroup: ~455 us for C2's
// 19 k rows, longer than the product it rides in).  Here a wave requests R rows -- each one or two buffer_load ... lds of 64
// dwords, no destination registers, the row's table address in the scalar resource -- waits once, and copies them out of
// LDS scaled; the ids of the next batch are requested right behind the rows of this one.  12 waves x 24 rows x 360 bytes in
// flight: ~390 us for C2's 19 k rows beside the tiles (20 ns per row; the register version: 540 us).
template <int R, int RS>
__device__ __forceinline__ void gather_rows_dma(const CarcaGatherArgs& ga, float* lds, const int wave, const int nwaves,
                                                const int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  float* my = lds + wave * (R * RS);
  const int nb = (ga.total_rows + R - 1) / R;
  auto id_of = [&](int b) -> int {
    const int row = min(b * R + min(lane, R - 1), ga.total_rows - 1);
    int s = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (j < ga.nseg && row >= ga.row_start[j]) s = j;
    return ga.ids[s][row - ga.row_start[s]];
  };
  const unsigned v0 = (unsigned)min(lane, ga.d - 1) * 4u, v1 = (unsigned)min(64 + lane, ga.d - 1) * 4u;
  const bool second = ga.d > 64 && lane < RS - 64;
  int idv = id_of(wave);
  for (int b = wave; b < nb; b += nwaves) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int id = __builtin_amdgcn_readlane(idv, i);
      const __amdgpu_buffer_rsrc_t rs = carca_rsrc(ga.items_w + (size_t)id * ga.d);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + i * RS), 4, v0, 0, 0, 0);
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(my + i * RS + 64), 4, v1, 0, 0, 0);
    }
    // (the next batch's ids BEHIND the rows: the compiler drains every outstanding load before an LDS-DMA instruction)
    const int idn = b + nwaves < nb ? id_of(b + nwaves) : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int row = b * R + i;
      if (row < ga.total_rows) {
        float* dst = ga.zq + (size_t)row * ga.ldz;
        if (lane < ga.d) dst[lane] = my[i * RS + lane] * ga.scale;
        if (RS > 64 && lane < RS - 64 && 64 + lane < ga.d) dst[64 + lane] = my[i * RS + 64 + lane] * ga.scale;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this batch has left LDS before the next one's rows are requested)
    idv = idn;
  }
}

template <int BM, int BN, int BK, int PF = 1, bool BUF = false>
__device__ __forceinline__ void gemm_rows_body(const GemmDev& args, const int id) {
  constexpr int NW = BM / 32, NT = NW * 64, LS = BK + 4, TN = BN / 32;
  constexpr int C4 = BK / 4;  // float4 slots per tile row
  constexpr int A_SLOTS = BM * C4, B_SLOTS = BN * C4;
  constexpr int A_PER = (A_SLOTS + NT - 1) / NT, B_PER = (B_SLOTS + NT - 1) / NT;
  static_assert(BK % 8 == 0 && BN % 32 == 0 && BM % 32 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) float As[BM * LS];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LS];

  const CarcaGemmDesc& D = args.d;
  // ---- block id -> (row block, col block): ids that are equal mod 8 land on one XCD under round-robin placement;
  // renumber them contiguously per XCD (bijective for any grid size, no phantom blocks: a padded grid can push real
  // blocks into a second round when the grid is sized to the chip) so that the column blocks of a row block share an L2
  const int total = args.nrb * args.ncb;
  const int xcd = id & 7, q8 = total >> 3, r8 = total & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int rb = wg / args.ncb, cb = wg - rb * args.ncb;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = (rb - args.rb_start[s]) * BM;  // first row of this block inside the segment
  const int n0 = cb * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = (D.K0 + BK - 1) / BK, nt1 = (D.K1 + BK - 1) / BK;
  const int ntiles = nt0 + nt1;

  f32x4 ra[PF][A_PER], rbv[PF][B_PER];  // PF tiles of global loads in flight (register ring, compile-time indexed)
  // element offsets of this thread's A rows inside each k-source (loop invariant; handles [B, T, K] views)
  size_t aoff0[A_PER], aoff1[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int slot = tid + i * NT;
    const int gr = min(row0 + slot / C4, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    aoff0[i] = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
               : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                               : (size_t)gr * D.lda0;
    aoff1[i] = sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
  }

  auto load4 = [](const float* p, bool full, int kk, int klen) -> f32x4 {
    if (full) return *reinterpret_cast<const f32x4_u*>(p);
    f32x4 v;
    v[0] = kk + 0 < klen ? p[0] : 0.f;
    v[1] = kk + 1 < klen ? p[1] : 0.f;
    v[2] = kk + 2 < klen ? p[2] : 0.f;
    v[3] = kk + 3 < klen ? p[3] : 0.f;
    return v;
  };
  // BUF: full tiles are fetched with buffer loads (scalar resource + 32-bit lane offset + scalar k offset; see
  // gload4 in carca_common.h for why) -- the launcher sets it when every offset provably fits 32 bits
  auto load_tile = [&](int t, f32x4 (&ra_)[A_PER], f32x4 (&rb_)[B_PER]) {
    const bool src1 = t >= nt0;
    const int k0 = (src1 ? t - nt0 : t) * BK;
    const int klen = src1 ? D.K1 : D.K0;
    const float* abase = src1 ? sg.a1 : sg.a0;
    const float* bbase = src1 ? D.bt1 : D.bt0;
    const int ldb = src1 ? D.ldb1 : D.ldb0;
    const bool full = k0 + BK <= klen;  // block-uniform
    if (BUF && full) {
      const __amdgpu_buffer_rsrc_t ar = carca_rsrc(abase), br = carca_rsrc(bbase);
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int slot = tid + i * NT;
        if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
        const unsigned off = (unsigned)(((src1 ? aoff1[i] : aoff0[i]) + (slot % C4) * 4) * sizeof(float));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ar, off, k0 * (int)sizeof(float), 0);
        ra_[i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int slot = tid + i * NT;
        if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
        const int r = slot / C4, c4 = slot - r * C4;
        const unsigned off = (unsigned)(((size_t)min(n0 + r, D.N - 1) * ldb + c4 * 4) * sizeof(float));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(br, off, k0 * (int)sizeof(float), 0);
        rb_[i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int c4 = slot % C4;
      ra_[i] = load4(abase + (src1 ? aoff1[i] : aoff0[i]) + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      const int gn = min(n0 + r, D.N - 1);
      rb_[i] = load4(bbase + (size_t)gn * ldb + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
  };
  auto store_tile = [&](const f32x4 (&ra_)[A_PER], const f32x4 (&rb_)[B_PER]) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int slot = tid + i * NT;
      if (A_SLOTS % NT != 0 && slot >= A_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&As[r * LS + c4 * 4]) = ra_[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int slot = tid + i * NT;
      if (B_SLOTS % NT != 0 && slot >= B_SLOTS) break;
      const int r = slot / C4, c4 = slot - r * C4;
      *reinterpret_cast<f32x4*>(&Bs[r * LS + c4 * 4]) = rb_[i];
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * LS + 4 * lh];
  const float* b_frag = &Bs[lr * LS + 4 * lh];

  auto compute_tile = [&]() {
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag + kg * 8);
      f32x4 b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS + kg * 8);
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
    }
  };
  if constexpr (PF == 1) {
    load_tile(0, ra[0], rbv[0]);
    store_tile(ra[0], rbv[0]);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
      if (t + 1 < ntiles) load_tile(t + 1, ra[0], rbv[0]);
      compute_tile();
      __syncthreads();
      if (t + 1 < ntiles) {
        store_tile(ra[0], rbv[0]);
        __syncthreads();
      }
    }
  } else {
    // short K loops at low occupancy (narrow outputs): the global latency of a tile is longer than its MFMAs,
    // so keep PF tiles of loads in flight in a register ring; the ring slot is a compile-time index.
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (u < ntiles) load_tile(u, ra[u], rbv[u]);
    for (int t0 = 0; t0 < ntiles; t0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int t = t0 + u;
        if (t < ntiles) {
          store_tile(ra[u], rbv[u]);
          __syncthreads();
          if (t + PF < ntiles) load_tile(t + PF, ra[u], rbv[u]);
          compute_tile();
          __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: D row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 ---------------------
  gemm_rows_epilogue<TN>(D, sg, acc, n0, row0 + wave * 32, lr, lh);
}

template <int BM, int BN, int BK, int PF = 1, bool BUF = false>
__global__ __launch_bounds__((BM / 32) * 64) void gemm_rows_kernel(const GemmDev args) {
  gemm_rows_body<BM, BN, BK, PF, BUF>(args, (int)blockIdx.x);
}

// Independent narrow products of a backward pass (dQ.W_Q beside dK.W_K + dV.W_V) are ~150-block, 15 us launches that
// leave half the chip idle: side by side in ONE launch they cost one.  Descriptors travel by value (kernel arguments);
// block -> (product, local block id); one body instantiation, so LDS and registers are those of the single kernel.
constexpr int GEMM_GROUP_MAX = 4;
struct GemmGroup {
  GemmDev g[GEMM_GROUP_MAX];
  int block_start[GEMM_GROUP_MAX + 1];
  int n;
};
template <int BM, int BN, int BK, int PF, bool BUF>
__global__ __launch_bounds__((BM / 32) * 64) void gemm_rows_group_kernel(const GemmGroup grp) {
  int p = 0;
#pragma unroll
  for (int i = 1; i < GEMM_GROUP_MAX; ++i)
    if (i < grp.n && (int)blockIdx.x >= grp.block_start[i]) p = i;
  gemm_rows_body<BM, BN, BK, PF, BUF>(grp.g[p], (int)blockIdx.x - grp.block_start[p]);
}

// ---------------------------------------------------------------------------------------------------
// gemm_rows_cu_kernel: the same product for the ONE launch that carries 97 % of the model's flops
// (feats_embed at n_attrs = 4096: ~19k rows x 4102 x 450).  One 768-thread block per CU, tile 384 x 96:
// 36 of the 9060 32x32 output tiles per CU (255 blocks on 256 CUs, one round) like the 128 x 96 kernel's
// three co-resident blocks, but the B tile is staged once instead of three times (107 instead of
// 149 bytes through L1 per MFMA).  With a single lock-stepped block nothing else covers a wave's
// non-MFMA instructions, and the SIMD serves its three waves oldest-first (measured: barrier waits
// 4200 / 2200 / 300 cycles per step for the 1st / 2nd / 3rd wave of a SIMD), so the step is scheduled
// by hand: every LDS read, LDS write and global load sits in the shadow of one of the wave's own
// MFMAs, fragments are double-buffered in registers one 8-k group ahead, LDS is double-buffered so
// that there is ONE barrier per K step, global loads run two tiles ahead on a scalar base + constant
// per-lane offsets (no address arithmetic in the loop), and all LDS addresses are immediates.
// TN = 3: tile 384 x 96 (the C2 feature GEMM: 5 column blocks of N = 450).  TN = 4: tile 384 x 128 for N that 128-wide
// blocks cover in one round where 96-wide ones need two (g = 640: 5 x 51 = 255 blocks instead of 7 x 51 = 357); its B
// tile is 1024 staging slots for 768 threads: every thread carries two, the second one live for threads 0..255 and a
// clamped load + a store into a dummy LDS strip for the others (the pipelined body stays branch-free).
template <int DBG, int TN = 3>
__global__ __launch_bounds__(768) void gemm_rows_cu_kernel(const GemmDev args) {
  constexpr int BM = 384, BN = 32 * TN, BK = 32, NW = 12, NT = 768, LS = BK + 4, C4 = BK / 4;
  constexpr int A_PER = BM * C4 / NT, B_PER = (BN * C4 + NT - 1) / NT;  // 4 + 1 (or 2) staged float4 per thread and tile
  static_assert(TN == 3 || TN == 4, "column tiles per block");
  constexpr int A_BUF = BM * LS, B_BUF = BN * LS;  // floats per LDS buffer
  constexpr int DUMMY = (B_PER * NT > BN * C4) ? (B_PER * NT - BN * C4) * 4 : 0;  // floats: dead B slots land here
  __shared__ __attribute__((aligned(16))) float As[2 * A_BUF];
  __shared__ __attribute__((aligned(16))) float Bs[2 * B_BUF + DUMMY];

  const CarcaGemmDesc& D = args.d;
  const int id = blockIdx.x, total = args.nrb * args.ncb;
  if (args.has_pas && id == total) {  // the passenger: this workgroup only copies item rows (an otherwise idle CU)
    carca_gather_rows<16>(args.pas, (int)threadIdx.x >> 6, NW, (int)threadIdx.x & 63);
    return;
  }
  const int xcd = id & 7, q8 = total >> 3, r8 = total & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int rb = wg / args.ncb, cb = wg - rb * args.ncb;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = (rb - args.rb_start[s]) * BM;
  const int n0 = cb * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nfast = D.K0 / BK;  // full tiles of k-source 0: the pipelined loop
  const int nt0 = (D.K0 + BK - 1) / BK, nt1 = (D.K1 + BK - 1) / BK;
  const int ntiles = nt0 + nt1;

  // ---- staging slots: element offsets are loop invariants; the fast loop adds them to a scalar base ----
  size_t aoff0[A_PER], aoff1[A_PER];
  unsigned a_byte[A_PER];  // byte offset of the slot's 16 bytes inside tile 0 of k-source 0 (host: < 4 GiB)
  int a_lds[A_PER];        // float index inside an A buffer
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int slot = tid + i * NT, r = slot / C4, c4 = slot - r * C4;
    const int gr = min(row0 + r, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    aoff0[i] = sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
               : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                               : (size_t)gr * D.lda0;
    aoff1[i] = sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
    a_byte[i] = (unsigned)((aoff0[i] + c4 * 4) * sizeof(float));
    a_lds[i] = r * LS + c4 * 4;
  }
  int b_gn[B_PER], b_c4[B_PER];
  int b_at[2][B_PER];  // float index inside Bs of the slot's 16 bytes, per LDS buffer (dead slots: the dummy strip)
  unsigned b_byte[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int slot = tid + i * NT;
    const bool live = DUMMY == 0 || slot < BN * C4;
    const int r = live ? slot / C4 : 0;
    b_c4[i] = slot % C4;
    b_gn[i] = min(n0 + r, D.N - 1);
    b_byte[i] = (unsigned)(((size_t)b_gn[i] * D.ldb0 + b_c4[i] * 4) * sizeof(float));
    b_at[0][i] = live ? r * LS + b_c4[i] * 4 : 2 * B_BUF + (slot - BN * C4) * 4;
    b_at[1][i] = live ? B_BUF + r * LS + b_c4[i] * 4 : 2 * B_BUF + (slot - BN * C4) * 4;
  }

  f32x4 ra[A_PER], rbv[B_PER];
  // Buffer loads: scalar resource + 32-bit per-lane offset + scalar tile offset -- no address arithmetic in the
  // loop and half the address data of a 64-bit global_load per instruction.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sg.a0, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)D.bt0, 0, -1, 0x00020000);
  int k_byte = 0;  // scalar: byte offset of the NEXT tile to load inside a row
  auto load_fast = [&](int i) {  // slot i of that tile (i >= A_PER: B slot i - A_PER)
    const int ia = i < A_PER ? i : 0, ib = i < A_PER ? 0 : i - A_PER;
    const u32x4 v = i < A_PER ? __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_byte[ia], k_byte, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_byte[ib], k_byte, 0);
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
    if (i < A_PER)
      ra[ia] = f;
    else
      rbv[ib] = f;
  };
  auto store_slot = [&](int i, int buf) {
    if (i < A_PER) {
      *reinterpret_cast<f32x4*>(&As[buf * A_BUF + a_lds[i]]) = ra[i];
    } else {
      // (dead slots -- the second B slot of threads >= 256 at TN = 4 -- write their own 16 bytes of the dummy strip)
      *reinterpret_cast<f32x4*>(&Bs[b_at[buf][i - A_PER]]) = rbv[i - A_PER];
    }
  };
  auto load4 = [](const float* p, bool full, int kk, int klen) -> f32x4 {
    if (full) return *reinterpret_cast<const f32x4_u*>(p);
    f32x4 v;
    v[0] = kk + 0 < klen ? p[0] : 0.f;
    v[1] = kk + 1 < klen ? p[1] : 0.f;
    v[2] = kk + 2 < klen ? p[2] : 0.f;
    v[3] = kk + 3 < klen ? p[3] : 0.f;
    return v;
  };
  auto load_tail = [&](int t) {  // any tile, either k-source, ragged K (the few tiles behind the fast loop)
    const bool src1 = t >= nt0;
    const int k0 = (src1 ? t - nt0 : t) * BK;
    const int klen = src1 ? D.K1 : D.K0;
    const float* abase = src1 ? sg.a1 : sg.a0;
    const float* bbase = src1 ? D.bt1 : D.bt0;
    const int ldb = src1 ? D.ldb1 : D.ldb0;
    const bool full = k0 + BK <= klen;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c4 = (tid + i * NT) % C4;
      ra[i] = load4(abase + (src1 ? aoff1[i] : aoff0[i]) + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      rbv[i] = load4(bbase + (size_t)b_gn[i] * ldb + k0 + b_c4[i] * 4, full, k0 + b_c4[i] * 4, klen);
  };

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * LS + 4 * lh];
  const float* b_frag = &Bs[lr * LS + 4 * lh];
  f32x4 fa0, fa1, fb0[TN], fb1[TN];  // fragment sets of two consecutive 8-k groups

#define CARCA_PIN() __builtin_amdgcn_sched_barrier(0)
  // fragment read j of group kg from LDS buffer `buf` into set (fa, fb): j = 0 is A, 1..3 the B tiles
  auto read_frag = [&](int j, int buf, int kg, f32x4& fa, f32x4(&fb)[TN]) {
    if (j == 0)
      fa = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + kg * 8);
    else
      fb[j - 1] = *reinterpret_cast<const f32x4*>(b_frag + buf * B_BUF + (j - 1) * 32 * LS + kg * 8);
  };
  // 4 TN MFMAs of one 8-k group; aux(i) is issued behind MFMA i and must be independent of it
  constexpr int NR = TN + 1;  // fragment reads per group: A, then the TN B tiles
  auto mfma_group = [&](const f32x4& fa, const f32x4(&fb)[TN], auto&& aux) {
#pragma unroll
    for (int i = 0; i < 4 * TN; ++i) {
      acc[i % TN] = mfma32(fa[i / TN], fb[i % TN][i / TN], acc[i % TN]);
      CARCA_PIN();
      aux(i);
      CARCA_PIN();
    }
  };
  unsigned long long w_vm = 0, w_bar = 0, t_begin = 0;

  // one K step on tile t (LDS buffer CUR); tile t+1 is in registers, tile t+2 gets loaded
  // one K step on tile t (LDS buffer CUR); tile t+1 is in registers, tile t+2 gets loaded.  Gaps 0..3 of every group
  // carry the fragment reads of the next group; gaps 4..8 of group 1 the LDS writes of tile t+1 and gaps 4..8 of
  // group 2 the loads of tile t+2 (behind the writes: the staging registers are reused).
  auto step = [&](auto cur_tag, int t) {
    constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
    const bool has1 = t + 1 < nfast, has2 = t + 2 < nfast;
    mfma_group(fa0, fb0, [&](int i) {
      if (i < NR) read_frag(i, CUR, 1, fa1, fb1);
    });
    if constexpr (DBG) {
      const unsigned long long ta = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      w_vm += __builtin_amdgcn_s_memtime() - ta;
      CARCA_PIN();
    }
    static_assert(NR + A_PER + B_PER <= 4 * TN, "a group's gaps hold its reads and the tile's staging slots");
    mfma_group(fa1, fb1, [&](int i) {
      if (i < NR)
        read_frag(i, CUR, 2, fa0, fb0);
      else if (i < NR + A_PER + B_PER && has1)
        store_slot(i - NR, NXT);
    });
    mfma_group(fa0, fb0, [&](int i) {
      if (i < NR)
        read_frag(i, CUR, 3, fa1, fb1);
      else if (i < NR + A_PER + B_PER && has2)
        load_fast(i - NR);
    });
    if (has2) k_byte += BK * sizeof(float);
    CARCA_PIN();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own LDS writes of tile t+1 and reads of tile t retired
    if constexpr (DBG) {
      const unsigned long long tb = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_barrier();
      w_bar += __builtin_amdgcn_s_memtime() - tb;
    } else {
      __builtin_amdgcn_s_barrier();  // raw: the loads of tile t+2 stay in flight across it
    }
    CARCA_PIN();
    mfma_group(fa1, fb1, [&](int i) {
      if (i < NR && has1) read_frag(i, NXT, 0, fa0, fb0);
    });
  };

  if (nfast > 0) {
    // prologue: tile 0 -> LDS buffer 0, tile 1 -> registers
#pragma unroll
    for (int i = 0; i < A_PER + B_PER; ++i) load_fast(i);
    k_byte += BK * sizeof(float);
#pragma unroll
    for (int i = 0; i < A_PER + B_PER; ++i) store_slot(i, 0);
    if (nfast > 1) {
#pragma unroll
      for (int i = 0; i < A_PER + B_PER; ++i) load_fast(i);
      k_byte += BK * sizeof(float);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; ++j) read_frag(j, 0, 0, fa0, fb0);
    if constexpr (DBG) t_begin = __builtin_amdgcn_s_memtime();
    int t = 0;
    for (; t + 1 < nfast; t += 2) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nfast) step(std::integral_constant<int, 0>{}, t);
    if constexpr (DBG) {
      if (args.dbg && lane == 0) {
        unsigned long long* o = args.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
        o[0] = __builtin_amdgcn_s_memtime() - t_begin;
        o[1] = w_vm;
        o[2] = w_bar;
        o[3] = 0;
      }
    }
  }
#undef CARCA_PIN
  // ---- tail tiles (ragged end of k-source 0, all of k-source 1): plain load -> LDS -> multiply ----
  // (the context columns as ONE 8-k group where they are all of the tail: see cu_tile)
  const bool ctx8 = D.K1 >= 4 && D.K1 <= 8 && D.K0 % BK == 0 && !(args.diag & 64);
  if (ctx8) {
    const int xr = tid >> 1, xh = tid & 1;
    const int cs = min(4 * xh, D.K1 - 4), sh = 4 * xh - cs;
    const int gr = min(row0 + xr, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t ao = sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
    const bool bx_live = tid < 2 * BN;
    const f32x4 av = *reinterpret_cast<const f32x4_u*>(sg.a1 + ao + cs);
    const f32x4 bv = *reinterpret_cast<const f32x4_u*>(D.bt1 + (size_t)min(n0 + (bx_live ? xr : 0), D.N - 1) * D.ldb1 + cs);
    auto fix = [&](const f32x4 v) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = sh == 0 ? v[e] : (sh == 1 ? v[(e + 1) & 3] : (sh == 2 ? v[(e + 2) & 3] : v[(e + 3) & 3]));
        o[e] = (4 * xh + e < D.K1) ? x : 0.f;
      }
      return o;
    };
    __syncthreads();  // every wave is done with both LDS buffers
    *reinterpret_cast<f32x4*>(&As[xr * LS + xh * 4]) = fix(av);
    if (bx_live) *reinterpret_cast<f32x4*>(&Bs[xr * LS + xh * 4]) = fix(bv);
    __syncthreads();
    const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag);
    f32x4 b[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
  }
  for (int t = ctx8 ? ntiles : nfast; t < ntiles; ++t) {
    load_tail(t);
    __syncthreads();  // every wave is done with both LDS buffers
#pragma unroll
    for (int i = 0; i < A_PER + B_PER; ++i) store_slot(i, 0);
    __syncthreads();
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag + kg * 8);
      f32x4 b[TN];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS + kg * 8);
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
    }
  }

  // ---- epilogue (same as gemm_rows_kernel) ----------------------------------------------------------
  gemm_rows_epilogue<TN>(D, sg, acc, n0, row0 + wave * 32, lr, lh);
}

// ---------------------------------------------------------------------------------------------------
// gemm_rows_sk_kernel: gemm_rows_cu_kernel with the PADDING of the last column block given back.
// N = 450 is 4 x 96 + 66: the fifth column block of every row block multiplies 96 columns for 66 results, and since
// every workgroup owns one tile the launch lasts as long as a full tile -- 6.3 % of the MFMA work is padding (DESIGN,
// section "Next").  Here the fifth tile is CHEAP: two 32-column MFMA tiles + the last XC <= 2 columns as VALU dot
// products on the A fragments the wave holds anyway (4 XC fused multiply-adds per 8-k group beside 8 MFMAs), and the time
// its workgroup saves goes to the four others: it computes the LAST `don` K steps of each of their tiles first (a
// 384 x 96 partial tile each, handed over through memory with a flag), then its own tile; the owners run `don` steps less
// and add the partial in their epilogue -- which is ~0.75 of the kernel later, so nobody ever waits (the flag is there
// for correctness).  A giver waits for nobody, every workgroup is resident (one round, one workgroup per CU): no cycle.
// cu_tile is gemm_rows_cu_kernel's body over a K-step range [t_lo, t_hi) with the three endings.
enum { SK_PLAIN = 0, SK_GIVE = 1, SK_TAKE = 2, SK_DYN = 3 };
// What a tile of gemm_rows_skc_kernel adds to cu_tile's arguments (MODE == SK_DYN): the ending as a value, the number of
// partial tiles a taker adds (those of the workgroups behind it: part / flag of the first, one slot further each), and the
// kept rows of the tile's segment.
struct SkcTile {
  int mode, ntake;
  int seg, row0, live;
  const int* rows;
};
template <int TN, int XC, int MODE>
__device__ __forceinline__ void cu_tile(const GemmDev& args, float* __restrict__ As, float* __restrict__ Bs, const int rb,
                                        const int n0, const int t_lo, const int t_hi, const bool with_tail,
                                        float* __restrict__ part, int* flag, const SkcTile dyn = SkcTile{}) {
  constexpr bool IDX = MODE == SK_DYN;  // rows through the kept-row list
  constexpr int BM = 384, BNS = 32 * TN + XC, BK = 32, NW = 12, NT = 768, LS = BK + 4, C4 = BK / 4;
  constexpr int A_PER = BM * C4 / NT, B_PER = (BNS * C4 + NT - 1) / NT;
  constexpr int A_BUF = BM * LS, B_BUF = BNS * LS;
  constexpr int XCA = XC > 0 ? XC : 1;
  static_assert(B_PER == 1 && 2 * B_BUF + (NT - BNS * C4) * 4 <= 2 * 96 * LS + 1024, "B tile: one slot per thread, inside the kernel's Bs");
  const CarcaGemmDesc& D = args.d;
  int s = 0;
  if constexpr (IDX) {
    s = dyn.seg;
  } else {
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  }
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = IDX ? dyn.row0 : (rb - args.rb_start[s]) * BM;
  const int nrows = IDX ? dyn.live : sg.rows;  // rows of the segment this kernel multiplies
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt0 = (D.K0 + BK - 1) / BK, nt1 = (D.K1 + BK - 1) / BK;
  const int ntiles = nt0 + nt1;

  // element offset of staging slot i's row inside k-source 0 / 1 (recomputed where the tail tiles need it: kept in registers
  // across the K loop -- sixteen VGPRs of 64-bit offsets -- they were what the 168-register kernel spilled)
  auto a_off = [&](int i, bool src1) -> size_t {
    const int slot = tid + i * NT, r = slot / C4;
    int gr = min(row0 + r, nrows - 1);
    if constexpr (IDX) gr = dyn.rows[gr];
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    if (src1) return sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
    return sg.a0_gather ? (size_t)sg.ids[gr] * D.lda0
           : sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0
                           : (size_t)gr * D.lda0;
  };
  unsigned a_byte[A_PER];
  int a_lds[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int slot = tid + i * NT, r = slot / C4, c4 = slot - r * C4;
    a_byte[i] = (unsigned)((a_off(i, false) + c4 * 4) * sizeof(float));
    a_lds[i] = r * LS + c4 * 4;
  }
  const bool b_live = tid < BNS * C4;  // (threads past the B tile's slots load a clamped row and store into the dummy strip)
  const int b_r = b_live ? tid / C4 : 0, b_c4 = tid % C4;
  const int b_gn = min(n0 + b_r, D.N - 1);
  const unsigned b_byte = (unsigned)(((size_t)b_gn * D.ldb0 + b_c4 * 4) * sizeof(float));
  const int b_at0 = b_live ? b_r * LS + b_c4 * 4 : 2 * B_BUF + (tid - BNS * C4) * 4;
  const int b_at1 = b_live ? B_BUF + b_r * LS + b_c4 * 4 : 2 * B_BUF + (tid - BNS * C4) * 4;

  f32x4 ra[A_PER], rbv;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sg.a0, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)D.bt0, 0, -1, 0x00020000);
  int k_byte = t_lo * BK * (int)sizeof(float);  // scalar: byte offset of the NEXT tile to load inside a row
  auto load_fast = [&](int i) {
    const u32x4 v = i < A_PER ? __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_byte[i < A_PER ? i : 0], k_byte, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_byte, k_byte, 0);
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
    if (i < A_PER)
      ra[i < A_PER ? i : 0] = f;
    else
      rbv = f;
  };
  auto store_slot = [&](int i, int buf) {
    if (i < A_PER)
      *reinterpret_cast<f32x4*>(&As[buf * A_BUF + a_lds[i < A_PER ? i : 0]]) = ra[i < A_PER ? i : 0];
    else
      *reinterpret_cast<f32x4*>(&Bs[buf ? b_at1 : b_at0]) = rbv;
  };
  auto load4 = [](const float* p, bool full, int kk, int klen) -> f32x4 {
    if (full) return *reinterpret_cast<const f32x4_u*>(p);
    f32x4 v;
    v[0] = kk + 0 < klen ? p[0] : 0.f;
    v[1] = kk + 1 < klen ? p[1] : 0.f;
    v[2] = kk + 2 < klen ? p[2] : 0.f;
    v[3] = kk + 3 < klen ? p[3] : 0.f;
    return v;
  };
  auto load_tail = [&](int t) {
    const bool src1 = t >= nt0;
    const int k0 = (src1 ? t - nt0 : t) * BK;
    const int klen = src1 ? D.K1 : D.K0;
    const float* abase = src1 ? sg.a1 : sg.a0;
    const float* bbase = src1 ? D.bt1 : D.bt0;
    const int ldb = src1 ? D.ldb1 : D.ldb0;
    const bool full = k0 + BK <= klen;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c4 = (tid + i * NT) % C4;
      ra[i] = load4(abase + a_off(i, src1) + k0 + c4 * 4, full, k0 + c4 * 4, klen);
    }
    rbv = load4(bbase + (size_t)b_gn * ldb + k0 + b_c4 * 4, full, k0 + b_c4 * 4, klen);
  };

  f32x16 acc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
  f32x4 xacc[XCA];  // per column: the four k positions of the lane's quad, summed at the end
#pragma unroll
  for (int c = 0; c < XCA; ++c) xacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 31, lh = lane >> 5;
  const float* a_frag = &As[(wave * 32 + lr) * LS + 4 * lh];
  const float* b_frag = &Bs[lr * LS + 4 * lh];
  const float* x_frag = &Bs[(32 * TN) * LS + 4 * lh];  // the XC extra rows of the B tile: one k quad per half wave
  f32x4 fa0, fa1, fb0[TN], fb1[TN], fx0[XCA], fx1[XCA];

#define CARCA_PIN() __builtin_amdgcn_sched_barrier(0)
  constexpr int NR = TN + 1, NRX = NR + XC;
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
        if (i < XC) {
          // two v_pk_fma_f32, written out: left to the compiler the same four FMAs cost 70 spilled VGPRs (it re-plans the
          // fragment registers of the whole pinned step around them)
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
  static_assert(NRX <= 4 * TN && NR + A_PER + B_PER <= 4 * TN, "a group's gaps hold its reads and the tile's staging slots");
  auto step = [&](auto cur_tag, int t) {
    constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
    const bool has1 = t + 1 < t_hi, has2 = t + 2 < t_hi;
    mfma_group(fa0, fb0, fx0, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 1, fa1, fb1, fx1);
    });
    mfma_group(fa1, fb1, fx1, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 2, fa0, fb0, fx0);
      if (i >= NR && i < NR + A_PER + B_PER && has1) store_slot(i - NR, NXT);
    });
    mfma_group(fa0, fb0, fx0, [&](int i) {
      if (i < NRX) read_frag(i, CUR, 3, fa1, fb1, fx1);
      if (i >= NR && i < NR + A_PER + B_PER && has2) load_fast(i - NR);
    });
    if (has2) k_byte += BK * sizeof(float);
    CARCA_PIN();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // raw: the loads of tile t+2 stay in flight across it
    CARCA_PIN();
    mfma_group(fa1, fb1, fx1, [&](int i) {
      if (i < NRX && has1) read_frag(i, NXT, 0, fa0, fb0, fx0);
    });
  };

  __syncthreads();  // (a workgroup that runs several tiles: the previous one is done with both LDS buffers)
  if (t_hi > t_lo) {
#pragma unroll
    for (int i = 0; i < A_PER + B_PER; ++i) load_fast(i);
    k_byte += BK * sizeof(float);
#pragma unroll
    for (int i = 0; i < A_PER + B_PER; ++i) store_slot(i, 0);
    if (t_hi - t_lo > 1) {
#pragma unroll
      for (int i = 0; i < A_PER + B_PER; ++i) load_fast(i);
      k_byte += BK * sizeof(float);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NRX; ++j) read_frag(j, 0, 0, fa0, fb0, fx0);
    int t = t_lo;
    for (; t + 1 < t_hi; t += 2) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < t_hi) step(std::integral_constant<int, 0>{}, t);
  }
#undef CARCA_PIN
  // The context columns (k-source 1 with 4 <= K1 <= 8 behind a K0 of whole 32-steps: AllEmbedding's six) as ONE 8-k group
  // (round 5; the scheme of gemm_rows_cus_kernel's context item, gemm_stream.hip): every thread requests one 16-byte group
  // of its row -- thread = (row tid >> 1, half tid & 1), the second half clamped to END at K1 and shifted / zeroed when
  // stored --, two barriers, 4 TN MFMAs.  The general tail below stages a whole 32-wide tile through element-wise
  // conditional loads (twenty dword loads per thread) and multiplies four 8-k groups, three of them zeros: ~7 us per
  // owned tile against ~3.  Same products in the same order (the other groups added zeros): bit-identical results.
  const bool ctx8 = with_tail && D.K1 >= 4 && D.K1 <= 8 && D.K0 % BK == 0 && !(args.diag & 64);
  if (ctx8) {
    const int xr = tid >> 1, xh = tid & 1;
    const int cs = min(4 * xh, D.K1 - 4), sh = 4 * xh - cs;
    int gr = min(row0 + xr, nrows - 1);
    if constexpr (IDX) gr = dyn.rows[gr];
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t ao = sg.a1_bstride ? (size_t)ub * sg.a1_bstride + (size_t)ut * D.lda1 : (size_t)gr * D.lda1;
    const bool bx_live = tid < 2 * BNS;
    const f32x4 av = *reinterpret_cast<const f32x4_u*>(sg.a1 + ao + cs);
    const f32x4 bv = *reinterpret_cast<const f32x4_u*>(D.bt1 + (size_t)min(n0 + (bx_live ? xr : 0), D.N - 1) * D.ldb1 + cs);
    auto fix = [&](const f32x4 v) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = sh == 0 ? v[e] : (sh == 1 ? v[(e + 1) & 3] : (sh == 2 ? v[(e + 2) & 3] : v[(e + 3) & 3]));
        o[e] = (4 * xh + e < D.K1) ? x : 0.f;
      }
      return o;
    };
    __syncthreads();  // every wave is done with both LDS buffers
    *reinterpret_cast<f32x4*>(&As[xr * LS + xh * 4]) = fix(av);
    *reinterpret_cast<f32x4*>(&Bs[bx_live ? xr * LS + xh * 4 : 2 * B_BUF + ((tid - 2 * BNS) & 255) * 4]) = fix(bv);
    __syncthreads();
    const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag);
    f32x4 b[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
    if constexpr (XC > 0) {
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(x_frag + c * LS);
        xacc[c] = __builtin_elementwise_fma(a, w, xacc[c]);
      }
    }
  } else if (with_tail) {
    for (int t = D.K0 / BK; t < ntiles; ++t) {
      load_tail(t);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < A_PER + B_PER; ++i) store_slot(i, 0);
      __syncthreads();
#pragma unroll
      for (int kg = 0; kg < BK / 8; ++kg) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(a_frag + kg * 8);
        f32x4 b[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(b_frag + tn * 32 * LS + kg * 8);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tn] = mfma32(a[st], b[tn][st], acc[tn]);
        if constexpr (XC > 0) {
#pragma unroll
          for (int c = 0; c < XC; ++c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(x_frag + c * LS + kg * 8);
            xacc[c] = __builtin_elementwise_fma(a, w, xacc[c]);
          }
        }
      }
    }
  }

  const int mode = MODE == SK_DYN ? dyn.mode : MODE;
  if (mode == SK_GIVE) {
    // the partial tile in register order (256 contiguous bytes per wave store), written through the XCD's L2 (the taker
    // may sit on another XCD), then the flag
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __hip_atomic_store(&part[(tn * 16 + r) * NT + tid], acc[tn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (XC > 0) {
#pragma unroll
      for (int c = 0; c < XC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __hip_atomic_store(&part[(TN * 16 + c * 4 + e) * NT + tid], xacc[c][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the hand-over is MI355X_MICROARCH.md's "drained sc1" form: every byte of the partial leaves through an agent-scope
    // (sc1, write-through) store, every storing wave drains its stores, the workgroup meets, THEN one lane raises the flag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && (int)(flag - args.sk_flag) + 1 != args.sk_withhold)
      __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int ntake = MODE == SK_DYN ? (mode == SK_TAKE ? dyn.ntake : 0) : (MODE == SK_TAKE ? 1 : 0);
  for (int tk = 0; tk < ntake; ++tk, part += 384 * 96, ++flag) {
    if (tk > 0) __syncthreads();  // (everybody has read the previous flag's partial before lane 0 moves on)
    if (tid == 0) {
      // (the giver is resident -- one round of workgroups -- and ends before its takers by the launcher's balance: the wait
      // is a few us.  It is BOUNDED all the same, args.sk_spin sleeps (~5 s by default): a taker that gives up says so in
      // the library's host-visible error word -- carca_poll_errors and the next launch fail loudly (launch_gemm_rows_sk)
      // instead of the GPU hanging.  tests/test_hip_stream_k.py runs that path with a withheld flag and a short bound)
      unsigned spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins >= args.sk_spin) {
          if (args.sk_err) __hip_atomic_store(args.sk_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
      __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for the next launch)
      // ONE relaxed poll -> ONE agent-scope acquire -> its completion -> the workgroup's barrier -> plain loads: the
      // acquire (buffer_inv sc1) drops whatever this CU's L1 holds of the partial's lines -- the previous launch or the
      // previous replay of a hipGraph read the same addresses -- and the giver's sc1 stores reached memory past its own
      // XCD's L2 before the flag did.  ~1.7 us, once per taker, under the 12-wave epilogue that follows.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tn][r] += part[(tn * 16 + r) * NT + tid];
    if constexpr (XC > 0) {
#pragma unroll
      for (int c = 0; c < XC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) xacc[c][e] += part[(TN * 16 + c * 4 + e) * NT + tid];
    }
  }

  // ---- epilogue (as gemm_rows_cu_kernel) ------------------------------------------------------------
  int erow[16];  // the output row of accumulator register r (-1: past the segment's rows)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    erow[r] = row < nrows ? (IDX ? dyn.rows[row] : row) : -1;
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + tn * 32 + lr;
    if (n >= D.ncols_out) continue;
    const bool n_ok = n < D.N;
    const float bias = (n_ok && D.bias) ? D.bias[n] : 0.f;
    const float cv = (n_ok && D.colvec) ? D.colvec[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = erow[r];
      if (row < 0) continue;
      float v = 0.f;
      if (n_ok) {
        v = (D.alpha != 0.f ? D.alpha * acc[tn][r] : acc[tn][r]) + bias;
        if constexpr (!IDX) {  // (gemm_rows_skc_kernel is admitted for the plain epilogue only: alpha, bias, kept rows)
          if (sg.add_pos) v += D.pos[(size_t)(row % sg.T) * D.N + n];
          if (sg.add) v += sg.add[(size_t)row * D.ld_add + n];
          if (sg.rowscale) v += sg.rowscale[row] * cv;
          if (sg.gate) {
            const float gv = sg.gate[(size_t)row * D.ld_gate + n];
            const float gs = D.gate_scale != 0.f ? D.gate_scale : 1.0f;
            v *= gv > 0.f ? gs : ((gv < 0.f || !D.gate_zero_drops) ? D.gate_slope * gs : 0.f);
          }
          if (D.mask_rows) v = sg.ids[row] != 0 ? v : 0.f;
        }
      }
      sg.c[(size_t)row * D.ldc + n] = v;
    }
  }
  if constexpr (XC > 0) {
    // the VALU columns: lane (lr, lh) holds row wave * 32 + lr's sum over the k quads of its half; the launcher admits
    // this kernel only for the plain epilogue (alpha, bias, row mask)
    int row = row0 + wave * 32 + lr;
    const bool row_ok = row < nrows;
    if constexpr (IDX) row = dyn.rows[min(row, nrows - 1)];
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const float mine = (xacc[c][0] + xacc[c][1]) + (xacc[c][2] + xacc[c][3]);
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
      const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      const int n = n0 + 32 * TN + c;
      if (lh == 0 && row_ok && n < D.N) {
        float v = (D.alpha != 0.f ? D.alpha * tot : tot) + (D.bias ? D.bias[n] : 0.f);
        if (!IDX && D.mask_rows) v = sg.ids[row] != 0 ? v : 0.f;
        sg.c[(size_t)row * D.ldc + n] = v;
      }
    }
  }
}

template <int XC>
__global__ __launch_bounds__(768) void gemm_rows_sk_kernel(const GemmDev args) {
  __shared__ __attribute__((aligned(16))) float As[2 * 384 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[2 * 96 * 36 + 1024];
  const CarcaGemmDesc& D = args.d;
  const int id = blockIdx.x, total = args.nrb * args.ncb;
  if (args.has_pas && id == total) {
    static_assert(12 * 24 * 96 <= 2 * 384 * 36 && 12 * 18 * 128 <= 2 * 384 * 36, "the passenger's rows fit the A buffers");
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (args.pas.d <= 96)
      gather_rows_dma<24, 96>(args.pas, As, wave, 12, (int)threadIdx.x & 63);
    else
      gather_rows_dma<18, 128>(args.pas, As, wave, 12, (int)threadIdx.x & 63);
    return;
  }
  const int xcd = id & 7, q8 = total >> 3, r8 = total & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int rb = wg / args.ncb, j = wg - rb * args.ncb;
  const int nfast = D.K0 / 32, nfull = args.ncb - 1;
  if (j == 0) {  // the row block's cheap workgroup (first of the block's five to be dispatched)
    // Its own tile FIRST, in step with the four owners: the five workgroups of a row block stream the same A rows through
    // their XCD's L2.  (Partials first -- the first version -- put this workgroup ~30 K steps behind the others and it
    // fetched the whole A row block from HBM again: PMC 482 -> 823 MB per launch.)  The owners then find their partial
    // between 0.79 and 0.95 of the kernel; the last one is what the launcher's choice of `don` leaves a margin for.
    cu_tile<2, XC, SK_PLAIN>(args, As, Bs, rb, nfull * 96, 0, nfast, true, nullptr, nullptr);
    for (int cb = 0; cb < nfull; ++cb)
      cu_tile<3, 0, SK_GIVE>(args, As, Bs, rb, cb * 96, nfast - args.sk_don, nfast, false,
                             args.sk_part + ((size_t)rb * nfull + cb) * (384 * 96), args.sk_flag + rb * nfull + cb);
  } else {
    const int cb = j - 1;
    cu_tile<3, 0, SK_TAKE>(args, As, Bs, rb, cb * 96, 0, nfast - args.sk_don, true,
                           args.sk_part + ((size_t)rb * nfull + cb) * (384 * 96), args.sk_flag + rb * nfull + cb);
  }
}

// ---------------------------------------------------------------------------------------------------
// gemm_rows_skc_kernel: the feature product over the rows that COUNT.  AllEmbedding multiplies every slot of the padded id
// matrices (carca.py:86) and the padding's rows are zeroed two lines later (carca.py:92-94: "* mask"); with BASELINE's
// profile lengths U{3..50} that is 47 % of the profile rows, 16 % of an evaluation batch's rows and 47 % of a training
// batch's.  Here the row blocks are 384 KEPT rows (id != 0), and they are shared by a FIXED grid of one workgroup per CU:
//  * every workgroup counts the kept rows itself (the ids are 77 KB at C2: one pass of ballots over them, from L2 after
//    the first workgroup, ~3 us -- a pre-kernel that wrote the plan cost a launch and ~20 us of a single block's latency
//    chain in front of the product) and lists the rows of ITS row blocks in LDS;
//  * TEAMS of ncb workgroups: workgroup c of team j multiplies column block c (the narrow last one as two MFMA column
//    tiles + VALU columns, as in gemm_rows_sk_kernel) over the stretch [j, j + 1) * total / nteams of the row blocks' K
//    steps laid end to end -- a team's workgroups stream the same A rows at the same time through their XCD's L2 (a first
//    version dealt the TILES' steps out workgroup by workgroup: every tile fetched its own A rows, 3 TB/s of 128-byte
//    pieces of 16 KB-strided rows, and the kernel turned memory-bound: 501 us against 480);
//  * a stretch starts in the middle of a row block (its first piece: a partial tile handed to the block's owner team,
//    computed FIRST), runs over whole row blocks, and ends in the head of one (the team owns it: computed LAST, the
//    partial tiles of the teams behind it added).  Givers never wait, every workgroup is resident: no cycle; a taker
//    finds partials written most of a kernel ago.
// Left-out rows of the output are cleared (each workgroup its share).  Results: the same sums as gemm_rows_sk_kernel in
// another grouping of the K range (1e-7 relative), the same bits from run to run.
constexpr int SKC_RB = 8;       // row blocks one workgroup's stretch may touch (the launcher checks rows against it)
constexpr int SKC_CH = 1536;    // 64-row chunks of all segments (+ one entry per segment)
constexpr int SKC_LDS_IDS = 2 * 384 * 36;  // ids the prologue keeps in LDS (the A buffers' space: 27,648)
constexpr int SKC_MIN_STEPS = 64;  // K steps of k-source 0 below which the product stays on the one-tile-per-workgroup kernels
constexpr int SKC_OV_MAX = 12;  // bound on the row-block ownership cost in K steps (a stretch holds at least 16 steps)
// defaults (tuning keys 17 / 18 = value + 1).  MEASURED, round 5, interleaved A/B at C2 (tools/ab_eval.py): ownership cost 0 / 4
// / 8 steps -> feature GEMM 468.1 / 470.4 / 475.6 us: although the workgroups that own a row block END ~20 us after those
// that own none (tools/stamp_skc.py), handing them fewer steps makes the launch longer -- the late finishers are late for
// what the stamps do not show (their taken partials' loads contend with the givers' last stores), not for their step count.
// The correction stays as a switch; the stretches stay equal in steps.
constexpr int SKC_OV_TEAM = 0, SKC_OV_LONE = 0;

template <int XC>
__global__ __launch_bounds__(768) void gemm_rows_skc_kernel(const GemmDev args) {
  __shared__ __attribute__((aligned(16))) float As[2 * 384 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[2 * 96 * 36 + 1024];
  __shared__ int Rl[SKC_RB * 384];  // kept rows of this workgroup's row blocks
  __shared__ int Cp[SKC_CH];        // kept rows before each 64-row chunk of its segment (+ the segment's total)
  __shared__ int Pc[8 * SKC_RB + 4];  // this workgroup's pieces (step 5)
  carca_warm_kernargs<sizeof(GemmDev)>();
  if (args.dbg && (threadIdx.x & 63) == 0) args.dbg[65536 + 4096 + blockIdx.x * 16 + (threadIdx.x >> 6)] = wall_clock64();
  const CarcaGemmDesc& D = args.d;
  const int id = blockIdx.x, nblk = gridDim.x - args.has_pas;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (args.has_pas && id == nblk) {
    if (args.dbg && tid == 0) args.dbg[65536 + id * 16] = wall_clock64();
    if (args.pas.d <= 96)
      gather_rows_dma<24, 96>(args.pas, As, wave, 12, lane);
    else
      gather_rows_dma<18, 128>(args.pas, As, wave, 12, lane);
    __syncthreads();
    if (args.dbg && tid == 0) args.dbg[65536 + id * 16 + 1] = wall_clock64();
    return;
  }
  // (tools/stamp_skc.py: wall-clock stamps of workgroup `id`, 100 MHz, behind the other kernels' 64 k entries -- start, after
  // steps 1 / 2 / 4, around each piece)
#define SKC_STAMP(k) do { if (args.dbg && tid == 0) args.dbg[65536 + id * 16 + (k)] = wall_clock64(); } while (0)
  SKC_STAMP(0);
  // This workgroup's share of the left-out rows (global rows id, id + nblk, ...): a lane looks at one row's id, the wave
  // then clears the rows it found one by one.  Run LAST (after the pieces; at once by the workgroups without a stretch):
  // nothing in this launch reads those rows, and in the prologue the barrier behind it waited for the stores (~4 us).
  // (Reads the segments from the kernel arguments again: nothing of it stays alive across the K loops.)
  auto clear_left_out = [&]() {
    int ro[CARCA_MAX_SEGS + 1];
    ro[0] = 0;
#pragma unroll
    for (int i = 0; i < CARCA_MAX_SEGS; ++i) ro[i + 1] = ro[i] + D.seg[i].rows;
    const int R = ro[CARCA_MAX_SEGS];
#pragma unroll 1
    for (int k0 = wave * 64; id + (long)k0 * nblk < R; k0 += 12 * 64) {
      const long g = id + (long)(k0 + lane) * nblk;
      int sg = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (D.seg[i].rows > 0 && g >= ro[i]) sg = i;
      const int pv = D.seg[sg].ids[min(g, (long)R - 1) - ro[sg]];
      const bool pad = g < R && pv == 0;
      unsigned long long bal = __ballot(pad);
      while (bal) {
        const int b = __ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const long gz = id + (long)(k0 + b) * nblk;
        int sz = 0;
#pragma unroll
        for (int i = 1; i < CARCA_MAX_SEGS; ++i)
          if (D.seg[i].rows > 0 && gz >= ro[i]) sz = i;
        float* crow = D.seg[sz].c + (size_t)(gz - ro[sz]) * D.ldc;
        for (int cc = lane; cc < D.ncols_out; cc += 64) crow[cc] = 0.f;
      }
    }
  };
  // ---- 1. kept rows per 64-row chunk ----------------------------------------------------------------------------
  // (first chunk entry / first global row of segment s: running sums over at most four segments, recomputed where needed --
  // arrays indexed by a run-time s would live in scratch)
  // (the segments' row counts and id pointers read UNCONDITIONALLY, all at once -- unused segments are zero-filled by the
  // launcher: under `i < nseg` every one of them was a dependent scalar load, ~4 us of round trips when the kernel starts)
  int nrows_[CARCA_MAX_SEGS];
  const int32_t* ids_[CARCA_MAX_SEGS];
#pragma unroll
  for (int i = 0; i < CARCA_MAX_SEGS; ++i) {
    nrows_[i] = D.seg[i].rows;
    ids_[i] = D.seg[i].ids;
  }
  auto seg_cbase = [&](int s) {  // (s = 0 .. CARCA_MAX_SEGS: the entries in front of segment s, all of them for s = nseg)
    int v = 0;
#pragma unroll
    for (int i = 0; i < CARCA_MAX_SEGS; ++i) v += i < s ? (nrows_[i] + 63) / 64 + 1 : 0;
    return v;
  };
  auto seg_roff = [&](int s) {
    int v = 0;
#pragma unroll
    for (int i = 0; i < CARCA_MAX_SEGS; ++i) v += i < s ? nrows_[i] : 0;
    return v;
  };
  const unsigned long long below = (1ull << lane) - 1;
  // All segments' chunks as one run 0 .. G - 1; 32 chunks' ids per wave requested together (four groups of eight dealt
  // round the waves): one chunk at a time is a dependent load per iteration (~19 us at C2), one segment at a time a round
  // trip per segment.  The loop body stays small: unrolled over segments with the clearing inside, it was 100 KB of code
  // run once.
  const int G = seg_cbase(D.nseg) - D.nseg;
  // (every load UNCONDITIONAL, from a clamped row of a segment picked by compares: a load under a branch gets its own wait
  // at the join -- the first version ran its 32 loads one after the other, 12 us)
  int seg_rows[CARCA_MAX_SEGS], seg_g0[CARCA_MAX_SEGS];
  const int32_t* seg_ids[CARCA_MAX_SEGS];
  {
    int gb = 0;
#pragma unroll
    for (int i = 0; i < CARCA_MAX_SEGS; ++i) {
      const bool on = nrows_[i] > 0;
      seg_rows[i] = on ? nrows_[i] : 1;
      seg_ids[i] = on ? ids_[i] : ids_[0];
      seg_g0[i] = on ? gb : 0x7fffffff;
      gb += (nrows_[i] + 63) / 64;
    }
  }
  // Up to SKC_LDS_IDS ids go to LDS by LDS-DMA, from two LOOPS of a few instructions each: the prologue runs once per
  // workgroup, so every line of its code is an instruction-cache miss -- the unrolled version below (32 loads' address
  // arithmetic, 32 ballots: ~10 KB of straight-line code) spent 8.4 us in front of its first barrier whether its loads read
  // anything or not (tuning key 15 bit 0), and the waves of a workgroup start within 0.05 us of each other.
  int* const Ai = reinterpret_cast<int*>(As);
  const bool ids_in_lds = G * 64 <= SKC_LDS_IDS;
  SKC_STAMP(4096 + 12);
  if (ids_in_lds) {
    // (segment by segment: picking the segment of every chunk with compares made each iteration a chain of ~20 dependent
    // scalar instructions, 150 cycles -- 25 iterations of it were the larger part of the 8.4 us)
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll 1
    for (int sgi = 0; sgi < D.nseg; ++sgi) {
      int rows = nrows_[0], gb = 0;
      const int32_t* ip = ids_[0];
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (i <= sgi) {
          gb += (nrows_[i - 1] + 63) / 64;
          if (i == sgi) {
            rows = nrows_[i];
            ip = ids_[i];
          }
        }
      const __amdgpu_buffer_rsrc_t rs = carca_rsrc(ip);
      const int nch = (rows + 63) / 64;
#pragma unroll 4
      for (int c = wave; c < nch; c += 12)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(Ai + (gb + c) * 64), 4, min(c * 64 + lane, rows - 1) * 4, 0, 0, 0);
    }
    SKC_STAMP(4096 + 13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SKC_STAMP(4096 + 14);
#pragma unroll 1
    for (int sgi = 0; sgi < D.nseg; ++sgi) {
      int rows = nrows_[0], gb = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (i <= sgi) {
          gb += (nrows_[i - 1] + 63) / 64;
          if (i == sgi) rows = nrows_[i];
        }
      const int nch = (rows + 63) / 64;
#pragma unroll 4
      for (int c = wave; c < nch; c += 12) {
        const bool keep = c * 64 + lane < rows && Ai[(gb + c) * 64 + lane] != 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) Cp[gb + c + sgi] = __popcll(bal);
      }
    }
    SKC_STAMP(4096 + 15);
  }
#pragma unroll 1
  for (int g0 = ids_in_lds ? G : 0; g0 < G; g0 += 12 * 32) {
    int idv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int g = g0 + ((u >> 3) * 12 + wave) * 8 + (u & 7);  // (wave-uniform)
      int rows = seg_rows[0], gb = 0;
      const int32_t* ip = seg_ids[0];
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (g >= seg_g0[i]) {
          rows = seg_rows[i];
          gb = seg_g0[i];
          ip = seg_ids[i];
        }
      // (a buffer load bounded by the segment's rows: rows past its end -- and chunks past the last one -- read as id 0;
      // scalar base + one lane offset instead of a 64-bit address per lane)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(ip), 0, (g < G && !(args.diag & 1)) ? rows * 4 : 0, 0x00020000);
      idv[u] = (int)__builtin_amdgcn_raw_buffer_load_b32(rs, ((g - gb) * 64 + lane) * 4, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int g = g0 + ((u >> 3) * 12 + wave) * 8 + (u & 7);
      int sgi = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (g >= seg_g0[i]) sgi = i;
      const unsigned long long bal = __ballot(idv[u] != 0);
      if (lane == 0 && g < G) Cp[g + sgi] = __popcll(bal);  // (entry of chunk g: one spare entry behind each segment)
    }
  }
  __syncthreads();
  SKC_STAMP(1);
  // ---- 2. wave 0: the chunks' counts become running sums per segment -------------------------------------------
  if (wave == 0) {
#pragma unroll 1
    for (int s = 0; s < D.nseg; ++s) {
      const int nch = (D.seg[s].rows + 63) / 64, per = (nch + 63) / 64, cb0 = seg_cbase(s);
      int mine = 0;
      for (int i = 0; i < per; ++i) {
        const int c = lane * per + i;
        mine += c < nch ? Cp[cb0 + c] : 0;
      }
      int incl = mine;  // inclusive sum over the lanes
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) {
        const int o = __shfl_up(incl, sh);
        if (lane >= sh) incl += o;
      }
      int run = incl - mine;
      for (int i = 0; i < per; ++i) {
        const int c = lane * per + i;
        if (c < nch) {
          const int v = Cp[cb0 + c];
          Cp[cb0 + c] = run;
          run += v;
        }
      }
      if (lane == 63) Cp[cb0 + nch] = incl;  // the segment's kept rows
    }
  }
  __syncthreads();
  SKC_STAMP(2);
  // ---- 3. the plan, the same in every workgroup ----------------------------------------------------------------
  int live[CARCA_MAX_SEGS], rbs[CARCA_MAX_SEGS + 1];
  rbs[0] = 0;
#pragma unroll
  for (int s = 0; s < CARCA_MAX_SEGS; ++s) {
    live[s] = s < D.nseg ? Cp[seg_cbase(s) + (D.seg[s].rows + 63) / 64] : 0;
    rbs[s + 1] = rbs[s] + (live[s] + 383) / 384;
  }
  const int nrb = rbs[CARCA_MAX_SEGS], ncb = args.ncb;
  const int nfast = D.K0 / 32, nfull = ncb - 1;
  const long total = (long)nfast * nrb;  // K steps of all row blocks
  // Two kinds of workgroups: x TEAMS of nfull (column blocks 0 .. nfull - 1 of the team's stretch: they stream the same A
  // rows together) and y LONE ones for the narrow last column block, whose K step costs skc_cheap / 100 of a full one --
  // x nfull + y = the grid, both kinds done at the same time: x = grid / (nfull + cheap); at least 16 K steps per stretch.
  const long cap = max(1l, total / 16);
  int x = (int)max(1l, min(cap, (long)nblk * 100 / (nfull * 100 + args.skc_cheap)));
  {
    auto span = [&](int xx) {  // time of the slower kind with xx teams (in 1/100 K steps)
      const long yy = max(1l, min(cap, (long)nblk - (long)xx * nfull));
      return max(total * 100 / xx, total * args.skc_cheap / yy);
    };
    if ((x + 1) * nfull < nblk && x + 1 <= cap && span(x + 1) < span(x)) ++x;
  }
  const int y = (int)max(1l, min(cap, (long)nblk - (long)x * nfull));
  const int xcd = id & 7, nw = x * nfull + y, q8 = nw >> 3, r8 = nw & 7;
  const int cnt = xcd < r8 ? q8 + 1 : q8;
  if (nrb == 0 || (id >> 3) >= cnt) {
    clear_left_out();
    return;
  }
  const int w = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const bool lone = w >= x * nfull;
  const int nteams = lone ? y : x;                       // stretches of this workgroup's kind
  const int tj = lone ? w - x * nfull : w / nfull;       // its stretch
  const int cb = lone ? nfull : w - tj * nfull;          // its column block
  // The stretches are equal in COST, not in steps: the workgroup that owns a row block (the one whose stretch holds the
  // block's first step) also runs the block's ctx tail tile, adds the partial tiles of the stretches behind it and writes
  // the block out -- `ov` K steps' worth.  Every row block is therefore laid out `ov` steps longer on a virtual axis (the
  // overhead in front of its first step), the axis is cut into equal stretches and mapped back: a cut inside an
  // overhead zone falls on the block's boundary.  (Cut equal in steps, the workgroups that own nothing -- a stretch inside
  // one block: 11 of C2's 54 teams -- ended ~15 us before the owners: tools/stamp_skc.py.)
  const int ov = lone ? args.skc_ov_lone : args.skc_ov;
  const long vw = nfast + ov, vtotal = vw * nrb;
  auto cut = [&](long j) {  // first step of stretch j (j = nteams: one past the last step)
    const long v = vtotal * j / nteams, b = v / vw, r = v - b * vw;
    return b * nfast + max(0l, r - (long)ov);
  };
  const long lo = cut(tj), hi = cut(tj + 1);
  const int rbA = (int)(lo / nfast), sA = (int)(lo - (long)rbA * nfast);
  int rbB = (int)(hi / nfast), sB = (int)(hi - (long)rbB * nfast);
  if (sB == 0) {  // (the stretch ends on a row block's boundary)
    --rbB;
    sB = nfast;
  }
  // ---- 4. the kept rows of row blocks rbA .. rbB, in row order ---------------------------------------------------
  // (the row blocks go round the waves: wave % npc takes block rbA + wave % npc with the waves that share it)
  const int npc = min(rbB - rbA + 1, SKC_RB);
  {
    const int pi = wave % npc, sub = wave / npc, nsub = (12 - pi + npc - 1) / npc;
    const int rb = rbA + pi;
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && rb >= rbs[i]) s = i;
    const CarcaGemmSeg sg = D.seg[s];
    const int p0 = (rb - rbs[s]) * 384, p1 = min(p0 + 384, live[s]);
    const int nch = (sg.rows + 63) / 64, cb0 = seg_cbase(s);
    // the chunks that hold the block's rows: a contiguous run (the running sums are monotone); every wave finds its ends
    // by itself (two searches over <= 2304 sums in LDS), then the run's chunks go round the waves, four requested together
    int cl = 0, ch = nch;  // first chunk whose end is past p0; first chunk that starts at or after p1
    {
      int l = 0, h = nch;
      while (l < h) {
        const int m = (l + h) >> 1;
        if (Cp[cb0 + m + 1] <= p0) l = m + 1; else h = m;
      }
      cl = l;
      l = cl;
      h = nch;
      while (l < h) {
        const int m = (l + h) >> 1;
        if (Cp[cb0 + m] < p1) l = m + 1; else h = m;
      }
      ch = l;
    }
    if (ids_in_lds) {
      const int gbase = cb0 - s;  // (chunk c of the segment is chunk gbase + c of all segments: one Cp entry fewer per segment)
#pragma unroll 1
      for (int c = cl + sub; c < ch; c += nsub) {
        const bool keep = c * 64 + lane < sg.rows && Ai[(gbase + c) * 64 + lane] != 0;
        const unsigned long long bal = __ballot(keep);
        const int pos = Cp[cb0 + c] + __popcll(bal & below);
        if (keep && pos >= p0 && pos < p1) Rl[(rb - rbA) * 384 + pos - p0] = c * 64 + lane;
      }
    }
    for (int c0 = ids_in_lds ? ch : cl + sub * 4; c0 < ch; c0 += nsub * 4) {
      int idv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (c0 + u) * 64 + lane;
        const int v = sg.ids[min(row, sg.rows - 1)];
        idv[u] = (c0 + u < ch && row < sg.rows) ? v : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u;
        const bool keep = idv[u] != 0;
        const unsigned long long bal = __ballot(keep);
        if (c < ch) {
          const int pos = Cp[cb0 + c] + __popcll(bal & below);
          if (keep && pos >= p0 && pos < p1) Rl[(rb - rbA) * 384 + pos - p0] = c * 64 + lane;
        }
      }
    }
  }
  __syncthreads();
  SKC_STAMP(3);
  // ---- 5. the pieces, written down: the K loops below keep none of the plan's scalars alive (with them in registers the
  // compiler re-read the buffer descriptors from the kernel arguments INSIDE the K loop, a scalar load whose wait also
  // waits for the step's LDS reads: +5 % per step) -----------------------------------------------------------------
  if (tid == 0) {
    // partial tile / flag of this workgroup's first piece (consecutive over the stretches of a column block)
    const int slot = (lone ? nfull * x : cb * x) + tj;
    if (rbB - rbA + 1 > SKC_RB && args.sk_err)  // (the launcher's bound on rows rules this out)
      __hip_atomic_store(args.sk_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int np = 0;
    for (int rb = rbA; rb <= rbB && np < SKC_RB; ++rb, ++np) {
      const int s0 = rb == rbA ? sA : 0, s1 = rb == rbB ? sB : nfast;
      int sg = 0;
#pragma unroll
      for (int i = 1; i < CARCA_MAX_SEGS; ++i)
        if (i < D.nseg && rb >= rbs[i]) sg = i;
      int mode = s0 > 0 ? SK_GIVE : (s1 < nfast ? SK_TAKE : SK_PLAIN), ntake = 0;
      if (mode == SK_TAKE) {
        // the stretches behind this one that start inside the row block: tj + 1 .. the last j' with lo(j') < end (this
        // stretch ends inside the block, so every later start is past the block's first step)
        const long end = (long)(rb + 1) * nfast;
        for (int j2 = tj + 1; j2 < nteams && cut(j2) < end; ++j2) ++ntake;
        if (ntake <= 0) mode = SK_PLAIN;  // (cannot happen with s1 < nfast; a taker must never wait for nobody)
      }
      int* pc = Pc + 8 * np;
      pc[0] = rb;
      pc[1] = s0;
      pc[2] = s1;
      pc[3] = sg;
      pc[4] = min(384, live[sg] - (rb - rbs[sg]) * 384);
      pc[5] = mode;
      pc[6] = ntake;
      pc[7] = mode == SK_GIVE ? slot : slot + 1;
    }
    Pc[8 * SKC_RB] = np;
    Pc[8 * SKC_RB + 1] = cb;
    Pc[8 * SKC_RB + 2] = (int)lone;
  }
  __syncthreads();
  // ---- 6. the pieces, read back -------------------------------------------------------------------------------------
  const int np = __builtin_amdgcn_readfirstlane(Pc[8 * SKC_RB]);
  const int mycb = __builtin_amdgcn_readfirstlane(Pc[8 * SKC_RB + 1]);
  const int nfull_ = args.ncb - 1;
#pragma unroll 1
  for (int p = 0; p < np; ++p) {
    const int* pc = Pc + 8 * p;
    const int rb = __builtin_amdgcn_readfirstlane(pc[0]), s0 = __builtin_amdgcn_readfirstlane(pc[1]);
    const int s1 = __builtin_amdgcn_readfirstlane(pc[2]);
    SkcTile dyn;
    dyn.seg = __builtin_amdgcn_readfirstlane(pc[3]);
    dyn.row0 = 0;
    dyn.live = __builtin_amdgcn_readfirstlane(pc[4]);
    dyn.rows = Rl + p * 384;
    dyn.mode = __builtin_amdgcn_readfirstlane(pc[5]);
    dyn.ntake = __builtin_amdgcn_readfirstlane(pc[6]);
    const int slot = __builtin_amdgcn_readfirstlane(pc[7]);
    SKC_STAMP(4 + 2 * p);
    if (args.dbg && tid == 0 && p < 3)
      args.dbg[65536 + id * 16 + 10 + p] = (unsigned long long)(s1 - s0) * 8 + (Pc[8 * SKC_RB + 2] ? 4 : 0) + dyn.mode;
    const bool give = dyn.mode == SK_GIVE;
    float* part = args.sk_part + (size_t)slot * (384 * 96);
    int* flag = args.sk_flag + slot;
    if (mycb < nfull_)
      cu_tile<3, 0, SK_DYN>(args, As, Bs, rb, mycb * 96, s0, s1, !give, part, flag, dyn);
    else
      cu_tile<2, XC, SK_DYN>(args, As, Bs, rb, nfull_ * 96, s0, s1, !give, part, flag, dyn);
    SKC_STAMP(5 + 2 * p);
  }
  clear_left_out();
#undef SKC_STAMP
}

// ---------------------------------------------------------------------------------------------------
// gemm_rows_n96_kernel: the same product for a NARROW output (64 < N <= 96: the joint embedding e = [z ; q] W_j^T,
// carca.py:89) over one k-source with a short K (540 at C2).  The 128 x 32 blocks above leave one dependent chain per
// K step on eight waves a CU and two unequal rounds of blocks; here ONE 768-thread block per CU takes 80 rows x all 96
// columns (19328 rows = 242 blocks on 256 CUs, one round): 5 x 6 16x16 tiles, wave (ct = wave % 6, kh = wave / 6) owns
// column tile ct over the five row tiles for one half of every 64-wide K stage -- 2 x 20 MFMAs (16x16x4) in five
// independent chains behind twelve 16-byte fragment reads, three waves per SIMD -- and the two K halves meet in LDS at
// the end.  Stages are double-buffered in LDS (one barrier per stage) behind a two-stage register ring of buffer loads;
// what the epilogue reads per row (ids, positional rows, bias) is requested before the K loop.
// Per CU the product needs 80 x 540 of A (173 KB) and all of W_j (207 KB) for 4050 MFMAs = 32.4 k cycles per SIMD =
// 13.5 us; the launch itself (prologue, first round trip, epilogue without its stores) measures 7.8 us in the pipeline.
namespace n96 {
constexpr int BM = 80, BN = 96, BK = 64, LS = BK + 4, NT = 768, C4 = BK / 4;
constexpr int A_SLOTS = BM * C4, B_SLOTS = BN * C4;        // 1280 + 1536 sixteen-byte slots per stage
constexpr int NSL = (A_SLOTS + B_SLOTS + NT - 1) / NT;    // 4 per thread (the last one for the first 512 threads)
constexpr int A_BUF = BM * LS, B_BUF = BN * LS;
static_assert(A_SLOTS % 64 == 0 && (A_SLOTS + B_SLOTS) % 64 == 0, "a wave's slots are all A, all B or all void");
}  // namespace n96

__global__ __launch_bounds__(768) void gemm_rows_n96_kernel(const GemmDev args) {
  using namespace n96;
  extern __shared__ __attribute__((aligned(16))) float Sm[];  // 2 x (A stage + B stage)
  float* As = Sm;
  float* Bs = Sm + 2 * A_BUF;
  static_assert(2 * (A_BUF + B_BUF) >= BM * (BN + 4), "the K-half partial sums reuse the stage buffers");
  const CarcaGemmDesc& D = args.d;
  const int rb = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && rb >= args.rb_start[i]) s = i;
  const CarcaGemmSeg sg = D.seg[s];
  const int row0 = (rb - args.rb_start[s]) * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform to the compiler too: scalar branches, scalar resource choice)
  const int ln = lane & 15, mq = lane >> 4;
  const int ct = wave % 6, kh = wave / 6;
  const int nst = (D.K0 + BK - 1) / BK;
  const int diag = args.diag;

  // staging slots tid + 768 i: A slots first, then B slots; a slot past the end repeats the thread's previous one (same
  // load, same store: harmless) so that the pipelined body has no branch -- behind a branch with loads inside, hipcc
  // turns every counted wait into vmcnt(0)
  unsigned off[NSL];   // byte offset of the slot's row start
  int c4s[NSL], ldsx[NSL], bstr[NSL];
  bool slot_a[NSL];
#pragma unroll
  for (int i = 0; i < NSL; ++i) {
    int slot = tid + i * NT;
    if (slot >= A_SLOTS + B_SLOTS) slot -= NT;
    const bool is_a = slot < A_SLOTS;
    slot_a[i] = (i * NT + (wave << 6) >= A_SLOTS + B_SLOTS ? (i - 1) * NT + (wave << 6) : i * NT + (wave << 6)) < A_SLOTS;  // (uniform)
    const int bs = slot - A_SLOTS;
    const int r = is_a ? slot / C4 : bs / C4;
    c4s[i] = (is_a ? slot : bs) % C4;
    const int gr = min(row0 + r, sg.rows - 1);
    const int ub = gr / sg.T, ut = gr - ub * sg.T;
    const size_t ao = sg.a0_bstride ? (size_t)ub * sg.a0_bstride + (size_t)ut * D.lda0 : (size_t)gr * D.lda0;
    off[i] = is_a ? (unsigned)(ao * sizeof(float)) : (unsigned)((size_t)min(r, D.N - 1) * D.ldb0 * sizeof(float));
    ldsx[i] = (is_a ? 0 : 2 * A_BUF) + r * LS + c4s[i] * 4;  // float index inside Sm for buffer 0
    bstr[i] = is_a ? A_BUF : B_BUF;
  }
  const __amdgpu_buffer_rsrc_t r_a = carca_rsrc(sg.a0), r_b = carca_rsrc(D.bt0);
  f32x4 rg[3][NSL];  // three stages of loads in flight (the fabric streams while the MFMAs run)
  auto load_stage = [&](int st, int ring) {
    // (the last stage of a K that is no multiple of 64: 16-byte groups past K are clamped to the last valid one and
    // zeroed when stored; K0 % 4 == 0, so a group is all inside or all outside)
    const int kb = st * BK;
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const unsigned kc = 4u * (unsigned)min(kb + c4s[i] * 4, D.K0 - 4);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(slot_a[i] ? r_a : r_b, off[i] + kc, 0, 0);
      rg[ring][i] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
  };
  auto store_stage = [&](int st, int ring, int buf) {
    const int kb = st * BK;
    if (kb + BK <= D.K0) {  // (uniform: a full stage)
#pragma unroll
      for (int i = 0; i < NSL; ++i) *reinterpret_cast<f32x4*>(&Sm[ldsx[i] + buf * bstr[i]]) = rg[ring][i];
      return;
    }
    // the last stage of a K that is no multiple of 64 (and, for K0 % 4 != 0 -- the joint product over the g = 450 columns
    // of q --, no multiple of 4): the load of a group that starts at column kc < K0 was clamped to start at K0 - 4, so
    // its elements sit `sh` places further up; what lies past K0 is zeros in both operands
#pragma unroll
    for (int i = 0; i < NSL; ++i) {
      const int kc = kb + c4s[i] * 4, sh = kc - min(kc, D.K0 - 4);  // (0 for groups that end inside K0)
      const f32x4 r = rg[ring][i];
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = sh == 0 ? r[e] : (sh == 1 ? r[(e + 1) & 3] : (sh == 2 ? r[(e + 2) & 3] : r[(e + 3) & 3]));
        v[e] = (kc + e < D.K0) ? x : 0.f;
      }
      *reinterpret_cast<f32x4*>(&Sm[ldsx[i] + buf * bstr[i]]) = v;
    }
  };

  // ---- the epilogue's per-row operands (ids, positional rows, bias) are requested behind the K loop, all at once, as
  // unconditional buffer loads (a load under a branch, or a select on a loaded value, is waited for right there, one row
  // after the other); where a pointer is absent the weights stand in and the value is dropped.
  // acc[rt][r] = C[row0 + 16 rt + 4 mq + r][16 ct + ln]
  const int n = 16 * ct + ln;
  const bool n_ok = n < D.N;
  const bool fancy = !(diag & 4);
  const bool use_pos = sg.add_pos && D.pos;

  const bool use_tab = D.add_table != nullptr && sg.ids != nullptr;
  int idv0[5][4];  // with add_table: the ids of the lane's twenty rows, requested now (the table rows need them after the loop)
#pragma unroll
  for (int rt = 0; rt < 5; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      idv0[rt][r] = use_tab ? gload1i(sg.ids, min(row0 + 16 * rt + 4 * mq + r, sg.rows - 1)) : 0;
  f32x4 acc[5];
#pragma unroll
  for (int rt = 0; rt < 5; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragments: lane (ln, mq) reads 16 bytes = k 4 mq .. + 3 of a 16-wide chunk of its row; this wave's chunks of a stage
  // are 2 kh and 2 kh + 1
  const float* a_frag = &As[ln * LS + 32 * kh + 4 * mq];
  const float* b_frag = &Bs[(16 * ct + ln) * LS + 32 * kh + 4 * mq];
  auto compute = [&](int buf, int kleft) {  // kleft: columns of K0 from this stage's first one on (>= 64: a full stage)
    // (reading all twelve fragments of the stage ahead of its 40 MFMAs measured 1 us SLOWER than chunk by chunk)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      // (the ragged last stage: a 16-wide chunk that starts past K0 is all zeros in both operands -- K0 = 450 ends two
      // columns into its eighth stage, K0 = 540 twenty-eight into its ninth: the wave-uniform test skips 20 MFMAs a chunk)
      if (32 * kh + 16 * ch >= kleft) continue;
      const f32x4 b = *reinterpret_cast<const f32x4*>(b_frag + buf * B_BUF + 16 * ch);
      f32x4 a[5];
#pragma unroll
      for (int rt = 0; rt < 5; ++rt) a[rt] = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + rt * 16 * LS + 16 * ch);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) acc[rt] = mfma16(a[rt][e], b[e], acc[rt]);
    }
  };

  load_stage(0, 0);
  if (nst > 1) load_stage(1, 1);
  if (nst > 2) load_stage(2, 2);
  store_stage(0, 0, 0);
  __syncthreads();
  // ---- six HAND-SCHEDULED steps in front (round 5; the schedule of gemm_rows_n96s_kernel's step, gemm_stream.hip): a full
  // stage is two groups of twenty pinned MFMAs (this wave's two 16-k chunks, five accumulator chains); the gaps of group 0
  // carry the six fragment reads of chunk 1 and the four LDS stores of stage st + 1, ONE barrier, the gaps of group 1 the six
  // fragment reads of the next stage's chunk 0 and the four requests of stage st + 3.  Left to the compiler (N96_STEP below:
  // store, request, multiply, barrier) a stage took ~2.5 us against 1.6 us of MFMA time.  Six of them, or none: LDS buffer and
  // ring slot are compile-time constants of a step (period six), and a loop in which a pinned and a generic step are
  // alternatives costs a second copy of the accumulators; taken when stages 0..6 are full ones.
  int st0 = 0;
  if (!(diag & 32) && nst >= 8 && 7 * BK <= D.K0 && !(diag & 3)) {
#define N96_PIN() __builtin_amdgcn_sched_barrier(0)
    f32x4 fa0[5], fa1[5], fb0, fb1;
    auto read_slot = [&](int i, int buf, int ch, f32x4(&fa)[5], f32x4& fb) {
      if (i == 0)
        fb = *reinterpret_cast<const f32x4*>(b_frag + buf * B_BUF + 16 * ch);
      else
        fa[i > 0 ? i - 1 : 0] = *reinterpret_cast<const f32x4*>(a_frag + buf * A_BUF + (i > 0 ? i - 1 : 0) * 16 * LS + 16 * ch);
    };
    auto group = [&](const f32x4(&fa)[5], const f32x4& fb, auto&& aux) {
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        acc[i % 5] = mfma16(fa[i % 5][i / 5], fb[i / 5], acc[i % 5]);
        N96_PIN();
        aux(i);
        N96_PIN();
      }
    };
    auto pinned = [&](auto b0_tag, auto r1_tag, auto r3_tag, int st) {
      constexpr int B0 = decltype(b0_tag)::value, B1 = B0 ^ 1, R1 = decltype(r1_tag)::value, R3 = decltype(r3_tag)::value;
      const int kb3 = (st + 3) * BK;
      group(fa0, fb0, [&](int i) {
        if (i < 6) read_slot(i, B0, 1, fa1, fb1);
        if (i >= 6 && i < 6 + NSL) *reinterpret_cast<f32x4*>(&Sm[ldsx[i - 6 < NSL ? i - 6 : 0] + B1 * bstr[i - 6 < NSL ? i - 6 : 0]]) = rg[R1][i - 6 < NSL ? i - 6 : 0];
      });
      N96_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // raw: the requested stages stay in flight across it
      N96_PIN();
      group(fa1, fb1, [&](int i) {
        if (i < 6) read_slot(i, B1, 0, fa0, fb0);
        if (i >= 6 && i < 6 + NSL) {  // stage st + 3 (st <= 5 and nst >= 8: it exists; its clamp matters for the ragged one)
          const int j = i - 6 < NSL ? i - 6 : 0;
          const unsigned kc = 4u * (unsigned)min(kb3 + c4s[j] * 4, D.K0 - 4);
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(slot_a[j] ? r_a : r_b, off[j] + kc, 0, 0);
          rg[R3][j] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        }
      });
    };
#pragma unroll
    for (int i = 0; i < 6; ++i) read_slot(i, 0, 0, fa0, fb0);
    using std::integral_constant;
    pinned(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, 0);
    pinned(integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{}, 1);
    pinned(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 2>{}, 2);
    pinned(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, 3);
    pinned(integral_constant<int, 0>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{}, 4);
    pinned(integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 2>{}, 5);
#undef N96_PIN
    st0 = 6;
  }
  // stage st: LDS buffer st % 2, ring slot st % 3; unrolled by six so that both are compile-time constants
#define N96_STEP(o, R1, B1, R3, B0)                                        \
  if (st + (o) < nst) {                                                    \
    if (st + (o) + 1 < nst) store_stage(st + (o) + 1, R1, B1);             \
    if (st + (o) + 3 < nst && !(diag & 1)) load_stage(st + (o) + 3, R3);   \
    if (!(diag & 2)) compute(B0, D.K0 - (st + (o)) * BK);                  \
    __syncthreads();                                                       \
  }
  for (int st = (diag & 16) ? nst : st0; st < nst; st += 6) {
    N96_STEP(0, 1, 1, 0, 0)
    N96_STEP(1, 2, 0, 1, 1)
    N96_STEP(2, 0, 1, 2, 0)
    N96_STEP(3, 1, 0, 0, 1)
    N96_STEP(4, 2, 1, 1, 0)
    N96_STEP(5, 0, 0, 2, 1)
  }
#undef N96_STEP
  // ---- the two K halves meet: kh = 1 leaves its sums in LDS (the stage buffers are dead), kh = 0 adds and finishes ------
  float* Part = Sm;  // [BM][BN + 4]
  constexpr int PS = BN + 4;
  int idv[5][4];
  float posv[5][4], tabv[5][4];
  float bias;
  {
    const float* posp = use_pos ? D.pos : D.bt0;
    const int32_t* idp = sg.ids ? sg.ids : reinterpret_cast<const int32_t*>(D.bt0);
    const int nn = min(n, D.N - 1);
    bias = gload1(D.bias ? D.bias : D.bt0, nn);
    const int t0 = use_pos ? row0 % sg.T : 0;
#pragma unroll
    for (int rt = 0; rt < 5; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 16 * rt + 4 * mq + r;
        const int row = min(row0 + lrow, sg.rows - 1);
        int tt = t0 + lrow;  // (row % T without twenty divisions: t0 + 79 < 4 T for T >= 27; smaller T takes the slow form)
        if (use_pos) {
          if (sg.T >= 27) {
            tt = tt >= sg.T ? tt - sg.T : tt;
            tt = tt >= sg.T ? tt - sg.T : tt;
            tt = tt >= sg.T ? tt - sg.T : tt;
          } else {
            tt %= sg.T;
          }
        }
        idv[rt][r] = use_tab ? idv0[rt][r] : gload1i(idp, sg.ids ? row : 0);
        posv[rt][r] = gload1(posp, use_pos ? tt * D.N + nn : 0);
        // (the addend gathered by id, CarcaGemmDesc.add_table: the rows' ids were requested before the K loop)
        tabv[rt][r] = gload1(use_tab ? D.add_table : D.bt0, use_tab ? idv0[rt][r] * D.ld_add_table + nn : 0);
      }
  }
  if (kh == 1) {
#pragma unroll
    for (int rt = 0; rt < 5; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Part[(16 * rt + 4 * mq + r) * PS + 16 * ct + ln] = acc[rt][r];
  }
  __syncthreads();
  if (kh == 1) return;
  if (n >= D.ncols_out) return;
  const float cv = (n_ok && D.colvec) ? D.colvec[n] : 0.f;
#pragma unroll
  for (int rt = 0; rt < 5; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lrow = 16 * rt + 4 * mq + r, row = row0 + lrow;
      if (row >= sg.rows) continue;
      float v = 0.f;
      if (n_ok) {
        const float a = acc[rt][r] + Part[lrow * PS + n];
        v = (D.alpha != 0.f ? D.alpha * a : a) + (D.bias ? bias : 0.f);
        if (!fancy) {
          if (!(diag & 8) || v == 123.456f) sg.c[(size_t)row * D.ldc + n] = v;
          continue;
        }
        if (use_pos) v += posv[rt][r];
        if (use_tab) v += tabv[rt][r];
        if (sg.add) v += sg.add[(size_t)row * D.ld_add + n];
        if (sg.rowscale) v += sg.rowscale[row] * cv;
        if (sg.gate) {
          const float gv = sg.gate[(size_t)row * D.ld_gate + n];
          const float gs = D.gate_scale != 0.f ? D.gate_scale : 1.0f;
          v *= gv > 0.f ? gs : ((gv < 0.f || !D.gate_zero_drops) ? D.gate_slope * gs : 0.f);
        }
        if (D.mask_rows) v = idv[rt][r] != 0 ? v : 0.f;
      }
      sg.c[(size_t)row * D.ldc + n] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
struct WgradDev {
  CarcaWgradDesc d;
  int chunk_start[CARCA_MAX_SEGS + 1];  // 32-row chunks per segment, prefix sums
  int nnb, nkb, nkb0, nsplit, chunks_per_split;
  int diag_plain_store;  // diagnostic (tuning key 3): overwrite instead of atomicAdd, to time the kernel without atomics
  // Row splits WITHOUT atomics: block (split, nb, kb) stores its 96 x 128 tile plainly, in register order, at
  // part[((split * nnb + nb) * nkb + kb) * 12288 ..] and wgrad_part_reduce adds a tile's splits in order into dw.  (A/B at
  // C2 with plain stores in place of the atomics, wrong results: train step -43 us -- an fp32 atomic costs ~5 ns and the
  // thirteen d x d products + the joint-embedding dW issue 11 M of them per step.)  NULL = atomics (grad_add).
  float* part;
};

template <int BNO, int BKO, int BR, bool BUF>
__device__ __forceinline__ void wgrad_body(const WgradDev& args, int b) {
  static_assert(BNO == 96 && BKO == 128 && BR == 32, "tile shape baked into the lane maps below");
  constexpr int NT = 256;
  __shared__ __attribute__((aligned(16))) float Ys[BR * BNO];  // dY tile [row][n]
  __shared__ __attribute__((aligned(16))) float Xs[BR * BKO];  // X  tile [row][k]

  const CarcaWgradDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kb = b % args.nkb;
  b /= args.nkb;
  const int nb = b % args.nnb;
  const int split = b / args.nnb;
  const bool src1 = kb >= args.nkb0;  // this block's dW columns come from the second X source
  const int n0 = nb * BNO, k0 = (src1 ? kb - args.nkb0 : kb) * BKO;
  const int klen = src1 ? D.K1 : D.K;
  const int ldx = src1 ? D.ld_x1 : D.ld_x;
  const int c_begin = split * args.chunks_per_split;
  const int c_end = min(c_begin + args.chunks_per_split, args.chunk_start[D.nseg]);
  const bool n_full = n0 + BNO <= D.N, k_full = k0 + BKO <= klen;

  constexpr int Y4 = BNO / 4, X4 = BKO / 4;                            // float4 per tile row
  constexpr int Y_PER = BR * Y4 / NT, X_PER = BR * X4 / NT;            // 3, 4
  f32x4 ry[Y_PER], rx[X_PER];

  auto load_chunk = [&](int c) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < CARCA_MAX_SEGS; ++i)
      if (i < D.nseg && c >= args.chunk_start[i]) s = i;
    const CarcaWgradSeg sg = D.seg[s];
    const int r0 = (c - args.chunk_start[s]) * BR;
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int slot = tid + i * NT;
      const int r = slot / Y4, c4 = slot - r * Y4;
      const int row = r0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      bool ok = row < sg.rows;
      if (ok && D.mask_rows) ok = sg.ids[row] != 0;
      if (ok) {
        const float* p = sg.dy + (size_t)row * D.ld_dy + n0 + c4 * 4;
        if (n_full) {
          if constexpr (BUF)
            v = gload4(sg.dy, row * D.ld_dy + n0 + c4 * 4);
          else
            v = *reinterpret_cast<const f32x4_u*>(p);
        } else {
          const int nn = n0 + c4 * 4;
          v[0] = nn + 0 < D.N ? p[0] : 0.f;
          v[1] = nn + 1 < D.N ? p[1] : 0.f;
          v[2] = nn + 2 < D.N ? p[2] : 0.f;
          v[3] = nn + 3 < D.N ? p[3] : 0.f;
        }
      }
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
      const int slot = tid + i * NT;
      const int r = slot / X4, c4 = slot - r * X4;
      const int row = r0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < sg.rows) {
        const int64_t bs = src1 ? sg.x1_bstride : sg.x_bstride;
        const size_t roff = (!src1 && sg.x_gather) ? (size_t)sg.ids[row] * ldx
                            : bs                    ? (size_t)(row / sg.T) * bs + (size_t)(row % sg.T) * ldx
                                                    : (size_t)row * ldx;
        const float* p = (src1 ? sg.x1 : sg.x) + roff + k0 + c4 * 4;
        if (k_full) {
          if constexpr (BUF)
            v = gload4(src1 ? sg.x1 : sg.x, (int)roff + k0 + c4 * 4);
          else
            v = *reinterpret_cast<const f32x4_u*>(p);
        } else {
          const int kk = k0 + c4 * 4;
          v[0] = kk + 0 < klen ? p[0] : 0.f;
          v[1] = kk + 1 < klen ? p[1] : 0.f;
          v[2] = kk + 2 < klen ? p[2] : 0.f;
          v[3] = kk + 3 < klen ? p[3] : 0.f;
        }
      }
      rx[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int slot = tid + i * NT;
      *reinterpret_cast<f32x4*>(&Ys[slot * 4]) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
      const int slot = tid + i * NT;
      *reinterpret_cast<f32x4*>(&Xs[slot * 4]) = rx[i];
    }
  };

  // wave w owns k columns 32w..32w+31 of the block's 128 and all three 32-wide n tiles
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;  // thread t < 96 of a kb == 0 block sums column n0 + t of dY

  const int lr = lane & 31, lh = lane >> 5;
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    if (c + 1 < c_end) load_chunk(c + 1);
    // D[m = n index][n = k index] = sum_r Ys[r][m] * Xs[r][n]:  A lane (i, kk) = Ys[2s + kk][i]
#pragma unroll 4
    for (int st = 0; st < BR / 2; ++st) {
      const int r = 2 * st + lh;
      const float xb = Xs[r * BKO + wave * 32 + lr];
      const float y0 = Ys[r * BNO + lr], y1 = Ys[r * BNO + 32 + lr], y2 = Ys[r * BNO + 64 + lr];
      acc[0] = mfma32(y0, xb, acc[0]);
      acc[1] = mfma32(y1, xb, acc[1]);
      acc[2] = mfma32(y2, xb, acc[2]);
    }
    if (D.db && kb == 0 && tid < BNO) {
#pragma unroll 8
      for (int r = 0; r < BR; ++r) bsum += Ys[r * BNO + tid];
    }
    __syncthreads();
    if (c + 1 < c_end) {
      store_chunk();
      __syncthreads();
    }
  }

  // D row (= n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (= k) = lane&31
  const int k = k0 + wave * 32 + lr;
  if (args.part) {
    float* dst = args.part + ((size_t)(split * args.nnb + nb) * args.nkb + kb) * (BNO * BKO);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(t * 16 + r) * NT + tid] = acc[t][r];
  } else if (k < klen) {
    const int kcol = (src1 ? D.K : 0) + k;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < D.N) {
          if (args.diag_plain_store)
            D.dw[(size_t)n * D.ldw + kcol] = acc[t][r];
          else
            grad_add(&D.dw[(size_t)n * D.ldw + kcol], acc[t][r]);
        }
      }
  }
  if (D.db && kb == 0 && tid < BNO && n0 + tid < D.N) grad_add(&D.db[n0 + tid], bsum);
}

template <int BNO, int BKO, int BR, bool BUF = false>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const WgradDev args) {
  wgrad_body<BNO, BKO, BR, BUF>(args, blockIdx.x);
}

// Several independent products in ONE launch: the d x d weight gradients of a backward pass are ~20 us latency-bound
// launches of ~100 blocks each; side by side they fill the chip and cost one launch.  Block -> (problem, local block).
constexpr int WGRAD_GROUP_MAX = 32;
struct WgradGroupIndex {
  int n;
  int block_start[WGRAD_GROUP_MAX + 1];
};
template <int BNO, int BKO, int BR>
__global__ __launch_bounds__(256) void gemm_wgrad_group_kernel(const WgradDev* __restrict__ devs,
                                                               const WgradGroupIndex idx) {
  int p = 0;
  for (int i = 1; i < idx.n; ++i)
    if ((int)blockIdx.x >= idx.block_start[i]) p = i;
  wgrad_body<BNO, BKO, BR, true>(devs[p], (int)blockIdx.x - idx.block_start[p]);
}

// dw tile (nb, kb) += its splits' partial tiles, in split order (fixed: bit-reproducible).  12 blocks of 256 threads per
// tile; a thread takes four consecutive floats of the register-order tile: the same n, four consecutive k.
__device__ __forceinline__ void wgrad_part_reduce_body(const WgradDev& g, int lb) {
  const CarcaWgradDesc& D = g.d;
  const int tile = lb / 12, sl = lb - tile * 12;
  const int nb = tile / g.nkb, kb = tile - nb * g.nkb;
  const int q = (sl * 256 + (int)threadIdx.x) * 4;  // 0 .. 12284
  const int e = q >> 8, t = q & 255;
  const int wave = t >> 6, lane = t & 63, lr = lane & 31, lh = lane >> 5;
  const size_t tile_fl = 96 * 128;
  // eight splits' loads in flight at a time (a plain loop waits for every load before it issues the next: 22 us for
  // 40 MB); the additions stay in split order
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  const float* base = g.part + ((size_t)nb * g.nkb + kb) * tile_fl + q;
  const size_t step = (size_t)g.nnb * g.nkb * tile_fl;
  int s = 0;
  for (; s + 8 <= g.nsplit; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + (size_t)(s + i) * step);
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
  }
  for (; s < g.nsplit; ++s) sum += *reinterpret_cast<const f32x4*>(base + (size_t)s * step);
  const int tt = e >> 4, r = e & 15;
  const int n = nb * 96 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
  if (n >= D.N) return;
  const bool src1 = kb >= g.nkb0;
  const int k0 = (src1 ? kb - g.nkb0 : kb) * 128, klen = src1 ? D.K1 : D.K;
  float* row = D.dw + (size_t)n * D.ldw + (src1 ? D.K : 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + wave * 32 + lr + i;
    if (k < klen) row[k] += sum[i];
  }
}
__global__ __launch_bounds__(256) void wgrad_part_reduce_kernel(const WgradDev g) { wgrad_part_reduce_body(g, blockIdx.x); }
__global__ __launch_bounds__(256) void wgrad_part_reduce_group_kernel(const WgradDev* __restrict__ devs,
                                                                      const WgradGroupIndex idx) {
  int p = 0;
  for (int i = 1; i < idx.n; ++i)
    if ((int)blockIdx.x >= idx.block_start[i]) p = i;
  wgrad_part_reduce_body(devs[p], (int)blockIdx.x - idx.block_start[p]);
}

}  // namespace

template <int BM, int BN, int BK, int PF, bool BUF = false>
static int launch_gemm_rows(const CarcaGemmDesc* desc, hipStream_t stream) {
  GemmDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (desc->seg[s].rows + BM - 1) / BM;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = (desc->ncols_out + BN - 1) / BN;
  const int grid = rb * g.ncb;
  if (g_rows_log_on) {
    char nm[64];
    snprintf(nm, sizeof(nm), "gemm_rows_kernel<%d,%d,%d,%d,%d>", BM, BN, BK, PF, (int)BUF);
    carca_rows_log(nm, desc, grid);
  }
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))
    hipExtLaunchKernelGGL((gemm_rows_kernel<BM, BN, BK, PF, BUF>), dim3(grid), dim3((BM / 32) * 64), 0, stream, e0, e1, 0, g);
  else
    hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, BK, PF, BUF>), dim3(grid), dim3((BM / 32) * 64), 0, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

template <int DBG, int TN = 3>
static int launch_gemm_rows_cu(const CarcaGemmDesc* desc, hipStream_t stream, const CarcaGatherArgs* pas = nullptr,
                               int* rode = nullptr) {
  GemmDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (desc->seg[s].rows + 383) / 384;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = (desc->ncols_out + 32 * TN - 1) / (32 * TN);
  g.dbg = carca_debug_buffer();
  g.diag = carca_tuning(CARCA_TUNE_DIAG);  // (bit 6: the general tail instead of the 8-wide context group -- A/B)
  int grid = rb * g.ncb;
  // A CU left over in the (single) round takes the gather -- if the tiles keep the others busy for longer than the lone
  // workgroup needs: ~4.2 us per 32-k step here, ~12 ns per gathered row there (19 k rows: 0.2 ms against 0.54 ms at C2).
  // With few attributes the launch would last as long as its passenger (n_attrs = 64: 0.55 ms per forward instead of
  // 0.15), so the gather keeps its own 8 us launch then.
  const double tiles_ns = 4200.0 * ((desc->K0 + 31) / 32 + (desc->K1 + 31) / 32);
  if (pas && grid < carca_num_cus() && tiles_ns >= 2.0 * 12.0 * pas->total_rows) {
    g.has_pas = 1;
    g.pas = *pas;
    ++grid;
    if (rode) *rode = 1;
  }
  if (g_rows_log_on) carca_rows_log(TN == 4 ? "gemm_rows_cu_kernel<0,4>" : "gemm_rows_cu_kernel<0,3>", desc, grid);
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))
    hipExtLaunchKernelGGL((gemm_rows_cu_kernel<DBG, TN>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
  else
    hipLaunchKernelGGL((gemm_rows_cu_kernel<DBG, TN>), dim3(grid), dim3(768), 0, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

// gemm_rows_sk_kernel's launch (see the kernel): returns 1 when the product is not its shape -- the caller then launches
// gemm_rows_cu_kernel<0, 3>.  The partial tiles and flags are stream scratch (carca_common.h: one buffer per stream, its
// launches ordered by the stream -- no ring, no event, no host wait); inside a hipGraph capture they belong to the capture.
int* g_sk_err_host = nullptr;  // mapped host memory: written by a taker whose wait expired; read before every launch and by
int* g_sk_err_dev = nullptr;   // carca_poll_errors

// A kernel's failure that no launch status can carry (today: a stream-K taker that gave up waiting): CARCA_OK, or
// CARCA_ERR_UNSUPPORTED with the message set -- the word is cleared by the call that reports it.
// the word's device view, allocated on first use (mapped + portable host memory: one word for every device of the process);
// null when it cannot be had -- the kernels then keep their findings to themselves
int* carca_kernel_error_word() {
  if (!g_sk_err_host) {
    if (hipHostMalloc((void**)&g_sk_err_host, sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    *g_sk_err_host = 0;
    if (hipHostGetDevicePointer((void**)&g_sk_err_dev, g_sk_err_host, 0) != hipSuccess) g_sk_err_dev = nullptr;
  }
  return g_sk_err_dev;
}
static int sk_check_error_word() {
  if (g_sk_err_host && *(volatile int*)g_sk_err_host == 3) {
    *(volatile int*)g_sk_err_host = 0;
    carca_set_error("gemm_wgrad: an EARLIER gemm_wgrad_cu_kernel launch met a group whose range touches more k blocks than it "
                    "has partial-tile slots (that tile's gradient is wrong): a launcher bug -- carca_set_tuning(0, 14) "
                    "selects the atomic flush");
    return CARCA_ERR_UNSUPPORTED;
  }
  if (g_sk_err_host && *(volatile int*)g_sk_err_host == 2) {
    *(volatile int*)g_sk_err_host = 0;
    carca_set_error("gemm_rows: an EARLIER gemm_rows_skc_kernel launch met a stretch of more row blocks than its lists hold "
                    "(its output is wrong): a launcher bug -- carca_set_tuning(0, 23) selects the kernel over every row");
    return CARCA_ERR_UNSUPPORTED;
  }
  if (g_sk_err_host && *(volatile int*)g_sk_err_host != 0) {
    *(volatile int*)g_sk_err_host = 0;
    carca_set_error("gemm_rows: an EARLIER stream-K launch gave up waiting for a partial tile (its output is wrong): "
                    "were its workgroups not all resident?  carca_set_tuning(0, 15) selects the kernel without the hand-over");
    return CARCA_ERR_UNSUPPORTED;
  }
  return CARCA_OK;
}
extern "C" int carca_poll_errors(void) { return sk_check_error_word(); }

static int launch_gemm_rows_sk(const CarcaGemmDesc* desc, hipStream_t stream, const CarcaGatherArgs* pas, int* rode) {
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  if (variant == 15 || variant == 158) return 1;  // (15: never -- A/B switch; 158: 15 + 8)
  const int ncb = (desc->ncols_out + 95) / 96, nfull = ncb - 1;
  if (ncb < 2 || desc->ncols_out != desc->N) return 1;
  const int rem = desc->N - 96 * nfull, xc = rem - 64;
  if (xc < 1 || xc > 2) return 1;  // (the cheap tile is two MFMA column tiles + 1..2 VALU columns)
  if (desc->colvec || desc->pos || desc->gate_scale != 0.f || desc->add_table) return 1;
  int rb = 0;
  GemmDev g{};
  g.d = *desc;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    if (sg.add || sg.gate || sg.rowscale || sg.add_pos) return 1;  // (the VALU columns take the plain epilogue only)
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (sg.rows + 383) / 384;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = ncb;
  int grid = rb * ncb;
  const int nfast = desc->K0 / 32, ntail = (desc->K0 + 31) / 32 - nfast + (desc->K1 + 31) / 32;
  if (grid > carca_num_cus() || nfast < 64) return 1;  // (one round: every workgroup resident, nobody waits for an unborn giver)
  // balance: owner (nfast - don + ntail) steps of a full tile; giver its own (nfast + ntail) steps at c of a full step (two
  // MFMA column tiles of three + the VALU columns: 0.74 measured) + nfull x (don + p) of them, p = 1.5 steps of pipeline fill
  // per partial tile -- and it must END first (the last owner waits for its partial).  C2: 6 (tools/sk_sweep.py: 4 / 5 / 6 /
  // 7 / 8 steps 526 / 522 / 520 / 532 / 548 us).  Tuning key 11 overrides don.
  const double c = (2.0 + 0.08 * xc) / 3.0;
  int don = (int)(((1.0 - c) * (nfast + ntail) - 1.5 * nfull) / ncb);
  if (carca_tuning(CARCA_TUNE_SK_DON) > 0) don = carca_tuning(CARCA_TUNE_SK_DON);
  if (don < 1 || don >= nfast) return 1;
  g.sk_don = don;
  // The item-row gather rides as workgroup `grid` on the CU the tiles leave idle (gather_rows_dma: ~100 us; the register
  // version measured ~540 us beside the tiles and was the launch's long pole: 0.640 ms per forward against 0.614).
  // Tuning variant 19: the gather keeps its own launch (A/B switch).
  if (pas && grid < carca_num_cus() && pas->d <= 128 && variant != 19 &&
      4200.0 * (nfast - don + ntail) >= 1.25 * 20.0 * pas->total_rows) {  // (the tiles outlast the passenger, 20 ns per row, with a margin)
    g.has_pas = 1;
    g.pas = *pas;
    ++grid;
    if (rode) *rode = 1;
  }
  if (!g_sk_err_host) {
    if (hipHostMalloc((void**)&g_sk_err_host, sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return 1;
    *g_sk_err_host = 0;
    if (hipHostGetDevicePointer((void**)&g_sk_err_dev, g_sk_err_host, 0) != hipSuccess) g_sk_err_dev = nullptr;
  }
  if (int rc = sk_check_error_word()) return rc;
  g.sk_err = g_sk_err_dev;
  {
    const int lg = carca_tuning(CARCA_TUNE_SK_SPIN_LOG2);
    g.sk_spin = 1u << (lg > 0 && lg < 31 ? lg : 23);
    g.sk_withhold = carca_tuning(CARCA_TUNE_SK_WITHHOLD);
  }
  // (a FIXED flag area in front: one buffer serves every shape launched on its stream, and what was cleared when it was
  // allocated must cover the flags of all of them -- one per partial tile, at most one workgroup per CU gives)
  const size_t n_part = (size_t)rb * nfull, flag_bytes = 4096;
  if (n_part * sizeof(int) > flag_bytes) return 1;
  const size_t bytes = flag_bytes + n_part * 384 * 96 * sizeof(float);
  // (the flags are cleared when the buffer is allocated and are zero between launches from then on: every taker resets its own)
  char* buf = (char*)(carca_stream_capturing(stream) ? carca_capture_alloc(stream, bytes, false, nullptr, flag_bytes)
                                                     : carca_stream_scratch(stream, CARCA_SCRATCH_SK, bytes, flag_bytes));
  if (!buf) return (int)hipErrorOutOfMemory;
  g.sk_flag = (int*)buf;
  g.sk_part = (float*)(buf + flag_bytes);
  if (g_rows_log_on) carca_rows_log(xc == 1 ? "gemm_rows_sk_kernel<1>" : "gemm_rows_sk_kernel<2>", desc, grid);
  hipEvent_t e0, e1;
  const bool ev = carca_take_launch_events(&e0, &e1);
  if (xc == 1) {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_sk_kernel<1>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_sk_kernel<1>), dim3(grid), dim3(768), 0, stream, g);
  } else {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_sk_kernel<2>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_sk_kernel<2>), dim3(grid), dim3(768), 0, stream, g);
  }
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

// gemm_rows_skc_kernel: the stream-K kernel over the rows with id != 0.  1 = not this product's kernel.
static int launch_gemm_rows_skc(const CarcaGemmDesc* desc, hipStream_t stream, const CarcaGatherArgs* pas, int* rode) {
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  if (variant == 15 || variant == 158 || variant == 23) return 1;  // (23: gemm_rows_sk_kernel, every row -- A/B switch)
  if (!desc->mask_rows) return 1;  // (rows with id 0 may be left out only where the product masks them)
  const int ncb = (desc->ncols_out + 95) / 96, nfull = ncb - 1;
  if (ncb < 2 || desc->ncols_out != desc->N) return 1;
  // (the narrow last column block: two MFMA column tiles -- of 33..64 columns -- + up to 2 VALU columns)
  const int rem = desc->N - 96 * nfull, xc = rem > 64 ? rem - 64 : 0;
  if (rem <= 32 || xc > 2) return 1;
  if (desc->colvec || desc->pos || desc->add_table) return 1;  // (gate_scale only matters with a gate, and a segment with one is refused below)
  long rows = 0;
  GemmDev g{};
  g.d = *desc;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    if (sg.add || sg.gate || sg.rowscale || sg.add_pos || !sg.ids) return 1;
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    rows += sg.rows;
  }
  for (int s = desc->nseg; s < CARCA_MAX_SEGS; ++s) g.d.seg[s] = CarcaGemmSeg{};  // (the kernel reads all four: rows = 0)
  const int nfast = desc->K0 / 32;
  const int ncu = carca_num_cus();
  // (worth a persistent grid: at least most of a round of tiles if every row counted, and K long enough to share)
  const int min_fast = carca_tuning(19) > 0 ? carca_tuning(19) : SKC_MIN_STEPS;  // (tuning key 19: the bound, A/B)
  if (nfast < min_fast || ((rows + 383) / 384) * ncb < ncu / 2 || ncu > 1024) return 1;
  // (what the kernel's LDS lists hold: 64-row chunks of all segments, and SKC_RB row blocks per workgroup's stretch)
  {
    // a stretch of S row blocks touches at most ceil(S) + 1 of them; the longer stretches are the lone workgroups' (the
    // kernel's own split of the grid into x teams and y lone workgroups, with every row kept)
    const long nrb_max = (rows + 383) / 384 + desc->nseg, nblk = ncu - 1;
    const long cheap = xc == 2 ? 74 : (xc == 1 ? 71 : 68);
    const long x = std::max(1l, nblk * 100 / (nfull * 100 + cheap)), y = std::max(1l, nblk - (x + 1) * nfull);
    // (a stretch is at most ceil(nrb (nfast + ov) / teams) steps long and touches one block more than it spans)
    const long ovm = SKC_OV_MAX, wv = nfast + ovm;
    if (rows / 64 + 2 * desc->nseg > SKC_CH || (nrb_max * wv + x * nfast - 1) / (x * nfast) + 1 > SKC_RB ||
        (nrb_max * wv + y * nfast - 1) / (y * nfast) + 1 > SKC_RB)
      return 1;
  }
  g.ncb = ncb;
  g.skc_cheap = xc == 2 ? 74 : (xc == 1 ? 71 : 68);  // (two MFMA column tiles of three + the VALU columns: 0.74 of a full step measured)
  {
    // owning a row block, in K steps (tuning keys 17 / 18: value - 1, so that 1 switches the correction off; A/B)
    const int t = carca_tuning(17), tl = carca_tuning(18);
    g.skc_ov = std::min(SKC_OV_MAX, t > 0 ? t - 1 : SKC_OV_TEAM);
    g.skc_ov_lone = std::min(SKC_OV_MAX, tl > 0 ? tl - 1 : SKC_OV_LONE);
  }
  if (!g_sk_err_host) {
    if (hipHostMalloc((void**)&g_sk_err_host, sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return 1;
    *g_sk_err_host = 0;
    if (hipHostGetDevicePointer((void**)&g_sk_err_dev, g_sk_err_host, 0) != hipSuccess) g_sk_err_dev = nullptr;
  }
  if (int rc = sk_check_error_word()) return rc;
  g.sk_err = g_sk_err_dev;
  {
    const int lg = carca_tuning(CARCA_TUNE_SK_SPIN_LOG2);
    g.sk_spin = 1u << (lg > 0 && lg < 31 ? lg : 23);
    g.sk_withhold = carca_tuning(CARCA_TUNE_SK_WITHHOLD);
  }
  int grid = ncu;
  // The gather rides (last workgroup: gather_rows_dma, ~20 ns per row measured beside the tiles: 392 us for C2's 19 k rows)
  // only under a product that lasts well beyond it even with half of its rows left out -- not at C2 (the kernel is ~480 us in
  // evaluation, ~330 us in training; riding measured 0.5653 ms per forward against 0.5637 with the gather's own 7 us launch).
  if (pas && pas->d <= 128 && variant != 19 && 4200.0 * nfast * 0.5 >= 1.5 * 20.0 * pas->total_rows) {
    g.has_pas = 1;
    g.pas = *pas;
    if (rode) *rode = 1;
  }
  // partial tiles: one slot per workgroup
  const bool cap = carca_stream_capturing(stream);
  const size_t flag_bytes = 4096, part_bytes = flag_bytes + (size_t)grid * 384 * 96 * sizeof(float);
  char* parts = (char*)(cap ? carca_capture_alloc(stream, part_bytes, false, nullptr, flag_bytes)
                            : carca_stream_scratch(stream, CARCA_SCRATCH_SKC_PART, part_bytes, flag_bytes));
  if (!parts) return (int)hipErrorOutOfMemory;
  g.sk_flag = (int*)parts;
  g.sk_part = (float*)(parts + flag_bytes);
  g.dbg = carca_debug_buffer();
  g.diag = carca_tuning(CARCA_TUNE_DIAG);  // (bit 0: the prologue's id loads read nothing -- timing experiment, wrong results)
  if (g_rows_log_on)
    carca_rows_log(xc == 0 ? "gemm_rows_skc_kernel<0>" : (xc == 1 ? "gemm_rows_skc_kernel<1>" : "gemm_rows_skc_kernel<2>"), desc, grid);
  hipEvent_t e0, e1;
  const bool ev = carca_take_launch_events(&e0, &e1);
  if (xc == 0) {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_skc_kernel<0>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_skc_kernel<0>), dim3(grid), dim3(768), 0, stream, g);
  } else if (xc == 1) {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_skc_kernel<1>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_skc_kernel<1>), dim3(grid), dim3(768), 0, stream, g);
  } else {
    if (ev) hipExtLaunchKernelGGL((gemm_rows_skc_kernel<2>), dim3(grid), dim3(768), 0, stream, e0, e1, 0, g);
    else hipLaunchKernelGGL((gemm_rows_skc_kernel<2>), dim3(grid), dim3(768), 0, stream, g);
  }
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

static int launch_gemm_rows_n96(const CarcaGemmDesc* desc, hipStream_t stream) {
  GemmDev g{};
  g.d = *desc;
  int rb = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.rb_start[s] = rb;
    rb += (desc->seg[s].rows + n96::BM - 1) / n96::BM;
  }
  g.rb_start[desc->nseg] = rb;
  g.nrb = rb;
  g.ncb = 1;
  g.diag = carca_tuning(CARCA_TUNE_DIAG);
  constexpr size_t lds_bytes = sizeof(float) * 2 * (n96::A_BUF + n96::B_BUF);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_rows_n96_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
      carca_set_error("gemm_rows: cannot reserve %zu B of LDS: %s", lds_bytes, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  if (g_rows_log_on) carca_rows_log("gemm_rows_n96_kernel", desc, rb);
  hipEvent_t e0, e1;
  if (carca_take_launch_events(&e0, &e1))
    hipExtLaunchKernelGGL(gemm_rows_n96_kernel, dim3(rb), dim3(768), lds_bytes, stream, e0, e1, 0, g);
  else
    hipLaunchKernelGGL(gemm_rows_n96_kernel, dim3(rb), dim3(768), lds_bytes, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

enum GemmChoice { GEMM_NARROW_BUF, GEMM_NARROW, GEMM_CU, GEMM_CU_STAMPS, GEMM_CU128, GEMM_TILED_BUF, GEMM_TILED, GEMM_WIDE64, GEMM_WIDE64_PF2, GEMM_N96 };

// argument checks + kernel selection of one product
static int gemm_rows_choose(const CarcaGemmDesc* desc, GemmChoice* choice, bool* fits_out = nullptr) {
  CARCA_CHECK_ARG(desc && desc->nseg >= 1 && desc->nseg <= CARCA_MAX_SEGS, "gemm_rows: bad segment count");
  CARCA_CHECK_ARG(desc->bt0 && desc->K0 >= 1 && desc->N >= 1 && desc->ldc >= desc->N && desc->lda0 >= desc->K0 &&
                      desc->ldb0 >= desc->K0,
                  "gemm_rows: bad k-source 0 / output geometry");
  CARCA_CHECK_ARG(desc->K1 == 0 || (desc->bt1 && desc->lda1 >= desc->K1 && desc->ldb1 >= desc->K1),
                  "gemm_rows: bad k-source 1");
  CARCA_CHECK_ARG(desc->ncols_out >= desc->N && desc->ncols_out <= desc->ldc, "gemm_rows: ncols_out outside [N, ldc]");
  int rb128 = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.a0 && sg.c && (desc->K1 == 0 || sg.a1), "gemm_rows: segment %d malformed", s);
    CARCA_CHECK_ARG(sg.T >= 1 || (!sg.a0_bstride && !sg.a1_bstride && !sg.add_pos), "gemm_rows: segment %d needs T >= 1",
                    s);
    CARCA_CHECK_ARG(!(sg.add_pos && (!desc->pos || sg.T < 1 || !sg.ids)) && !(desc->mask_rows && !sg.ids) &&
                        !(sg.rowscale && !desc->colvec) && !(sg.a0_gather && !sg.ids) && !(desc->add_table && !sg.ids),
                    "gemm_rows: segment %d epilogue needs a pointer that is NULL", s);
    rb128 += (sg.rows + 127) / 128;
  }
  // Narrow outputs (the joint embedding, every d-wide product of the backward pass) give too few 128 x 96 blocks
  // to fill 256 CUs and leave one long MFMA chain per wave: 32-column blocks triple the wave count instead
  // (the A tile is re-read from L2 by the three column blocks of a row block, which share an XCD).
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);  // 0 auto, 1 force 128x96, 2 force one-block-per-CU, 3 = 2 + stamps, 4 = no buffer loads
  // do all operand offsets fit 32 bits of bytes?  (buffer loads; the gather table's size is only known when stated)
  const uint64_t lim = 1ull << 30;  // elements
  bool fits = (uint64_t)(desc->N - 1) * desc->ldb0 + desc->K0 < lim &&
              (desc->K1 == 0 || (uint64_t)(desc->N - 1) * desc->ldb1 + desc->K1 < lim);
  int rb384 = 0;
  for (int s = 0; s < desc->nseg && fits; ++s) {
    const CarcaGemmSeg& sg = desc->seg[s];
    const int T = sg.T >= 1 ? sg.T : 1;
    const uint64_t ub = (uint64_t)((sg.rows - 1) / T), ut = (uint64_t)(T - 1);
    const uint64_t last0 = sg.a0_gather ? (sg.a0_gather > 1 ? (uint64_t)(sg.a0_gather - 1) * desc->lda0 : lim)
                           : sg.a0_bstride ? ub * sg.a0_bstride + ut * desc->lda0
                                           : (uint64_t)(sg.rows - 1) * desc->lda0;
    const uint64_t last1 = desc->K1 == 0 ? 0
                           : sg.a1_bstride ? ub * sg.a1_bstride + ut * desc->lda1
                                           : (uint64_t)(sg.rows - 1) * desc->lda1;
    fits = last0 + desc->K0 < lim && last1 + desc->K1 < lim;
    rb384 += (sg.rows + 383) / 384;
  }
  if (variant == 4) fits = false;
  if (fits_out) *fits_out = fits;
  const bool narrow = variant != 1 && rb128 * ((desc->ncols_out + 95) / 96) < 384;
  if (narrow) {
    *choice = fits ? GEMM_NARROW_BUF : GEMM_NARROW;
    if (fits && (variant == 9 || variant == 10)) *choice = variant == 9 ? GEMM_WIDE64 : GEMM_WIDE64_PF2;
    // one 80 x 96 block per CU: narrow output over ONE k-source with a K worth pipelining, rows that fill the chip in about
    // one round, every operand row 16-byte aligned (variant 11 forces it where it applies, 12 forbids it)
    // (a product with a gathered addend -- the joint embedding over q's columns, K0 = g = 450 at C2 -- comes with rows
    // that are only 8-byte aligned and a K that is no multiple of 4: the kernel's loads take any dword alignment and its
    // last stage shifts / masks by element; everybody else keeps the 16-byte conditions the kernel was measured under)
    const bool loose = desc->add_table != nullptr && desc->K0 >= 4;
    if (fits && variant != 12 && desc->K1 == 0 && desc->N > 64 && desc->ncols_out <= 96 &&
        (loose || (desc->K0 % 4 == 0 && desc->lda0 % 4 == 0 && desc->ldb0 % 4 == 0 && ((uintptr_t)desc->bt0 & 15) == 0))) {
      bool ok = true;
      int rb80 = 0;
      for (int s = 0; s < desc->nseg; ++s) {
        const CarcaGemmSeg& sg = desc->seg[s];
        ok = ok && !sg.a0_gather && (loose || (((uintptr_t)sg.a0 & 15) == 0 && sg.a0_bstride % 4 == 0));
        rb80 += (sg.rows + n96::BM - 1) / n96::BM;
      }
      const int cus = carca_num_cus();
      if (ok && (variant == 11 || (variant == 0 && desc->K0 >= 256 && rb80 > cus / 2 && rb80 <= cus))) *choice = GEMM_N96;
    }
    return CARCA_OK;
  }
  // One 384 x 96 block per CU when the grid fills the chip's 256 CUs about as well as the 128 x 96 blocks (3 per CU)
  // would: compare rounds x tiles per block.
  if (fits && desc->K0 >= 64 && variant != 1) {
    const int ncb = (desc->ncols_out + 95) / 96, ncb128 = (desc->ncols_out + 127) / 128;
    const long units_cu = (long)((rb384 * ncb + 255) / 256) * 36, units_3 = (long)((rb128 * ncb + 255) / 256) * 12;
    const long units_cu128 = (long)((rb384 * ncb128 + 255) / 256) * 48;  // 384 x 128 tiles: 48 32x32 tiles per block
    // (a unit of the one-block-per-CU kernel is ~1.2x cheaper than one of the 128 x 96 kernel -- 84 % against 68 % of the
    // MFMA peak on the feature GEMM -- so a round-up that costs it a few per cent more units is still a win: n_attrs =
    // 512, N = 1001, B = 512 gives 1008 against 996 units and ran 141.7 k users/s tiled between 148 k at B = 256 and
    // 157 k at B = 1024, both one-block-per-CU)
    if (variant == 3 || variant == 2 || (units_cu * 10 <= units_3 * 11 && units_cu <= units_cu128)) {
      *choice = variant == 3 ? GEMM_CU_STAMPS : GEMM_CU;
      return CARCA_OK;
    }
    if (variant == 7 || (variant == 0 && units_cu128 < units_cu && units_cu128 * 10 <= units_3 * 11)) {  // (7: force 384 x 128)
      *choice = GEMM_CU128;
      return CARCA_OK;
    }
  }
  *choice = fits ? GEMM_TILED_BUF : GEMM_TILED;
  return CARCA_OK;
}

extern "C" int carca_gemm_rows(const CarcaGemmDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  GemmChoice c;
  bool fits = false;
  if (int rc = gemm_rows_choose(desc, &c, &fits)) return rc;
  if (desc->N <= 96) {  // (a narrow output over many rows per CU: gemm_stream.hip)
    const int rc = carca_gemm_rows_n96s_try(desc, fits, stream);
    if (rc != 1) return rc;
  }
  if ((carca_tuning(CARCA_TUNE_SPLIT_GEMM) & 16) && fits && c != GEMM_CU) {
    // (key 16, bit 4: the split-precision kernel wherever its own conditions hold, whatever the grid -- how the fixture-sized
    // parity tests reach it)
    const int rc = carca_gemm_rows_split_try(desc, stream);
    if (rc != 1) return rc;
  }
  switch (c) {
    case GEMM_NARROW_BUF: return launch_gemm_rows<128, 32, 32, 4, true>(desc, stream);
    case GEMM_NARROW: return launch_gemm_rows<128, 32, 32, 4>(desc, stream);
    case GEMM_CU: {
      int rc = carca_gemm_rows_split_try(desc, stream);  // (opt-in, tuning key 16; 1 = not asked for / not its product)
      if (rc != 1) return rc;
      rc = launch_gemm_rows_skc(desc, stream, nullptr, nullptr);
      if (rc != 1) return rc;
      rc = launch_gemm_rows_sk(desc, stream, nullptr, nullptr);
      if (rc != 1) return rc;
      rc = carca_gemm_rows_stream_try(desc, fits, stream);  // (short K, many tiles per CU: gemm_stream.hip)
      return rc == 1 ? launch_gemm_rows_cu<0>(desc, stream) : rc;
    }
    case GEMM_CU_STAMPS: return launch_gemm_rows_cu<1>(desc, stream);
    case GEMM_CU128: {  // (384 x 128 tiles fill one round where 96-wide ones would not -- unless the rows with id 0 can go)
      int rc = launch_gemm_rows_skc(desc, stream, nullptr, nullptr);
      if (rc != 1) return rc;
      rc = carca_gemm_rows_stream_try(desc, fits, stream);
      return rc == 1 ? launch_gemm_rows_cu<0, 4>(desc, stream) : rc;
    }
    case GEMM_WIDE64: return launch_gemm_rows<64, 96, 32, 4, true>(desc, stream);
    case GEMM_WIDE64_PF2: return launch_gemm_rows<64, 96, 32, 2, true>(desc, stream);
    case GEMM_N96: return launch_gemm_rows_n96(desc, stream);
    case GEMM_TILED_BUF: return launch_gemm_rows<128, 96, 32, 1, true>(desc, stream);
    default: return launch_gemm_rows<128, 96, 32, 1>(desc, stream);
  }
}

int carca_gemm_rows_passenger(const CarcaGemmDesc* desc, const CarcaGatherArgs* ga, int* rode, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  *rode = 0;
  GemmChoice c;
  bool fits = false;
  if (int rc = gemm_rows_choose(desc, &c, &fits)) return rc;
  if (carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 8 || carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 158) ga = nullptr;  // (8: never let the gather ride -- A/B switch)
  if (c == GEMM_CU) {
    int rc = carca_gemm_rows_split_try(desc, stream);  // (the gather keeps its own launch beside this kernel: rode stays 0)
    if (rc != 1) return rc;
    rc = launch_gemm_rows_skc(desc, stream, ga, rode);
    if (rc != 1) return rc;
    rc = launch_gemm_rows_sk(desc, stream, ga, rode);
    if (rc != 1) return rc;
    rc = carca_gemm_rows_stream_try(desc, fits, stream);  // (the gather keeps its own launch beside it: rode stays 0)
    return rc == 1 ? launch_gemm_rows_cu<0, 3>(desc, stream, ga, rode) : rc;
  }
  if (c == GEMM_CU128) {
    int rc = launch_gemm_rows_skc(desc, stream, ga, rode);
    if (rc != 1) return rc;
    rc = carca_gemm_rows_stream_try(desc, fits, stream);
    return rc == 1 ? launch_gemm_rows_cu<0, 4>(desc, stream, ga, rode) : rc;
  }
  return carca_gemm_rows(desc, stream_);
}

// n independent products; those that select the narrow buffer-load kernel share launches (GEMM_GROUP_MAX per launch),
// the rest are issued one by one.  Same results as n carca_gemm_rows calls: the products must not depend on each other.
extern "C" int carca_gemm_rows_group(const CarcaGemmDesc* descs, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(descs && n >= 1, "gemm_rows_group: no products");
  constexpr int BM = 128, BN = 32;
  GemmGroup grp{};
  int blocks = 0;
  auto flush = [&]() -> int {
    if (grp.n == 0) return CARCA_OK;
    grp.block_start[grp.n] = blocks;
    if (grp.n == 1) {
      hipLaunchKernelGGL((gemm_rows_kernel<BM, BN, 32, 4, true>), dim3(blocks), dim3(256), 0, stream, grp.g[0]);
    } else {
      hipLaunchKernelGGL((gemm_rows_group_kernel<BM, BN, 32, 4, true>), dim3(blocks), dim3(256), 0, stream, grp);
    }
    CARCA_LAUNCH_CHECK();
    grp.n = 0;
    blocks = 0;
    return CARCA_OK;
  };
  for (int i = 0; i < n; ++i) {
    GemmChoice c;
    if (int rc = gemm_rows_choose(&descs[i], &c)) return rc;
    if (c != GEMM_NARROW_BUF || carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 6) {  // (variant 6: never group -- A/B switch)
      if (int rc = carca_gemm_rows(&descs[i], stream_)) return rc;
      continue;
    }
    GemmDev& g = grp.g[grp.n];
    g = GemmDev{};
    g.d = descs[i];
    int rb = 0;
    for (int s = 0; s < descs[i].nseg; ++s) {
      if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
      g.rb_start[s] = rb;
      rb += (descs[i].seg[s].rows + BM - 1) / BM;
    }
    g.rb_start[descs[i].nseg] = rb;
    g.nrb = rb;
    g.ncb = (descs[i].ncols_out + BN - 1) / BN;
    grp.block_start[grp.n] = blocks;
    blocks += rb * g.ncb;
    if (++grp.n == GEMM_GROUP_MAX)
      if (int rc = flush()) return rc;
  }
  return flush();
}

int carca_wgrad_cu_try(const CarcaWgradDesc* desc, hipStream_t stream);  // wgrad_cu.hip

static int wgrad_check(const CarcaWgradDesc* desc) {
  CARCA_CHECK_ARG(desc && desc->nseg >= 1 && desc->nseg <= CARCA_MAX_SEGS, "gemm_wgrad: bad segment count");
  CARCA_CHECK_ARG(desc->dw && desc->N >= 1 && desc->K >= 1 && desc->K1 >= 0 && desc->ldw >= desc->K + desc->K1 &&
                      desc->ld_dy >= desc->N && desc->ld_x >= desc->K && (desc->K1 == 0 || desc->ld_x1 >= desc->K1),
                  "gemm_wgrad: bad geometry");
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaWgradSeg& sg = desc->seg[s];
    CARCA_CHECK_ARG(sg.rows >= 1 && sg.dy && sg.x && !(desc->mask_rows && !sg.ids) && (desc->K1 == 0 || sg.x1),
                    "gemm_wgrad: segment %d malformed", s);
    CARCA_CHECK_ARG(sg.T >= 1 || (!sg.x_bstride && !sg.x1_bstride), "gemm_wgrad: segment %d needs T >= 1", s);
    CARCA_CHECK_ARG(!(sg.x_gather && !sg.ids), "gemm_wgrad: segment %d gathers without ids", s);
  }
  return CARCA_OK;
}

// tiling / row splits of the tiled kernel for one product; returns the grid size; *fits: buffer loads are safe
static int wgrad_prepare(const CarcaWgradDesc* desc, WgradDev& g, bool* fits_out, int slot_budget = 0) {
  constexpr int BNO = 96, BKO = 128, BR = 32;
  g = WgradDev{};
  g.d = *desc;
  int chunks = 0;
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaWgradSeg& sg = desc->seg[s];
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    g.chunk_start[s] = chunks;
    chunks += (sg.rows + BR - 1) / BR;
  }
  g.chunk_start[desc->nseg] = chunks;
  g.nnb = (desc->N + BNO - 1) / BNO;
  g.nkb0 = (desc->K + BKO - 1) / BKO;
  g.nkb = g.nkb0 + (desc->K1 + BKO - 1) / BKO;
  // row splits: fill the chip's 4 x 256 resident slots in ONE round (119 registers -> 4 blocks per CU;
  // measured at C2: 512 slots 1017 us, 768 972, 1024 809, 1536 865), but keep >= 2 chunks (64 rows) per split
  const int tiles = g.nnb * g.nkb;
  // Products of a few tiles (joint embedding: 5) pay for every split with a full tile of atomics: 384 slots there
  // (N = 90, K = 540, 19328 rows: 1024 slots 74.5 us, 768 65.8, 640 61.6, 512 59.6, 384 57.5, 256 67.6)
  const int slots = slot_budget > 0 ? slot_budget
                    : carca_tuning(CARCA_TUNE_WGRAD_SLOTS) > 0 ? carca_tuning(CARCA_TUNE_WGRAD_SLOTS)
                    : tiles >= 64 ? 1024 : 640;  // (640 since the splits end in plain stores: 384 / 512 / 640 / 768 / 1024 -> train
                                                 // step 1.749 / 1.743 / 1.737 / 1.740 / 1.736 ms, tools/ab_train.py "2=...")
  int nsplit = tiles >= slots ? 1 : slots / tiles;
  const int min_chunks = carca_tuning(4) > 0 ? carca_tuning(4) : 2;  // (measured on the d x d products: 4 -> 21 us, 2 -> 18 us, 1 -> 23 us)
  nsplit = max(1, min(nsplit, (chunks + min_chunks - 1) / min_chunks));
  g.diag_plain_store = carca_tuning(3);
  g.chunks_per_split = (chunks + nsplit - 1) / nsplit;
  g.nsplit = (chunks + g.chunks_per_split - 1) / g.chunks_per_split;
  // buffer loads when every operand offset provably fits 32 bits of bytes (a gather table's size must be stated)
  const uint64_t lim = 1ull << 30;
  bool fits = carca_tuning(CARCA_TUNE_GEMM_VARIANT) != 4;
  for (int s = 0; s < desc->nseg && fits; ++s) {
    const CarcaWgradSeg& sg = desc->seg[s];
    const int T = sg.T >= 1 ? sg.T : 1;
    const uint64_t ub = (uint64_t)((sg.rows - 1) / T), ut = (uint64_t)(T - 1);
    const uint64_t lx = sg.x_gather ? (sg.x_gather > 1 ? (uint64_t)(sg.x_gather - 1) * desc->ld_x : lim)
                        : sg.x_bstride ? ub * sg.x_bstride + ut * desc->ld_x
                                       : (uint64_t)(sg.rows - 1) * desc->ld_x;
    const uint64_t lx1 = desc->K1 == 0 ? 0
                         : sg.x1_bstride ? ub * sg.x1_bstride + ut * desc->ld_x1
                                         : (uint64_t)(sg.rows - 1) * desc->ld_x1;
    fits = lx + desc->K < lim && lx1 + desc->K1 < lim && (uint64_t)sg.rows * desc->ld_dy < lim;
  }
  *fits_out = fits;
  return tiles * g.nsplit;
}

// Partial tiles of the row splits (WgradDev.part): stream scratch (carca_common.h) -- the product's kernel writes them, its
// reduce launch reads them, the next product on the stream is ordered behind both; inside a hipGraph capture the capture
// gets storage of its own.  Tuning variant 19 = atomics (A/B).
namespace {
float* wpart_take(size_t floats, hipStream_t stream) {
  if (carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 19 || carca_tuning(3) != 0) return nullptr;
  if (carca_stream_capturing(stream)) return (float*)carca_capture_alloc(stream, floats * sizeof(float), false, nullptr);
  return (float*)carca_stream_scratch(stream, CARCA_SCRATCH_WPART, floats * sizeof(float));
}
}  // namespace

extern "C" int carca_gemm_wgrad(const CarcaWgradDesc* desc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (int rc = wgrad_check(desc)) return rc;
  // the big product (dW of feats_embed) goes to the persistent one-block-per-CU kernel; tuning variant 4 / 5 = never
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  if (variant != 4 && variant != 5) {
    const int r = carca_wgrad_cu_try(desc, stream);
    if (r != 1) return r;
  }
  constexpr int BNO = 96, BKO = 128, BR = 32;
  WgradDev g;
  bool fits = false;
  const int grid = wgrad_prepare(desc, g, &fits);
  const int tiles = g.nnb * g.nkb;
  if (g.nsplit > 1) g.part = wpart_take((size_t)g.nsplit * tiles * BNO * BKO, stream);  // (one split: nothing to combine)
  if (fits)
    hipLaunchKernelGGL((gemm_wgrad_kernel<BNO, BKO, BR, true>), dim3(grid), dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_wgrad_kernel<BNO, BKO, BR>), dim3(grid), dim3(256), 0, stream, g);
  if (g.part) hipLaunchKernelGGL(wgrad_part_reduce_kernel, dim3(tiles * 12), dim3(256), 0, stream, g);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

// Grouped launch: the kernel reads its descriptors from pinned, device-mapped host memory that THIS thread writes, so a
// slot may only be rewritten once the launch that reads it has finished: every slot carries an event recorded behind its
// launch, and a launch takes the first slot whose event has completed (hipEventQuery -- the host never waits; while none
// has, the pool grows: its size follows the number of launches in flight).  Products that are big enough for the
// persistent kernel, or whose offsets do not fit the buffer-load path, are issued one by one instead.
namespace {
struct GroupSlot {
  WgradDev* host;  // [WGRAD_GROUP_MAX] pinned, mapped into the device's address space (no copy command: a small async H2D
  WgradDev* dev;   // copy turned out to stall the issuing thread until the stream had drained)
  hipEvent_t ev;
  bool used;
};
std::vector<GroupSlot> g_group_slots;
int group_slot_take() {
  int found = -1;
  for (size_t i = 0; i < g_group_slots.size() && found < 0; ++i)
    if (!g_group_slots[i].used || hipEventQuery(g_group_slots[i].ev) == hipSuccess) found = (int)i;
  (void)hipGetLastError();  // (a query of a pending event leaves hipErrorNotReady behind: not the next launch's error)
  if (found >= 0) return found;
  GroupSlot sl{};
  if (hipHostMalloc((void**)&sl.host, sizeof(WgradDev) * WGRAD_GROUP_MAX, hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&sl.dev, sl.host, 0) != hipSuccess ||
      hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess)
    return -1;
  g_group_slots.push_back(sl);
  return (int)g_group_slots.size() - 1;
}
}  // namespace

extern "C" int carca_gemm_wgrad_group(const CarcaWgradDesc* descs, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(descs && n >= 1, "gemm_wgrad_group: no products");
  for (int i = 0; i < n; ++i)
    if (int rc = wgrad_check(&descs[i])) return rc;
  constexpr int BNO = 96, BKO = 128, BR = 32;
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  int done = 0;
  const bool capturing = carca_stream_capturing(stream);  // (hipGraph capture: storage of its own, see carca_common.h)
  while (done < n) {
    int slot = -1;
    WgradDev *host, *dev;
    if (capturing) {
      host = (WgradDev*)carca_capture_alloc(stream, sizeof(WgradDev) * WGRAD_GROUP_MAX, true, (void**)&dev);
      if (!host) return CARCA_ERR_BADARG;
    } else {
      slot = group_slot_take();
      if (slot < 0) {
        carca_set_error("gemm_wgrad_group: cannot allocate a descriptor slot");
        return CARCA_ERR_BADARG;
      }
      host = g_group_slots[slot].host;
      dev = g_group_slots[slot].dev;
    }
    WgradGroupIndex idx{}, ridx{};
    int blocks = 0, rblocks = 0;
    size_t part_floats = 0;
    while (done < n && idx.n < WGRAD_GROUP_MAX) {
      bool fits = false;
      WgradDev g;
      // row-split budget per product (tuning key 5; 0 = the single-product default).  A/B at C2, interleaved in one
      // process (tools/ab_train.py): ungrouped 2.341 ms/step, grouped 2.229, grouped with 64 / 85 / 128 slots per
      // product 2.223 / 2.263 / 2.220 -- the budget does not matter, the single launch does
      // Second look with the kernel trace (tools/train_trace.sh, 13 products of a C2 backward pass in one launch):
      // 1024 slots per product (100 two-chunk splits each) 105 us, 64 -> 72 us, 48 -> 73, 40 -> 76, 32 -> 76, 24 -> 92,
      // 16 -> 108: every split ends with a 96 x 128 tile of atomics, so fewer, longer splits win until the chip runs dry
      // (with partial tiles instead of atomics: 32 -> 1.766 ms per train step, 64 -> 1.744, 96 -> 1.742, 128 -> 1.752)
      const int budget = carca_tuning(5) > 0 ? carca_tuning(5) : 96;
      const int grid = wgrad_prepare(&descs[done], g, &fits, budget);
      const bool big = (long)descs[done].N * (descs[done].K + descs[done].K1) > 96 * 1024;  // single-product path decides
      if (!fits || big || variant == 6) {  // (variant 6: never group -- A/B switch)
        if (int rc = carca_gemm_wgrad(&descs[done], stream_)) return rc;
        ++done;
        continue;
      }
      host[idx.n] = g;
      idx.block_start[idx.n] = blocks;
      ridx.block_start[idx.n] = rblocks;
      blocks += grid;
      rblocks += g.nnb * g.nkb * 12;
      part_floats += (size_t)g.nsplit * g.nnb * g.nkb * BNO * BKO;
      ++idx.n;
      ++done;
    }
    if (idx.n == 0) continue;
    idx.block_start[idx.n] = blocks;
    ridx.n = idx.n;
    ridx.block_start[idx.n] = rblocks;
    float* part = wpart_take(part_floats, stream);
    if (part) {  // every product its own stretch of the slot
      size_t at = 0;
      for (int i = 0; i < idx.n; ++i) {
        host[i].part = part + at;
        at += (size_t)host[i].nsplit * host[i].nnb * host[i].nkb * BNO * BKO;
      }
    }
    hipLaunchKernelGGL((gemm_wgrad_group_kernel<BNO, BKO, BR>), dim3(blocks), dim3(256), 0, stream, dev, idx);
    if (part) hipLaunchKernelGGL(wgrad_part_reduce_group_kernel, dim3(rblocks), dim3(256), 0, stream, dev, ridx);
    if (slot >= 0) {
      (void)hipEventRecord(g_group_slots[slot].ev, stream);
      g_group_slots[slot].used = true;
    }
    CARCA_LAUNCH_CHECK();
  }
  return CARCA_OK;
}
