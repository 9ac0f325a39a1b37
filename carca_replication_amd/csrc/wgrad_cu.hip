// Weight-gradient GEMM for the one product that is as large as the forward feature GEMM:
//   dW_f[n][k] += sum_r dq[r][n] * [attrs ; ctx][r][k]      (carca.py:86 seen from the backward side)
// ~19k rows x 450 x 4102 at C2.  Same recipe as gemm_rows_cu_kernel (gemm.hip): ONE 768-thread block per CU, all
// memory instructions in the shadow of the wave's own MFMAs, double-buffered LDS with one raw barrier per 32-row
// chunk, buffer loads.  What differs:
//   * the contraction runs over ROWS, so both MFMA operands are read from the row-major LDS tiles transposed
//     (lane (i, kk) reads element [2s + kk][i]; two rows per ds_read2), row strides 96 / 416 floats so that the
//     kk = 1 half of the wave lands on the other 32 banks;
//   * the work is (output tile 96 n x 384 k) x (32-row chunk) items, cut into equal contiguous ranges over the
//     persistent blocks ("stream-K"): perfect balance for any shape, a block touches at most two output tiles and
//     flushes each with fp32 atomic adds (dW is accumulated anyway: the caller zeroes it);
//   * every per-row special case (users of a [B, T, K] view, attribute rows gathered from a table by item id, rows
//     masked because ids == 0, the ragged last chunk of a segment) is folded by a tiny pre-kernel into one table of
//     32-bit byte offsets per row; an offset of 2^31 makes the buffer load return zeros, so the hot loop has no
//     conditions;
//   * the few columns of the second k-source (ctx, K1 <= 8) ride along in the last k block: wave 11, whose own
//     columns lie past K there, takes its B operand from the 32 PAD columns of the X tile instead, and fills them
//     itself (one extra 16-byte load per lane and chunk).  Every wave executes those instructions -- with the
//     "no row" offset and a dummy LDS address where it is not its job -- so the loop stays branch-free.
//     Columns past K / K1 are loaded as they come (the next row's data, zeros past the end of the buffer) into
//     accumulator columns that are never written out.
#include <type_traits>
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

constexpr int WG_BN = 96, WG_BK = 384, WG_BR = 32, WG_NW = 12, WG_NT = 768;
constexpr int WG_XS = WG_BK + 32;                       // X tile row stride (floats)
constexpr int WG_YB = WG_BR * WG_BN, WG_XB = WG_BR * WG_XS;  // floats per LDS buffer
constexpr unsigned WG_INV = 0x80000000u;                // row offset that reads as zeros (num_records <= 2^31)

struct WgradCuDev {
  CarcaWgradDesc d;
  int chunk_start[CARCA_MAX_SEGS + 1];
  unsigned y_bytes[CARCA_MAX_SEGS], x_bytes[CARCA_MAX_SEGS], x1_bytes[CARCA_MAX_SEGS];  // buffer extents
  const unsigned* tab;  // [3][V]: byte offsets of dY / X / X1 rows inside their segment, V = 32 * chunks
  int V, nnb, nkb;
  // stream-K ranges in units of 1/256 item: items of the k block that also carries the second k-source (the LAST k block:
  // k blocks are the outermost index) weigh slow_w / 256 -- its chunks run ~5 % longer (one more load per lane), and with
  // equal item counts the workgroups that sit entirely inside it finished 5.4 % behind the rest (measured, C2: block times
  // 1.134 .. 1.417 M cycles; 1.295 .. 1.364 M with a weight of 270 / 256, those workgroups then 2 % early: 264)
  long per_w, n_fast;
  int slow_w;
  int src1_kb;  // k block that also carries the second k-source's columns, or -1
  unsigned long long* dbg;    // phase stamps of a diagnostic run (tuning variant 3)
};
__device__ __forceinline__ unsigned long long* carca_debug_ptr(const WgradCuDev& a) { return a.dbg; }

__global__ void wgrad_rowtab_kernel(const WgradCuDev args, unsigned* tab) {
  const CarcaWgradDesc& D = args.d;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= args.V) return;
  const int c = v >> 5;
  int s = 0;
  for (int i = 1; i < CARCA_MAX_SEGS; ++i)
    if (i < D.nseg && c >= args.chunk_start[i]) s = i;
  const CarcaWgradSeg sg = D.seg[s];
  const int row = v - args.chunk_start[s] * WG_BR;
  unsigned yo = WG_INV, xo = WG_INV, x1o = WG_INV;
  if (row < sg.rows) {
    const int id = sg.ids ? sg.ids[row] : 1;
    if (!(D.mask_rows && id == 0)) {
      yo = (unsigned)((size_t)row * D.ld_dy * sizeof(float));
      const int T = sg.T >= 1 ? sg.T : 1;
      const size_t xe = sg.x_gather ? (size_t)id * D.ld_x
                        : sg.x_bstride ? (size_t)(row / T) * sg.x_bstride + (size_t)(row % T) * D.ld_x
                                       : (size_t)row * D.ld_x;
      xo = (unsigned)(xe * sizeof(float));
      if (D.K1 > 0) {
        const size_t x1e = sg.x1_bstride ? (size_t)(row / T) * sg.x1_bstride + (size_t)(row % T) * D.ld_x1
                                         : (size_t)row * D.ld_x1;
        x1o = (unsigned)(x1e * sizeof(float));
      }
    }
  }
  tab[v] = yo;
  tab[args.V + v] = xo;
  tab[2 * args.V + v] = x1o;
}

__device__ __forceinline__ f32x4 as_f4(u32x4 v) {
  return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

template <int DBG>
__global__ __launch_bounds__(WG_NT) void gemm_wgrad_cu_kernel(const WgradCuDev args) {
  unsigned long long w_loop = 0, w_bar = 0, w_vm = 0, w_all = 0;
  if constexpr (DBG) w_all = __builtin_amdgcn_s_memtime();
  __shared__ __attribute__((aligned(16))) float Ys[2 * WG_YB];
  __shared__ __attribute__((aligned(16))) float Xs[2 * WG_XB];
  __shared__ __attribute__((aligned(16))) float dummy[4];  // where a wave without the second-source job "stores"

  const CarcaWgradDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nchunks = args.chunk_start[D.nseg];
  const long total = (long)args.nnb * args.nkb * nchunks;
  auto item_at = [&](long wt) -> long {  // weight units -> item index (monotonic: consecutive blocks tile [0, total))
    const long fast_w = args.n_fast * 256;
    return wt <= fast_w ? wt / 256 : args.n_fast + (wt - fast_w) / args.slow_w;
  };
  const long w_begin = min(total, item_at((long)blockIdx.x * args.per_w));
  const long w_end = blockIdx.x + 1 == gridDim.x ? total : min(total, item_at((long)(blockIdx.x + 1) * args.per_w));

  // staging slots of this thread: dY (row tid / 24, float4 tid % 24), X rows tid / 96 + 8 i, float4 tid % 96
  const int y_r = tid / 24, y_c4 = tid - y_r * 24;
  const int x_r = tid / 96, x_c4 = tid - x_r * 96;
  const __amdgpu_buffer_rsrc_t tab_rsrc = carca_rsrc(args.tab);

  for (long w = w_begin; w < w_end;) {
    // ---- one run of chunks inside one output tile and one row segment (so that every buffer resource is a
    // loop invariant: a scalar load inside the loop would wait on lgkmcnt, i.e. on the LDS reads in flight) ----
    const int ot = (int)(w / nchunks), c_begin = (int)(w - (long)ot * nchunks);
    int seg = 0;
#pragma unroll
    for (int q = 1; q < CARCA_MAX_SEGS; ++q)
      if (q < D.nseg && c_begin >= args.chunk_start[q]) seg = q;
    const int c_end = (int)min((long)args.chunk_start[seg + 1], c_begin + (w_end - w));
    w += c_end - c_begin;
    const int kb = ot / args.nnb, nb = ot - kb * args.nnb;  // n block fastest: neighbours share the X columns
    const int n0 = nb * WG_BN, k0 = kb * WG_BK;
    const bool do_db = D.db != nullptr && kb == 0;
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)D.seg[seg].dy, 0, args.y_bytes[seg], 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)D.seg[seg].x, 0, args.x_bytes[seg], 0x00020000);
    const __amdgpu_buffer_rsrc_t x1_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(D.K1 > 0 ? D.seg[seg].x1 : D.seg[seg].x), 0, D.K1 > 0 ? args.x1_bytes[seg] : 0, 0x00020000);
    // second k-source: wave 11 of its host k block (scalar condition); lane -> (row lane / 2, columns 4 (lane & 1) ..)
    const bool s1w = kb == args.src1_kb && wave == WG_NW - 1;
    const int s1_row = lane >> 1, s1_half = lane & 1;
    const bool s1_live = s1w && 4 * s1_half < D.K1;
    const unsigned s1_colb = 16u * s1_half;
    float* const s1_dst0 = s1w ? &Xs[s1_row * WG_XS + WG_BK + 4 * s1_half] : dummy;
    const int s1_bufstride = s1w ? WG_XB : 0;
    unsigned t_x1;  // X1 row offset of the tile to load next
    f32x4 sx;       // X1 piece of tile t+1

    // the thread's 4 columns of X; a slot wholly past K is never fetched (offset 2^31 -> zeros)
    const int xv = k0 + 4 * x_c4;
    const bool x_live = xv < D.K;
    const int x_tab = args.V;
    const unsigned y_colb = (unsigned)((n0 + 4 * y_c4) * sizeof(float));
    const unsigned x_colb = x_live ? (unsigned)(xv * sizeof(float)) : 0u;

    unsigned tcur[5];  // row offsets of the tile to load next (refilled as soon as its loads are issued)
    auto load_tab = [&](int c, unsigned(&t)[5]) {
      const int v0 = c * WG_BR;
      t[4] = __builtin_amdgcn_raw_buffer_load_b32(tab_rsrc, (v0 + y_r) * 4, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)  // a thread whose columns lie past K gets "no row" (bit 31) for every row
        t[i] = __builtin_amdgcn_raw_buffer_load_b32(tab_rsrc, (x_tab + v0 + x_r + 8 * i) * 4, 0, 0) |
               (x_live ? 0u : WG_INV);
      t_x1 = __builtin_amdgcn_raw_buffer_load_b32(tab_rsrc, (2 * args.V + v0 + s1_row) * 4, 0, 0) |
             (s1_live ? 0u : WG_INV);
    };
    f32x4 st[2][5];  // two staging sets: tile t+1 waits for its LDS write while tile t+2 is in flight
    auto load_slot = [&](int i, const unsigned(&t)[5], f32x4& dst) {
      if (i == 4)
        dst = as_f4(__builtin_amdgcn_raw_buffer_load_b128(y_rsrc, t[4] + y_colb, 0, 0));
      else
        dst = as_f4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, t[i] + x_colb, 0, 0));
    };
    auto load_sx = [&](bool on) {
      sx = as_f4(__builtin_amdgcn_raw_buffer_load_b128(x1_rsrc, (on ? t_x1 : WG_INV) + s1_colb, 0, 0));
    };
    auto store_sx = [&](int buf) { *reinterpret_cast<f32x4*>(s1_dst0 + buf * s1_bufstride) = sx; };
    auto store_slot = [&](int i, int buf, const f32x4& v) {
      if (i == 4)
        *reinterpret_cast<f32x4*>(&Ys[buf * WG_YB + y_r * WG_BN + 4 * y_c4]) = v;
      else
        *reinterpret_cast<f32x4*>(&Xs[buf * WG_XB + (x_r + 8 * i) * WG_XS + 4 * x_c4]) = v;
    };

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;  // column (tid % 96) of dY over rows 4 (tid / 96) .. +3 of every chunk (kb == 0 tiles)

    // fragments of one group of 4 MFMA steps (8 rows): xb[j], y[j][t] for step j; A = dY (n index), B = X (k index)
    float xa[4], ya[4][3], xb4[4], yb[4][3];
    const float* yfrag = &Ys[lh * WG_BN + lr];
    const float* xfrag = &Xs[lh * WG_XS + (s1w ? WG_BK : wave * 32) + lr];
    // read `part` (0..7) of group g from LDS buffer buf: parts 0,1 = X pairs, 2..7 = Y pairs of n tile (part-2)/2
    auto read_part = [&](int part, int buf, int g, float(&xq)[4], float(&yq)[4][3]) {
      if (part < 2) {
        const int j = 2 * part;
        xq[j] = xfrag[buf * WG_XB + (8 * g + 2 * j) * WG_XS];
        xq[j + 1] = xfrag[buf * WG_XB + (8 * g + 2 * j + 2) * WG_XS];
      } else {
        const int t = (part - 2) >> 1, j = 2 * ((part - 2) & 1);
        yq[j][t] = yfrag[buf * WG_YB + (8 * g + 2 * j) * WG_BN + 32 * t];
        yq[j + 1][t] = yfrag[buf * WG_YB + (8 * g + 2 * j + 2) * WG_BN + 32 * t];
      }
    };
#define CARCA_PIN() __builtin_amdgcn_sched_barrier(0)
    auto mfma_group = [&](const float(&xq)[4], const float(&yq)[4][3], auto&& aux) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        acc[i % 3] = mfma32(yq[i / 3][i % 3], xq[i / 3], acc[i % 3]);
        CARCA_PIN();
        aux(i);
        CARCA_PIN();
      }
    };

    // ---- prologue: tile 0 -> LDS buffer 0, tile 1 -> staging set 1, row offsets of tile 2 ----------------
    const int nch = c_end - c_begin;
    __syncthreads();  // the previous run's readers are done with both LDS buffers
    load_tab(c_begin, tcur);
#pragma unroll
    for (int i = 0; i < 5; ++i) load_slot(i, tcur, st[0][i]);
    load_sx(true);
#pragma unroll
    for (int i = 0; i < 5; ++i) store_slot(i, 0, st[0][i]);
    store_sx(0);
    if (nch > 1) {
      load_tab(c_begin + 1, tcur);
#pragma unroll
      for (int i = 0; i < 5; ++i) load_slot(i, tcur, st[1][i]);
      load_sx(true);
    }
    if (nch > 2) load_tab(c_begin + 2, tcur);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 8; ++p) read_part(p, 0, 0, xa, ya);

    // one chunk: tile t in LDS buffer CUR, tile t+1 in staging set NXT, tile t+2 gets loaded into set CUR.  The body
    // is branch-free on purpose: with loads under a condition the compiler can no longer count them and falls back
    // to s_waitcnt vmcnt(0) before the LDS writes, which waits for the loads issued a moment ago.  Tiles past the end
    // of the run are "loaded" from the invalid offset (zeros, no memory access) and written to a buffer nobody reads.
    auto step = [&](auto cur_tag, int t) {
      constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
      const int c = c_begin + t;
      unsigned tl[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) tl[i] = t + 2 < nch ? tcur[i] : WG_INV;
      mfma_group(xa, ya, [&](int i) {
        if (i < 8)
          read_part(i, CUR, 1, xb4, yb);
        else
          load_slot(i - 8, tl, st[CUR][i - 8]);
      });
      if constexpr (DBG) {
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything but the four loads just issued
        w_vm += __builtin_amdgcn_s_memtime() - ta;
        CARCA_PIN();
      }
      mfma_group(xb4, yb, [&](int i) {
        if (i < 8)
          read_part(i, CUR, 2, xa, ya);
        else if (i == 8)
          load_slot(4, tl, st[CUR][4]);
        else
          store_slot(i - 9, NXT, st[NXT][i - 9]);
      });
      mfma_group(xa, ya, [&](int i) {
        if (i < 8)
          read_part(i, CUR, 3, xb4, yb);
        else if (i < 10)
          store_slot(i - 5, NXT, st[NXT][i - 5]);
        else if (i == 10) {
          store_sx(NXT);            // tile t+1's piece of the second source, then fetch tile t+2's
          load_sx(t + 2 < nch);
        } else
          load_tab(min(c + 3, c_end - 1), tcur);
      });
      if (do_db) {  // bias gradient: this thread's 4 rows of column tid % 96 (threads 0..767 cover 32 rows x 96)
#pragma unroll
        for (int r = 0; r < 4; ++r) bsum += Ys[CUR * WG_YB + (4 * x_r + r) * WG_BN + x_c4];
      }
      CARCA_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (DBG) {
        const unsigned long long ta = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        w_bar += __builtin_amdgcn_s_memtime() - ta;
      } else {
        __builtin_amdgcn_s_barrier();
      }
      CARCA_PIN();
      mfma_group(xb4, yb, [&](int i) {
        if (i < 8) read_part(i, NXT, 0, xa, ya);
      });
    };
    unsigned long long t_loop = 0;
    if constexpr (DBG) t_loop = __builtin_amdgcn_s_memtime();
    int t = 0;
    for (; t + 1 < nch; t += 2) {
      step(std::integral_constant<int, 0>{}, t);
      step(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nch) step(std::integral_constant<int, 0>{}, t);
    if constexpr (DBG) w_loop += __builtin_amdgcn_s_memtime() - t_loop;
#undef CARCA_PIN

    // ---- flush: D row (= n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (= virtual k) = lane&31 ------------
    const int vk = k0 + wave * 32 + lr;
    const int kcol = s1w ? (lr < D.K1 ? D.K + lr : -1) : vk < D.K ? vk : -1;
    if (kcol >= 0) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < D.N) grad_add(&D.dw[(size_t)n * D.ldw + kcol], acc[tt][r]);
        }
    }
    if (do_db && n0 + x_c4 < D.N) grad_add(&D.db[n0 + x_c4], bsum);
  }
  if constexpr (DBG) {
    unsigned long long* dbg = carca_debug_ptr(args);
    if (dbg && lane == 0) {
      unsigned long long* o = dbg + ((size_t)blockIdx.x * WG_NW + wave) * 4;
      o[0] = __builtin_amdgcn_s_memtime() - w_all;
      o[1] = w_loop;
      o[2] = w_vm;
      o[3] = w_bar;
    }
  }
}

// Lazily grown device workspaces for the row tables: a small ring, one slot per launch, each guarded by an event
// recorded behind its consumer -- consecutive launches never share a table, whatever streams they are issued on.
constexpr int TAB_RING = 4;
unsigned* g_tab[TAB_RING] = {nullptr};
size_t g_tab_elems[TAB_RING] = {0};
hipEvent_t g_tab_ev[TAB_RING];
bool g_tab_used[TAB_RING] = {false};
bool g_tab_init = false;
int g_tab_next = 0;
int g_num_cus = 0;

}  // namespace

// CARCA_OK: launched.  1: shape not suited / operands too large for 31-bit offsets -> caller uses the tiled kernel.
int carca_wgrad_cu_try(const CarcaWgradDesc* desc, hipStream_t stream) {
  WgradCuDev g{};
  g.d = *desc;
  int chunks = 0;
  const uint64_t lim = 1ull << 31;  // bytes
  for (int s = 0; s < desc->nseg; ++s) {
    const CarcaWgradSeg& sg = desc->seg[s];
    const int T = sg.T >= 1 ? sg.T : 1;
    if (g.d.seg[s].T < 1) g.d.seg[s].T = 1;
    const uint64_t ub = (uint64_t)((sg.rows - 1) / T), ut = (uint64_t)(T - 1);
    uint64_t xe;
    if (sg.x_gather) {
      if (sg.x_gather <= 1) return 1;  // table size unknown
      xe = (uint64_t)sg.x_gather * desc->ld_x;
    } else {
      xe = (sg.x_bstride ? ub * sg.x_bstride + ut * desc->ld_x : (uint64_t)(sg.rows - 1) * desc->ld_x) + desc->K;
    }
    const uint64_t ye = (uint64_t)(sg.rows - 1) * desc->ld_dy + desc->N;
    if (desc->K1 > 8) return 1;
    const uint64_t x1e = desc->K1 == 0 ? 0
                         : (sg.x1_bstride ? ub * sg.x1_bstride + ut * desc->ld_x1 : (uint64_t)(sg.rows - 1) * desc->ld_x1) +
                               desc->K1;
    if (x1e * 4 >= lim) return 1;
    g.x1_bytes[s] = (unsigned)(x1e * 4);
    if (xe * 4 >= lim || ye * 4 >= lim) return 1;
    g.x_bytes[s] = (unsigned)(xe * 4);
    g.y_bytes[s] = (unsigned)(ye * 4);
    g.chunk_start[s] = chunks;
    chunks += (sg.rows + WG_BR - 1) / WG_BR;
  }
  g.chunk_start[desc->nseg] = chunks;
  g.nnb = (desc->N + WG_BN - 1) / WG_BN;
  g.nkb = (desc->K + WG_BK - 1) / WG_BK;
  g.src1_kb = -1;
  if (desc->K1 > 0) {  // wave 11's own columns must be dead in the host block: else one more, otherwise empty, block
    g.src1_kb = desc->K - (g.nkb - 1) * WG_BK <= WG_BK - 32 ? g.nkb - 1 : g.nkb;
    g.nkb = g.src1_kb + 1;
  }
  const long total = (long)g.nnb * g.nkb * chunks;
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    g_num_cus = prop.multiProcessorCount;
  }
  // worth it only when every CU gets a long run of chunks (pipeline fill + two flushes per block are overhead)
  const bool forced = carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 2 || carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 3;
  // (24 chunk-tiles per CU: the d x F product of the re-associated embedding backward, 26 per CU at C2, runs 2x faster
  // here than on the tile kernel; the joint-embedding dW, 5 per CU, does not)
  if (!forced && (total < (long)g_num_cus * 24 || chunks < 8)) return 1;
  const long n_slow = (desc->K1 > 0 && g.src1_kb >= 0 && g.src1_kb == g.nkb - 1) ? (long)g.nnb * chunks : 0;
  g.n_fast = total - n_slow;
  g.slow_w = carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 13 ? 256 : 264;  // (variant 13: equal item counts -- A/B switch)
  const long total_w = g.n_fast * 256 + n_slow * g.slow_w;
  g.per_w = (total_w + g_num_cus - 1) / g_num_cus;
  const int grid = (int)((total_w + g.per_w - 1) / g.per_w);
  g.V = chunks * WG_BR;
  if (!g_tab_init) {
    for (int i = 0; i < TAB_RING; ++i) (void)hipEventCreateWithFlags(&g_tab_ev[i], hipEventDisableTiming);
    g_tab_init = true;
  }
  int slot = -1;
  unsigned* tab;
  if (carca_stream_capturing(stream)) {  // hipGraph capture: a table of the graph's own, no ring slot, no guard event
    tab = (unsigned*)carca_capture_alloc((size_t)3 * g.V * sizeof(unsigned), false, nullptr);
    if (!tab) return 1;
  } else {
    slot = g_tab_next;
    g_tab_next = (g_tab_next + 1) % TAB_RING;
    if (g_tab_used[slot]) (void)hipEventSynchronize(g_tab_ev[slot]);  // its last consumer: normally long finished
    if ((size_t)3 * g.V > g_tab_elems[slot]) {
      if (g_tab[slot]) (void)hipFree(g_tab[slot]);
      g_tab[slot] = nullptr;
      g_tab_elems[slot] = (size_t)3 * g.V * 2;
      if (hipMalloc(&g_tab[slot], g_tab_elems[slot] * sizeof(unsigned)) != hipSuccess) {
        g_tab_elems[slot] = 0;
        return 1;
      }
    }
    tab = g_tab[slot];
  }
  g.tab = tab;
  hipLaunchKernelGGL(wgrad_rowtab_kernel, dim3((g.V + 255) / 256), dim3(256), 0, stream, g, tab);
  g.dbg = carca_debug_buffer();
  if (carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 3 && g.dbg)
    hipLaunchKernelGGL(gemm_wgrad_cu_kernel<1>, dim3(grid), dim3(WG_NT), 0, stream, g);
  else
    hipLaunchKernelGGL(gemm_wgrad_cu_kernel<0>, dim3(grid), dim3(WG_NT), 0, stream, g);
  if (slot >= 0) {
    (void)hipEventRecord(g_tab_ev[slot], stream);
    g_tab_used[slot] = true;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    carca_set_error("HIP launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return CARCA_OK;
}
