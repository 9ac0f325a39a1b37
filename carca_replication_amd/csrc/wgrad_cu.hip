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
#include <algorithm>
#include <type_traits>
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

constexpr int WG_BN = 96, WG_BK = 384, WG_BR = 32, WG_NW = 12, WG_NT = 768;
constexpr int WG_XS = WG_BK + 32;                       // X tile row stride (floats)
constexpr int WG_YB = WG_BR * WG_BN, WG_XB = WG_BR * WG_XS;  // floats per LDS buffer
constexpr unsigned WG_INV = 0x80000000u;                // row offset that reads as zeros (num_records <= 2^31)

// What depends on HOW MANY rows take part is decided on the device: rows whose ids == 0 (mask_rows: the e * mask of
// carca.py:94 seen from the backward side) contribute nothing, and the row-table kernel leaves them OUT of the table instead
// of listing them as "reads as zeros" -- with left-padded profiles of U{3..50} items 47 % of a training
// batch's rows are pads, and the chunks they filled were 47 % of this kernel's MFMA work.  The compacted chunk counts, and
// with them the stream-K ranges, are written by that kernel into a WgPlan next to the table; the host sizes everything by
// the uncompacted upper bounds.
struct WgPlan {
  int chunk_start[CARCA_MAX_SEGS + 1];  // compacted 32-row chunks per segment, prefix sums
  int nchunks, ngroups, slow_w, pad;
  long total, n_fast, per_w;
};

struct WgradCuDev {
  CarcaWgradDesc d;
  int chunk_start[CARCA_MAX_SEGS + 1];  // UPPER BOUNDS (every row kept): where a segment's entries start in the table
  unsigned y_bytes[CARCA_MAX_SEGS], x_bytes[CARCA_MAX_SEGS], x1_bytes[CARCA_MAX_SEGS];  // buffer extents
  const unsigned* tab;  // [3][V]: byte offsets of dY / X / X1 rows inside their segment, V = 32 * chunks
  int V, nnb, nkb;
  // stream-K ranges in units of 1/256 item: items of the k block that also carries the second k-source (the LAST k block:
  // k blocks are the outermost index) weigh slow_w / 256 -- its chunks run ~5 % longer (one more load per lane), and with
  // equal item counts the workgroups that sit entirely inside it finished 5.4 % behind the rest (measured, C2: block times
  // 1.134 .. 1.417 M cycles; 1.295 .. 1.364 M with a weight of 270 / 256, those workgroups then 2 % early: 264)
  long per_w, n_fast;
  int slow_w;
  // Workgroups work in GROUPS of nnb: the members of a group walk the same (k block, chunk) range, one n block each, so
  // that an X chunk (32 rows x 384 columns, 49 KB) is fetched from HBM once and found in the L2 by the other members --
  // with consecutive workgroups on consecutive item ranges the nnb readers of a chunk came by at unrelated times and the
  // kernel read X nnb times over (PMC: 2.13 GB per launch at C2 against 357 MB algorithmic).  A group's members sit on
  // ONE XCD (workgroup b runs on XCD b % nxcd): gpx groups per XCD, the slots left over form groups across XCDs (they
  // still meet in the memory-side cache), workgroups beyond ngroups x nnb have no work.
  int nxcd, gpx, ngroups;
  // Output tiles are SUMMED WITHOUT ATOMICS: a workgroup that leaves a k block stores its 96 x 384 partial tile (plain,
  // 256 contiguous bytes per wave store) into its slot of `part`, and wgrad_reduce_kernel, launched behind this one, adds
  // each tile's partials in group order into dW.  (36,864 fp32 atomics per flush and workgroup, all workgroups at once
  // when the kernel ends: the stamps put ~8 % of the kernel outside its chunk loop.  Tried first: the workgroup that
  // arrives LAST at a tile reduces it inside this kernel -- agent-scope fences flush / invalidate the whole L2 of an XCD
  // under the workgroups still streaming through it (kernel 1.41 M -> 1.75 M cycles), and with agent-scope stores
  // instead the reducing workgroups became the tail, 1.47 M -> 1.54 M cycles for the slowest.)
  // Fixed summation order: dW is bit-reproducible run to run without the deterministic mode's shadow buffer.
  float* part;    // [ngroups][slots_pg][nnb][96 x 384], register order: element (tt, r) of thread tid at (16 tt + r) 768 + tid
  int slots_pg;   // k blocks a group's range can touch (2 unless the product is short); 0 = flush with fp32 atomics (A/B switch,
                  // tuning variant 14, and products whose tiles get so many partials that the atomics are cheaper)
  long* gbegin;   // [ngroups + 1] first unit of every group (written by the row-table kernel: the reduce kernel reads
  int* klo;       // [nkb][2] first / last group with units in the k block     them instead of redoing 64-bit divisions)
  int src1_kb;  // k block that also carries the second k-source's columns, or -1
  unsigned long long* dbg;    // phase stamps of a diagnostic run (tuning variant 3)
  WgPlan* plan;   // device: written by wgrad_rowtab_kernel, read by the two kernels behind it
  int compact;    // 0: masked rows stay in the table as "reads as zeros" (tuning variant 22, A/B; the round-3 behaviour)
  int* err;       // carca_kernel_error_word(): 3 = a group's range met more k blocks than slots_pg (launcher bug)
};
__device__ __forceinline__ unsigned long long* carca_debug_ptr(const WgradCuDev& a) { return a.dbg; }

// the groups' unit ranges (stream-K over (k block, chunk) units in weight units of 1/256 unit, see WgradCuDev)
struct WgRanges {
  long total, n_fast, per_w;
  int slow_w, ngroups;
  __device__ long item_at(long wt) const {
    const long fast_w = n_fast * 256;
    return wt <= fast_w ? wt / 256 : n_fast + (wt - fast_w) / slow_w;
  }
  __device__ long begin_of(int g) const { return g >= ngroups ? total : min(total, item_at((long)g * per_w)); }
  __device__ int group_of(long u) const {  // the group whose range holds unit u
    const long wt = u <= n_fast ? u * 256 : n_fast * 256 + (u - n_fast) * slow_w;
    int g = (int)min((long)ngroups - 1, wt / per_w);
    while (g > 0 && begin_of(g) > u) --g;
    while (g + 1 < ngroups && begin_of(g + 1) <= u) ++g;
    return g;
  }
};

// ONE 1024-thread block.  Per segment the rows go by in super-tiles of 8192 (row = base + 1024 j + tid, j < 8): all of a
// thread's ids are requested at once, every (j, wave) counts its rows that take part with a ballot, ONE barrier later each
// thread knows where its rows land -- the three byte offsets of a kept row are written back to back from the segment's
// first chunk on, the last chunk is padded with "no row" entries; then the plan and the groups' unit ranges.
// (19,200 rows at C2: three super-tiles, ~6 us; a first version with a contiguous run of rows per thread took 27 us: two
// passes of dependent id loads.)
__global__ __launch_bounds__(1024) void wgrad_rowtab_kernel(const WgradCuDev args, unsigned* tab) {
  constexpr int PER = 8;
  __shared__ int wcnt[PER][16];
  __shared__ WgPlan plan;
  const CarcaWgradDesc& D = args.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int chunk0 = 0;
  if (tid == 0) plan.chunk_start[0] = 0;
  for (int s = 0; s < D.nseg; ++s) {
    const CarcaWgradSeg sg = D.seg[s];
    const int T = sg.T >= 1 ? sg.T : 1;
    int kept_seg = 0;  // rows of this segment placed so far (block-uniform)
    for (int base = 0; base < sg.rows; base += 1024 * PER) {
      int id[PER];
      bool in[PER], keep[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        in[j] = base + 1024 * j + tid < sg.rows;
        id[j] = 1;
      }
      if (sg.ids) {
        // (UNCONDITIONAL loads from clamped rows, all eight requested before anything looks at them: under `in[j]` each
        // load sat in its own branch with its own wait, eight round trips per super-tile -- the kernel measured 28 us)
#pragma unroll
        for (int j = 0; j < PER; ++j) id[j] = sg.ids[min(base + 1024 * j + tid, sg.rows - 1)];
      }
      unsigned long long bal[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const bool m = D.mask_rows && sg.ids && id[j] == 0;
        keep[j] = in[j] && !(args.compact && m);
        bal[j] = __ballot(keep[j]);
        if (lane == 0) wcnt[j][wave] = __popcll(bal[j]);
      }
      __syncthreads();
      int before = kept_seg, total = 0;  // rows placed ahead of (j, wave), rows of the whole super-tile
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        int upto = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
          const int c = wcnt[j][w];
          upto += w < wave ? c : 0;
          all += c;
        }
        if (keep[j]) {
          const int row = base + 1024 * j + tid;
          const int v = chunk0 * WG_BR + before + total + upto + __popcll(bal[j] & ((1ull << lane) - 1));
          const bool m = D.mask_rows && sg.ids && id[j] == 0;
          unsigned yo = WG_INV, xo = WG_INV, x1o = WG_INV;
          if (!m) {
            yo = (unsigned)((size_t)row * D.ld_dy * sizeof(float));
            const size_t xe = sg.x_gather ? (size_t)id[j] * D.ld_x
                              : sg.x_bstride ? (size_t)(row / T) * sg.x_bstride + (size_t)(row % T) * D.ld_x
                                             : (size_t)row * D.ld_x;
            xo = (unsigned)(xe * sizeof(float));
            if (D.K1 > 0) {
              const size_t x1e = sg.x1_bstride ? (size_t)(row / T) * sg.x1_bstride + (size_t)(row % T) * D.ld_x1
                                               : (size_t)row * D.ld_x1;
              x1o = (unsigned)(x1e * sizeof(float));
            }
          }
          tab[v] = yo;
          tab[args.V + v] = xo;
          tab[2 * args.V + v] = x1o;
        }
        total += all;
      }
      kept_seg += total;
      __syncthreads();  // (wcnt is rewritten by the next super-tile / segment)
    }
    const int chunks = (kept_seg + WG_BR - 1) / WG_BR;
    for (int p = chunk0 * WG_BR + kept_seg + tid; p < (chunk0 + chunks) * WG_BR; p += 1024) {  // the last chunk's tail
      tab[p] = WG_INV;
      tab[args.V + p] = WG_INV;
      tab[2 * args.V + p] = WG_INV;
    }
    chunk0 += chunks;
    if (tid == 0) plan.chunk_start[s + 1] = chunk0;
  }
  __syncthreads();
  if (tid == 0) {
    for (int s = D.nseg + 1; s <= CARCA_MAX_SEGS; ++s) plan.chunk_start[s] = chunk0;
    const long total = (long)args.nkb * chunk0;
    const long n_slow = (D.K1 > 0 && args.src1_kb >= 0 && args.src1_kb == args.nkb - 1) ? (long)chunk0 : 0;
    plan.nchunks = chunk0;
    plan.total = total;
    plan.n_fast = total - n_slow;
    plan.slow_w = args.slow_w;
    const long total_w = plan.n_fast * 256 + n_slow * args.slow_w;
    plan.per_w = max(1l, (total_w + args.ngroups - 1) / args.ngroups);  // (args.ngroups: the groups the chip can host)
    plan.ngroups = (int)((total_w + plan.per_w - 1) / plan.per_w);      // (a short product: fewer groups)
    plan.pad = 0;
    *args.plan = plan;
  }
  __syncthreads();
  const WgRanges rg{plan.total, plan.n_fast, plan.per_w, plan.slow_w, plan.ngroups};
  for (int g = tid; g <= args.ngroups; g += 1024) args.gbegin[g] = rg.begin_of(g);
  for (int kb = tid; kb < args.nkb; kb += 1024) {
    const bool any = plan.ngroups > 0 && plan.nchunks > 0;
    args.klo[2 * kb] = any ? rg.group_of((long)kb * plan.nchunks) : 0;
    args.klo[2 * kb + 1] = any ? rg.group_of((long)(kb + 1) * plan.nchunks - 1) : -1;
  }
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
  // the plan of the row-table kernel, once, into scalars (a scalar load inside the loops below would wait on lgkmcnt,
  // i.e. on the LDS reads in flight)
  const WgPlan P = *args.plan;
  int cs[CARCA_MAX_SEGS + 1];
#pragma unroll
  for (int q = 0; q <= CARCA_MAX_SEGS; ++q) cs[q] = __builtin_amdgcn_readfirstlane(P.chunk_start[q]);
  const int nchunks = __builtin_amdgcn_readfirstlane(P.nchunks);
  const int ngroups = __builtin_amdgcn_readfirstlane(P.ngroups);
  const long total = P.total;  // (k block, chunk) units of a group; its members take an n block each
  const WgRanges rg{total, P.n_fast, P.per_w, P.slow_w, ngroups};
  int grp, nb;
  {
    const int per_xcd = (int)gridDim.x / args.nxcd, x = (int)blockIdx.x % args.nxcd, sl = (int)blockIdx.x / args.nxcd;
    const int in_groups = args.gpx * args.nnb;  // slots of an XCD that belong to its own groups
    if (sl < in_groups) {
      grp = x * args.gpx + sl / args.nnb;
      nb = sl % args.nnb;
    } else {
      const int l = x * (per_xcd - in_groups) + (sl - in_groups);
      grp = args.nxcd * args.gpx + l / args.nnb;
      nb = l % args.nnb;
    }
  }
  if (grp >= ngroups || nchunks == 0) return;
  const long w_begin = rg.begin_of(grp);
  const long w_end = rg.begin_of(grp + 1);

  // staging slots of this thread: dY (row tid / 24, float4 tid % 24), X rows tid / 96 + 8 i, float4 tid % 96
  const int y_r = tid / 24, y_c4 = tid - y_r * 24;
  const int x_r = tid / 96, x_c4 = tid - x_r * 96;
  const __amdgpu_buffer_rsrc_t tab_rsrc = carca_rsrc(args.tab);

  f32x16 acc[3];  // the output tile of the k block in progress: kept across the runs (row segments) inside it
  int acc_kb = -1;
  for (long w = w_begin; w < w_end;) {
    // ---- one run of chunks inside one output tile and one row segment (so that every buffer resource is a
    // loop invariant: a scalar load inside the loop would wait on lgkmcnt, i.e. on the LDS reads in flight) ----
    const int kb = (int)(w / nchunks), c_begin = (int)(w - (long)kb * nchunks);
    int seg = 0;
#pragma unroll
    for (int q = 1; q < CARCA_MAX_SEGS; ++q)
      if (q < D.nseg && c_begin >= cs[q]) seg = q;
    const int c_end = (int)min((long)cs[seg + 1], c_begin + (w_end - w));
    w += c_end - c_begin;
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

    if (kb != acc_kb) {
      acc_kb = kb;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
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

    // ---- flush, once per (group, k block): D row (= n) = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col (= virtual k) = lane&31
    if ((w >= w_end || (int)(w / nchunks) != kb) && args.slots_pg == 0) {
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
    } else if (w >= w_end || (int)(w / nchunks) != kb) {
      // the partial tile -> this group's slot for the k block
      const int kb_first = (int)(w_begin / nchunks);
      const size_t tile_fl = (size_t)WG_BN * WG_BK;
      // (slots_pg is the launcher's PROOF-ONLY bound -- the row table's plan is made on the device after it was sized: a
      // range that met more k blocks than that would write into the next group's slots.  Say so instead, and drop the tile)
      const bool slot_ok = kb - kb_first < args.slots_pg;
      if (!slot_ok && tid == 0 && args.err) __hip_atomic_store(args.err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      float* const mine = args.part + ((size_t)(grp * args.slots_pg + (slot_ok ? kb - kb_first : 0)) * args.nnb + nb) * tile_fl;
      if (slot_ok) {
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[(tt * 16 + r) * WG_NT + tid] = acc[tt][r];  // (register order)
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

// dW tile (kb, nb) += its groups' partial tiles, in group order.  grid (nkb x nnb, 36): a thread takes four consecutive
// floats of the tile in register order (element e = (tt, r) of the threads tid .. tid + 3 of gemm_wgrad_cu_kernel: the
// same n, four consecutive k) -- one 16-byte load per partial.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradCuDev args) {
  const CarcaWgradDesc& D = args.d;
  const int nchunks = args.plan->nchunks;
  const int kb = (int)blockIdx.x / args.nnb, nb = (int)blockIdx.x - kb * args.nnb;
  const int j4 = (int)blockIdx.y * 256 + threadIdx.x;  // float4 index inside the tile, 0 .. 9215
  const int e = (4 * j4) / WG_NT, tid = 4 * j4 - e * WG_NT;
  const int lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const bool s1w = kb == args.src1_kb && wave == WG_NW - 1;
  const long u0 = (long)kb * nchunks, u1 = u0 + nchunks;
  const int g_lo = args.klo[2 * kb], g_hi = args.klo[2 * kb + 1];
  const size_t tile_fl = (size_t)WG_BN * WG_BK;
  // four groups' partials in flight at a time (a plain loop waits for every load before the next is issued), added in group
  // order; a group without units in this k block contributes its slot 0 times 0
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int g0 = g_lo; g0 <= g_hi; g0 += 4) {
    f32x4 v[4];
    float live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g2 = min(g0 + i, g_hi);
      const long b2 = args.gbegin[g2];
      const bool has = g0 + i <= g_hi && min(args.gbegin[g2 + 1], u1) > max(b2, u0);
      const int slot_kb = has ? kb - (int)(b2 / nchunks) : 0;
      const float* src = args.part + ((size_t)(g2 * args.slots_pg + slot_kb) * args.nnb + nb) * tile_fl;
      v[i] = *reinterpret_cast<const f32x4*>(src + 4 * (size_t)j4);
      live[i] = has ? 1.f : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (live[i] != 0.f) sum += v[i];
  }
  const int tt = e >> 4, r = e & 15;
  const int n = nb * WG_BN + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
  if (n >= D.N) return;
  float* row = D.dw + (size_t)n * D.ldw;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vk = kb * WG_BK + wave * 32 + lr + i;
    const int kcol = s1w ? (lr + i < D.K1 ? D.K + lr + i : -1) : vk < D.K ? vk : -1;
    if (kcol >= 0) row[kcol] += sum[i];
  }
}

// The row table, the group ranges and the partial tiles of a launch are stream scratch (carca_common.h): written by this
// launch's first kernel, read by its second and third, the next launch on the stream ordered behind all three.
int g_num_cus = 0;

}  // namespace

// The row table depends on the ids alone (which rows take part, where their operands start): a caller that has other work to
// issue first -- carca_embed_bwd: d joint_embed, d [z ; q], the scatter-add -- FORKS a second stream off its own at entry
// (carca_wgrad_table_fork), the next carca_wgrad_cu_try of this thread launches wgrad_rowtab_kernel THERE (17-26 us of one
// block's latency chain beside the caller's launches instead of in front of the big kernel) and makes `stream` wait for it;
// carca_wgrad_table_join closes a fork that no launch used.  Inside a hipGraph capture the two event edges become graph
// edges.  The table's memory is ordered by `stream`: the fork's event sits behind every earlier launch that read it.
namespace {
struct TableFork {
  hipStream_t ts = nullptr;
  hipEvent_t ea = nullptr, eb = nullptr;
  bool open = false;
};
thread_local TableFork g_fork;
}  // namespace
int carca_wgrad_table_fork(hipStream_t stream, hipStream_t table_stream) {
  if (!table_stream || table_stream == stream) return CARCA_OK;
  TableFork& f = g_fork;
  // (a fresh pair of events per fork, destroyed at the join -- what torch's wait_stream does: an event object recorded a
  // second time from another stream of the same capture is one more thing this runtime's capture has not been seen to survive)
  f.ea = f.eb = nullptr;
  if (hipEventCreateWithFlags(&f.ea, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&f.eb, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (f.ea) (void)hipEventDestroy(f.ea);
    f.ea = f.eb = nullptr;
    return CARCA_OK;  // (no events: no fork -- the table is built on `stream` as before)
  }
  if (hipEventRecord(f.ea, stream) != hipSuccess || hipStreamWaitEvent(table_stream, f.ea, 0) != hipSuccess) {
    carca_set_error("wgrad_table_fork: cannot fork the row-table stream: %s", hipGetErrorString(hipGetLastError()));
    return CARCA_ERR_BADARG;
  }
  f.ts = table_stream;
  f.open = true;
  return CARCA_OK;
}
namespace {
__global__ void table_fork_noop_kernel() {}
}  // namespace
int carca_wgrad_table_join(hipStream_t stream, bool used) {
  TableFork& f = g_fork;
  if (!f.open) return CARCA_OK;
  f.open = false;
  // (a fork nothing was launched on: give the branch a node before it is joined -- see carca_wgrad_cu_suited)
  if (!used) hipLaunchKernelGGL(table_fork_noop_kernel, dim3(1), dim3(64), 0, f.ts);
  const bool ok = hipEventRecord(f.eb, f.ts) == hipSuccess && hipStreamWaitEvent(stream, f.eb, 0) == hipSuccess;
  (void)hipEventDestroy(f.ea);
  (void)hipEventDestroy(f.eb);
  f.ea = f.eb = nullptr;
  if (!ok) {
    carca_set_error("wgrad_table_join: cannot join the row-table stream: %s", hipGetErrorString(hipGetLastError()));
    return CARCA_ERR_BADARG;
  }
  return CARCA_OK;
}

// CARCA_OK: launched.  1: shape not suited / operands too large for 31-bit offsets -> caller uses the tiled kernel.
static int wgrad_cu_try(const CarcaWgradDesc* desc, hipStream_t stream, bool dry);
int carca_wgrad_cu_try(const CarcaWgradDesc* desc, hipStream_t stream) { return wgrad_cu_try(desc, stream, false); }
// would carca_gemm_wgrad send this product to the persistent kernel?  (no launch, no allocation: carca_embed_bwd asks
// before it forks a stream for the row table -- a fork that nothing is launched on is an EMPTY branch of a capture, and
// hipStreamEndCapture of this runtime crashes on one)
bool carca_wgrad_cu_suited(const CarcaWgradDesc* desc) {
  const int variant = carca_tuning(CARCA_TUNE_GEMM_VARIANT);
  return variant != 4 && variant != 5 && wgrad_cu_try(desc, nullptr, true) == CARCA_OK;
}
static int wgrad_cu_try(const CarcaWgradDesc* desc, hipStream_t stream, bool dry) {
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
  // (tuning key 10: the caller's CU budget for this launch -- the other CUs belong to a concurrent stream)
  int ncu = g_num_cus;
  if (carca_tuning(CARCA_TUNE_CU_CAP) > 0 && carca_tuning(CARCA_TUNE_CU_CAP) < ncu) ncu = std::max(8, carca_tuning(CARCA_TUNE_CU_CAP) / 8 * 8);
  // worth it only when every CU gets a long run of chunks (pipeline fill + two flushes per block are overhead)
  const bool forced = carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 2 || carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 3;
  // (24 chunk-tiles per CU: the d x F product of the re-associated embedding backward, 26 per CU at C2, runs 2x faster
  // here than on the tile kernel; the joint-embedding dW, 5 per CU, does not)
  if (!forced && (total < (long)ncu * 24 || chunks < 8)) return 1;
  // groups of nnb workgroups (see WgradCuDev): per XCD as many whole groups as fit, the left-over slots group across XCDs
  // (one n block: nothing to share -- consecutive workgroups take consecutive ranges, as before the groups; tuning variant 16
  // forces that numbering for any nnb: A/B switch)
  g.nxcd = (ncu % 8 == 0 && g.nnb <= ncu / 8 && g.nnb > 1 && carca_tuning(CARCA_TUNE_GEMM_VARIANT) != 16) ? 8 : 1;
  const int per_xcd = ncu / g.nxcd;
  g.gpx = per_xcd / g.nnb;
  g.ngroups = g.nxcd * g.gpx + (g.nxcd * (per_xcd - g.gpx * g.nnb)) / g.nnb;
  if (g.ngroups < 1) return 1;  // (more n blocks than CUs: the tile kernel)
  if (dry) return CARCA_OK;
  // (g.ngroups stays the number of groups the chip can HOST; how many of them get units, and which, is the row-table
  // kernel's decision once it has counted the rows that take part: WgPlan)
  g.slow_w = carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 13 ? 256 : 264;  // (variant 13: equal item counts -- A/B switch)
  g.compact = carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 22 ? 0 : 1;       // (variant 22: masked rows stay in the table -- A/B switch)
  const int grid = ncu;
  g.V = chunks * WG_BR;
  // k blocks a group's range can touch, for ANY number of chunks per k block c >= 1: a range holds at most
  // ceil(nkb c / ngroups) + 1 units, which reach into at most floor(nkb / ngroups) + 2 k blocks
  g.slots_pg = g.nkb / g.ngroups + 2;
  g.err = carca_kernel_error_word();
  // Partials pay when a tile receives FEW of them: every partial is 147 KB written and read back (C2's feats_embed: 5.6 per
  // tile, 45 MB, ~14 us of reduce kernel against ~60 us of atomics).  A product of few tiles cut over all the groups (the
  // d x F product of the re-associated backward: 11 tiles, 23 partials each) keeps the atomics.
  const long n_tiles = (long)g.nkb * g.nnb, n_parts = (long)g.ngroups * g.nnb;
  if (carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 14 || n_parts > 8 * n_tiles) g.slots_pg = 0;
  const size_t part_floats = (size_t)g.ngroups * g.slots_pg * g.nnb * WG_BN * WG_BK;
  const size_t plan_ints = (sizeof(WgPlan) + 15) / 16 * 4;
  const size_t cnt_ints = ((size_t)2 * (g.ngroups + 1) + 2 * g.nkb + 5) / 4 * 4 + plan_ints;  // gbegin (8-byte entries) | klo | plan
  const size_t tab_bytes = ((size_t)3 * g.V + cnt_ints) * sizeof(unsigned) + part_floats * sizeof(float);  // row table | group ranges | partial tiles
  unsigned* tab = (unsigned*)(carca_stream_capturing(stream) ? carca_capture_alloc(stream, tab_bytes, false, nullptr)
                                                              : carca_stream_scratch(stream, CARCA_SCRATCH_WTAB, tab_bytes));
  if (!tab) return 1;
  g.tab = tab;
  {
    unsigned* aux = tab + (size_t)3 * g.V;  // (V is a multiple of 32 and the buffer comes from hipMalloc: 16-byte aligned)
    g.gbegin = (long*)aux;
    g.klo = (int*)(aux + 2 * (size_t)(g.ngroups + 1));
    g.plan = (WgPlan*)(aux + cnt_ints - plan_ints);  // (16-byte aligned: everything in front of it is whole 16-byte groups)
    g.part = (float*)(aux + cnt_ints);
  }
  if (g_fork.open) {  // (the caller forked a stream for the table at its entry: see carca_wgrad_table_fork)
    hipLaunchKernelGGL(wgrad_rowtab_kernel, dim3(1), dim3(1024), 0, g_fork.ts, g, tab);
    if (int rc = carca_wgrad_table_join(stream, true)) return rc;
  } else {
    hipLaunchKernelGGL(wgrad_rowtab_kernel, dim3(1), dim3(1024), 0, stream, g, tab);
  }
  g.dbg = carca_debug_buffer();
  if (carca_tuning(CARCA_TUNE_GEMM_VARIANT) == 3 && g.dbg)
    hipLaunchKernelGGL(gemm_wgrad_cu_kernel<1>, dim3(grid), dim3(WG_NT), 0, stream, g);
  else
    hipLaunchKernelGGL(gemm_wgrad_cu_kernel<0>, dim3(grid), dim3(WG_NT), 0, stream, g);
  if (g.slots_pg > 0)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g.nkb * g.nnb, WG_BN * WG_BK / 1024), dim3(256), 0, stream, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    carca_set_error("HIP launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return CARCA_OK;
}
