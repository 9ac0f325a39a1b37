// Shared device helpers for the CARCA gfx950 kernels (wave64, fp32-input MFMA).
//
// MFMA operand maps used everywhere in this directory (MI355X guide, "FP32-input MFMA"):
//   v_mfma_f32_16x16x4_f32 : lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
//                            D: col j = l&15, row i = 4*(l>>4) + reg, reg in [0,4)
//   v_mfma_f32_32x32x2_f32 : lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
//                            D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5), reg in [0,16)
// Every product here is written D[m][n] = sum_k A[m][k] * Bt[n][k] with BOTH operands read
// k-contiguous, 16 B per lane: for a group of KG consecutive k (16 for 16x16x4, 8 for 32x32x2)
// lane (row, q) reads k = KG*kg + 4*q .. +3 and MFMA step s of the group contracts k = KG*kg + 4*q + s.
// The order of k inside a group is therefore permuted identically for A and Bt, which a sum allows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// 16-byte vector with 4-byte alignment: hipcc still emits global_load_dwordx4 for it
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

#define CARCA_WAVE 64

// Global loads go through BUFFER instructions: wave-uniform base (scalar resource) + 32-bit per-lane element offset.
// Measured on the feature GEMM (tools/stamp_gemm.py): a global_load_dwordx4 with 64-bit per-lane addresses costs the
// SIMD ~130 cycles of MFMA issue per instruction, the same load as buffer_load_dwordx4 ~7.  `base` MUST be uniform
// over the wave (kernel argument, or derived from blockIdx / a readfirstlane'd wave index); offsets are elements.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t carca_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ f32x4 gload4(const float* base, int elem_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(base), elem_off * 4, 0, 0);
  f32x4 f = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  return f;
}
// the same load with the wave-uniform part of the offset kept apart (scalar offset operand of the instruction): one
// lane-offset register serves every load of a fragment sweep instead of one pre-added register per load
__device__ __forceinline__ f32x4 gload4s(const float* base, int lane_elem_off, int uniform_elem_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(carca_rsrc(base), lane_elem_off * 4, uniform_elem_off * 4, 0);
  f32x4 f = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
  return f;
}
__device__ __forceinline__ float gload1(const float* base, int elem_off) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(carca_rsrc(base), elem_off * 4, 0, 0));
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 4 MFMA steps of one 16-k group: a and b hold the lane's four k values
__device__ __forceinline__ f32x4 mfma16_group(f32x4 a, f32x4 b, f32x4 c) {
  c = mfma16(a[0], b[0], c);
  c = mfma16(a[1], b[1], c);
  c = mfma16(a[2], b[2], c);
  c = mfma16(a[3], b[3], c);
  return c;
}

// ---- cross-lane reductions without LDS traffic (__shfl_xor compiles to ds_bpermute: ~100+ cycles a hop) ----
// all-reduce inside each 16-lane row: four DPP row rotations
#define CARCA_DPP_ROR(v, n) __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float row16_sum(float v) {
  v += CARCA_DPP_ROR(v, 8);
  v += CARCA_DPP_ROR(v, 4);
  v += CARCA_DPP_ROR(v, 2);
  v += CARCA_DPP_ROR(v, 1);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, CARCA_DPP_ROR(v, 8));
  v = fmaxf(v, CARCA_DPP_ROR(v, 4));
  v = fmaxf(v, CARCA_DPP_ROR(v, 2));
  v = fmaxf(v, CARCA_DPP_ROR(v, 1));
  return v;
}
// all-reduce over the four lanes that share l&15 (lanes l, l^16, l^32, l^48): gfx950's row / half swaps.
// v_permlane16_swap(a, a) leaves {rows 0,0,2,2} and {rows 1,1,3,3}; v_permlane32_swap(a, a) {lo,lo} and {hi,hi}.
__device__ __forceinline__ float quad4_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float quad4_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) { return quad4_sum(row16_sum(v)); }
// all-reduce inside each 32-lane half of the wave
__device__ __forceinline__ float half32_sum(float v) {
  v = row16_sum(v);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) { return quad4_max(row16_max(v)); }

__device__ __forceinline__ int gload1i(const int32_t* base, int elem_off) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(carca_rsrc(base), elem_off * 4, 0, 0);
}
__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// head-padded feature index -> original feature index, or -1 for a pad slot
__device__ __forceinline__ int unpad_feature(int fp, int dh, int dhp) {
  int h = fp / dhp, r = fp - h * dhp;
  return r < dh ? h * dh + r : -1;
}

// ---- deterministic gradient accumulation (carca_set_tuning(CARCA_TUNE_DETERMINISTIC, 1); api.hip) ----------------
// fp32 atomics make a gradient's last bits follow the order in which workgroups happen to arrive (and Adam turns a
// round-off-sized difference of a near-cancelling sum into a +-lr step).  Under the mode every accumulation into the
// backward's flat gradient buffer goes, as a 64-bit FIXED-POINT integer (value x 2^CARCA_DET_SHIFT, round to nearest),
// into a shadow buffer of the same geometry: integer addition is associative, so the sum does not depend on the order,
// bit for bit.  carca_det_flush adds the shadow into the fp32 buffer (one rounding per element) and clears it.
// Resolution 2^-36 = 1.5e-11 per contribution (an fp32 ulp at 1e-4), range +-1.3e8.  Addresses outside the registered
// buffer keep the plain fp32 atomic.
struct CarcaDetCtx {
  float* base;                 // the flat gradient buffer of the pass in flight (null: between passes)
  unsigned long long* shadow;  // n 64-bit accumulators
  long long n;
  float scale, inv_scale;
};
#define CARCA_DET_SHIFT 36
namespace {
__device__ const CarcaDetCtx* g_carca_det = nullptr;  // one copy per translation unit, all bound to the same context (api.hip)
}
__device__ __forceinline__ void grad_add(float* p, float v) {
  const CarcaDetCtx* c = g_carca_det;
  if (__builtin_expect(c != nullptr, 0)) {
    float* const base = c->base;
    const long long off = p - base;
    // (a contribution the fixed-point format cannot hold -- NaN, +-Inf, |v| >= 2^27 -- takes the fp32 atomic below instead
    // of saturating silently: a diverging run must show non-finite gradients here as it does on the default path)
    if (base && off >= 0 && off < c->n && fabsf(v) < 134217728.f) {
      atomicAdd(c->shadow + off, (unsigned long long)__float2ll_rn(v * c->scale));
      return;
    }
  }
  atomicAdd(p, v);
}
// every translation unit that includes this header registers a binder for its own copy of the symbol
typedef int (*carca_det_binder)(const CarcaDetCtx*);
void carca_det_register(carca_det_binder fn);
namespace {
int carca_det_bind_this_tu(const CarcaDetCtx* ctx) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_carca_det), &ctx, sizeof(ctx), 0, hipMemcpyHostToDevice);
}
struct CarcaDetRegistrar {
  CarcaDetRegistrar() { carca_det_register(&carca_det_bind_this_tu); }
};
CarcaDetRegistrar g_carca_det_registrar;
}  // namespace

// error codes of the C ABI (include/carca_hip.h)
#define CARCA_OK 0
#define CARCA_ERR_UNSUPPORTED (-1)
#define CARCA_ERR_BADARG (-2)

void carca_set_error(const char* fmt, ...);
// zq[r, 0:d] = items_w[ids[r]] * sqrt(d) for the rows of all segments (carca.py:87-88).  Runs either as its own launch
// (embed.hip) or as a PASSENGER block of the feature-GEMM launch (gemm.hip): that launch has 255 blocks for 256 CUs at
// C2, and one extra workgroup on the idle CU copies the 7 MB while the others multiply.
struct CarcaGatherArgs {
  const int32_t* ids[4 /*CARCA_MAX_SEGS*/];
  int row_start[5];
  int nseg;
  const float* items_w;
  float* zq;
  int d, ldz, total_rows;
  float scale;
};
// RIF rows in flight per wave: a lone workgroup (the passenger) needs them -- one row at a time is a ~1.3 us dependent
// chain per row (id, row, store), 2.1 ms for the 19 k rows of C2 on 12 waves; with 16 in flight ~0.2 ms.
template <int RIF = 1>
__device__ __forceinline__ void carca_gather_rows(const CarcaGatherArgs& ga, int wave, int nwaves, int lane) {
  for (int row0 = wave * RIF; row0 < ga.total_rows; row0 += nwaves * RIF) {
    const float* src[RIF];
#pragma unroll
    for (int i = 0; i < RIF; ++i) {
      const int row = min(row0 + i, ga.total_rows - 1);
      int s = 0;
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (j < ga.nseg && row >= ga.row_start[j]) s = j;
      src[i] = ga.items_w + (size_t)ga.ids[s][row - ga.row_start[s]] * ga.d;
    }
    for (int c0 = 0; c0 < ga.d; c0 += 64) {
      const int c = c0 + lane;
      float v[RIF];
#pragma unroll
      for (int i = 0; i < RIF; ++i) v[i] = c < ga.d ? src[i][c] : 0.f;
#pragma unroll
      for (int i = 0; i < RIF; ++i)
        if (c < ga.d && row0 + i < ga.total_rows) ga.zq[(size_t)(row0 + i) * ga.ldz + c] = v[i] * ga.scale;
    }
  }
}

// First statement of a kernel with a big argument block: ONE scalar load per 64-byte line of the kernel-argument segment, all
// in flight together.  hipcc loads arguments where the code first needs them; behind branches on other arguments that is a
// chain of dependent scalar loads, each a cold miss when the kernel starts -- gemm_rows_skc_kernel's eight of them were the
// 8.9 us in front of its first barrier (tools/stamp_skc.py).
template <int BYTES>
__device__ __forceinline__ void carca_warm_kernargs() {
  typedef const __attribute__((address_space(4))) int* karg_ptr;
  karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  int w = 0;
#pragma unroll
  for (int i = 0; i < (BYTES + 63) / 64; ++i) w ^= ka[i * 16];
  asm volatile("" ::"s"(w));
}

// tuning knobs (api.hip): small integers a tuning run selects with carca_set_tuning(); 0 = shipped choice.  ONE meaning
// per key (include/carca_hip.h lists them); keys 3..7 are used by number.
enum { CARCA_TUNE_GEMM_VARIANT = 0, CARCA_TUNE_ATTN_VARIANT = 1, CARCA_TUNE_WGRAD_SLOTS = 2, CARCA_TUNE_DETERMINISTIC = 8,
       CARCA_TUNE_STAMPS = 9, CARCA_TUNE_CU_CAP = 10, CARCA_TUNE_SK_DON = 11, CARCA_TUNE_SK_SPIN_LOG2 = 12,
       CARCA_TUNE_SK_WITHHOLD = 13, CARCA_TUNE_XS_OPT = 14, CARCA_TUNE_DIAG = 15, CARCA_TUNE_SPLIT_GEMM = 16,
       CARCA_TUNE_COUNT = 24 };
int carca_tuning(int key);
int carca_num_cus();  // compute units of the current device (cached)
// True while `stream` is being captured into a hipGraph.
bool carca_stream_capturing(hipStream_t stream);
// Memory the library hands its own kernels (partial-tile buffers, row tables, flags) -- two owners, no host wait in either:
//  * eager launches: carca_stream_scratch, ONE buffer per (device, stream, tag), grown on demand.  Its users are ordered by
//    the stream itself (a launch that reads the buffer has finished before the next launch on that stream starts), so
//    there is no event and no ring; an outgrown buffer is freed once its stream has drained.  zero_bytes: cleared on
//    `stream` when the buffer is (re)allocated (state the kernels keep clean themselves); *fresh tells that it was.
//  * launches on a stream that is being CAPTURED: carca_capture_alloc, memory that belongs to the capture in progress
//    (a replay runs without the host code that would pick a buffer, and captures may replay side by side).  It is
//    accounted to the capture's id (carca_capture_scope) and freed by carca_capture_release once the graph is gone.
void* carca_stream_scratch(hipStream_t stream, int tag, size_t bytes, size_t zero_bytes = 0, bool* fresh = nullptr);
enum { CARCA_SCRATCH_SK = 1, CARCA_SCRATCH_WPART = 2, CARCA_SCRATCH_WTAB = 3, CARCA_SCRATCH_SPLITW = 4, CARCA_SCRATCH_SKC = 5,
       CARCA_SCRATCH_SKC_PART = 6 };
void* carca_capture_alloc(hipStream_t stream, size_t bytes, bool host_mapped, void** device_view, size_t zero_bytes = 0);
// Timing events for this thread's NEXT row-GEMM launch (the roofline hooks of carca_forward): the launch binds them to
// its own dispatch packet (hipExtLaunchKernel), so elapsed(start, stop) is the kernel's duration and no barrier packet
// is queued around it (an hipEventRecord is one: ~6 us of GPU time between two kernels each).
void carca_arm_launch_events(void* start, void* stop);
bool carca_take_launch_events(hipEvent_t* start, hipEvent_t* stop);  // true (and disarms) when armed
struct CarcaGemmDesc;
// appends to the row-GEMM kernel log while carca_gemm_rows_log has it switched on (gemm.hip)
void carca_rows_log(const char* kernel, const CarcaGemmDesc* d, int grid);
// the opt-in split-precision feature GEMM (gemm_split.hip, tuning key 16): CARCA_OK = launched, 1 = not its product
int carca_gemm_rows_split_try(const CarcaGemmDesc* desc, hipStream_t stream);
// the persistent short-K kernel (gemm_stream.hip: many tiles per CU, K steps of consecutive tiles as one stream); fits32: every
// operand offset fits 32 bits of bytes (gemm_rows_choose).  CARCA_OK = launched, 1 = not its product
int carca_gemm_rows_stream_try(const CarcaGemmDesc* desc, bool fits32, hipStream_t stream);
// ... and the persistent narrow-output kernel (64 < N <= 96, many rows per CU: the joint embedding at C5 / C3)
int carca_gemm_rows_n96s_try(const CarcaGemmDesc* desc, bool fits32, hipStream_t stream);
// carca_gemm_rows with the item-row gather riding along where the kernel choice leaves a CU idle; *rode tells whether
// it did (otherwise the caller launches the gather itself)
int carca_gemm_rows_passenger(const CarcaGemmDesc* desc, const CarcaGatherArgs* ga, int* rode, void* stream);
// host-visible word a kernel sets when it meets something no launch status can carry (1: a stream-K taker gave up waiting,
// 2: gemm_rows_skc_kernel's row-block lists too short, 3: gemm_wgrad_cu_kernel's partial-tile slots too few); reported and
// cleared by carca_poll_errors and by the next stream-K launch (gemm.hip).  Device view, or null.
int* carca_kernel_error_word();
// the joint embedding over q's columns with the item term from a projected table (embed.hip; CarcaForwardDesc.z_table)
struct CarcaRowSeg;
int carca_embed_joint_ztab(const CarcaRowSeg* segs, int nseg, int d, int g, const float* joint_w, const float* joint_b,
                           const float* pos, const float* zq, int ld_e, const float* z_table, int ld_z_table, void* stream);
// the persistent weight-gradient kernel's row table on a second stream, forked at the caller's entry (wgrad_cu.hip)
int carca_wgrad_table_fork(hipStream_t stream, hipStream_t table_stream);
int carca_wgrad_table_join(hipStream_t stream, bool used = false);
struct CarcaWgradDesc;
bool carca_wgrad_cu_suited(const CarcaWgradDesc* desc);
// carca_embed_scatter over several row segments in one launch (backward.hip)
int carca_embed_scatter_segs(const float* const* dz, int ld_dz, const int32_t* const* ids, const int* rows, int nseg,
                             int d, float scale, float* d_items, void* stream);
// fused row chains of the SelfAttentionBlock backward (row_chain.hip)
int carca_sa_ffn_chain_bwd(const float* dy, const float* h1, const float* r, const float* w2_t, const float* w1_t,
                           const float* ln2_w, int rows, int d, int dpi, int residual, float* dh1pre, float* dr,
                           float* g_ln2_w, float* g_ln2_b, void* stream);
int carca_sa_input_chain_bwd(const float* dqh, const float* dkh, const float* dvh, const float* dr, const float* x_in,
                             const float* wq_t, const float* wk_t, const float* wv_t, const float* ln1_w, int rows, int d,
                             int dpi, int residual, float* dx, float* g_ln1_w, float* g_ln1_b, void* stream);
unsigned long long* carca_debug_buffer();  // device buffer for in-kernel phase stamps (diagnostic runs), or null
#define CARCA_CHECK_ARG(cond, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      carca_set_error(__VA_ARGS__);           \
      return CARCA_ERR_BADARG;                \
    }                                         \
  } while (0)
#define CARCA_CHECK_SUPPORTED(cond, ...)      \
  do {                                        \
    if (!(cond)) {                            \
      carca_set_error(__VA_ARGS__);           \
      return CARCA_ERR_UNSUPPORTED;           \
    }                                         \
  } while (0)
#define CARCA_LAUNCH_CHECK()                                               \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      carca_set_error("HIP launch failed: %s", hipGetErrorString(e_));     \
      return (int)e_;                                                      \
    }                                                                      \
  } while (0)
