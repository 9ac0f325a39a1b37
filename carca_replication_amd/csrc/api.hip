// Error plumbing, geometry helper and weight packing of the C ABI (include/carca_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "carca_common.h"
#include "../../include/carca_hip.h"

static thread_local char g_err[512] = "";

void carca_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int g_tuning[CARCA_TUNE_COUNT] = {0};
static unsigned long long* g_debug = nullptr;
unsigned long long* carca_debug_buffer() { return g_debug; }
extern "C" int carca_set_debug_buffer(void* p) {
  g_debug = (unsigned long long*)p;
  return CARCA_OK;
}
int carca_tuning(int key) { return (key >= 0 && key < CARCA_TUNE_COUNT) ? g_tuning[key] : 0; }
int carca_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return n;
}
// ---- deterministic gradient accumulation (carca_common.h: grad_add) ------------------------------------------------
static carca_det_binder g_det_binders[64];
static int g_det_nbinders = 0;
static CarcaDetCtx* g_det_ctx = nullptr;  // device memory, allocated at the first switch-on, lives as long as the process
void carca_det_register(carca_det_binder fn) {
  if (g_det_nbinders < 64) g_det_binders[g_det_nbinders++] = fn;
}
namespace {
__global__ void det_set_kernel(CarcaDetCtx* ctx, float* base, unsigned long long* shadow, long long n) {
  ctx->base = base;
  ctx->shadow = shadow;
  ctx->n = n;
  ctx->scale = (float)(1ull << CARCA_DET_SHIFT);
  ctx->inv_scale = 1.0f / (float)(1ull << CARCA_DET_SHIFT);
}
__global__ __launch_bounds__(256) void det_flush_kernel(float* __restrict__ flat, unsigned long long* __restrict__ shadow,
                                                        long long lo, long long hi) {
  const double inv = 1.0 / (double)(1ull << CARCA_DET_SHIFT);
  for (long long i = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (long long)gridDim.x * blockDim.x) {
    const long long s = (long long)shadow[i];
    if (s != 0) {
      flat[i] += (float)((double)s * inv);
      shadow[i] = 0;
    }
  }
}
}  // namespace
static int det_switch(bool on) {
  if (on && !g_det_ctx) {
    hipError_t e = hipMalloc((void**)&g_det_ctx, sizeof(CarcaDetCtx));
    if (e != hipSuccess) {
      carca_set_error("set_tuning: cannot allocate the determinism context: %s", hipGetErrorString(e));
      return (int)e;
    }
    e = hipMemset(g_det_ctx, 0, sizeof(CarcaDetCtx));
    if (e != hipSuccess) return (int)e;
  }
  for (int i = 0; i < g_det_nbinders; ++i) {
    const int rc = g_det_binders[i](on ? g_det_ctx : nullptr);
    if (rc != 0) {
      carca_set_error("set_tuning: cannot bind the determinism context: %s", hipGetErrorString((hipError_t)rc));
      return rc;
    }
  }
  return CARCA_OK;
}
extern "C" int carca_det_begin(float* flat, long long n, unsigned long long* shadow, void* stream) {
  CARCA_CHECK_ARG(g_det_ctx && carca_tuning(CARCA_TUNE_DETERMINISTIC) != 0,
                  "det_begin: the deterministic mode is off (carca_set_tuning(8, 1))");
  CARCA_CHECK_ARG((flat && shadow && n > 0) || (!flat && !shadow), "det_begin: buffer, shadow and size go together");
  hipLaunchKernelGGL(det_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g_det_ctx, flat, shadow, flat ? n : 0);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}
extern "C" int carca_det_flush(float* flat, unsigned long long* shadow, long long lo, long long hi, void* stream) {
  CARCA_CHECK_ARG(flat && shadow && lo >= 0 && hi >= lo, "det_flush: null buffer or bad range");
  if (hi == lo) return CARCA_OK;
  long long blocks = (hi - lo + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(det_flush_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, shadow, lo, hi);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_set_tuning(int key, int value) {
  CARCA_CHECK_ARG(key >= 0 && key < CARCA_TUNE_COUNT, "set_tuning: unknown key %d", key);
  if (key == CARCA_TUNE_DETERMINISTIC && (value != 0) != (g_tuning[key] != 0)) {
    const int rc = det_switch(value != 0);
    if (rc != CARCA_OK) return rc;
  }
  g_tuning[key] = value;
  return CARCA_OK;
}

extern "C" int carca_abi_version(void) { return CARCA_ABI_VERSION; }
extern "C" const char* carca_last_error(void) { return g_err; }

extern "C" int carca_padded_dims(int d, int H, int* dpi, int* dhp, int* dpo) {
  CARCA_CHECK_ARG(d >= 1 && H >= 1 && d % H == 0, "padded_dims: d=%d not divisible by H=%d", d, H);
  CARCA_CHECK_SUPPORTED(d <= 128, "padded_dims: d=%d > 128 is not built", d);
  const int p = round_up(d / H, 16);
  if (dpi) *dpi = d <= 64 ? 64 : (d <= 96 ? 96 : 128);
  if (dhp) *dhp = p;
  if (dpo) *dpo = H * p;
  return CARCA_OK;
}

namespace {
constexpr int PACK_CHUNK = 48;  // descriptors per launch (3.5 KB of kernel arguments): a model's forward packs fit one
struct PackArgs {
  CarcaPackDesc d[PACK_CHUNK];
};

// one blockIdx.y per descriptor; threads stride over the destination index space
__global__ void pack_kernel(const PackArgs pa) {
  const CarcaPackDesc& ds = pa.d[blockIdx.y];
  const int total = ds.dst_rows * ds.dst_cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int rp = i / ds.dst_cols, cp = i - rp * ds.dst_cols;
    if (ds.frag16) {  // i = ((tile_row * tiles_per_row + tile_col) * 64 + lane) * 4 + r, lane = 16 * mq + ln
      const int t = i >> 8, nkg = ds.dst_cols >> 4;
      rp = 16 * (t / nkg) + ((i >> 2) & 15);
      cp = 16 * (t % nkg) + 4 * ((i >> 6) & 3) + (i & 3);
    }
    const int r = ds.row_dh > 0 ? unpad_feature(rp, ds.row_dh, ds.row_dhp) : rp;
    const int c = ds.col_dh > 0 ? unpad_feature(cp, ds.col_dh, ds.col_dhp) : cp;
    float v = 0.f;
    if (ds.fold_vec) {  // logical source [fold_H, cols]: rows of src contracted per head with fold_vec
      if (r >= 0 && r < ds.fold_H && c >= 0 && c < ds.cols) {
        const int dh = ds.rows / ds.fold_H;
        for (int j = 0; j < dh; ++j) v += ds.fold_vec[r * dh + j] * ds.src[(size_t)(r * dh + j) * ds.src_ld + c];
      }
    } else if (r >= 0 && r < ds.rows && c >= 0 && c < ds.cols)
      v = ds.transposed ? ds.src[(size_t)c * ds.src_ld + r] : ds.src[(size_t)r * ds.src_ld + c];
    ds.dst[i] = v;
  }
}
// gradients travel the other way: real[r][c] (+)= packed[pad(r)][pad(c)]
__global__ void unpack_kernel(const PackArgs pa, int accumulate) {
  const CarcaPackDesc& ds = pa.d[blockIdx.y];
  const int total = ds.rows * ds.cols;
  float* real = const_cast<float*>(ds.src);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / ds.cols, c = i - r * ds.cols;
    const int rp = ds.row_dh > 0 ? (r / ds.row_dh) * ds.row_dhp + r % ds.row_dh : r;
    const int cp = ds.col_dh > 0 ? (c / ds.col_dh) * ds.col_dhp + c % ds.col_dh : c;
    const float v = ds.dst[(size_t)rp * ds.dst_cols + cp];
    float* q = ds.transposed ? &real[(size_t)c * ds.src_ld + r] : &real[(size_t)r * ds.src_ld + c];
    *q = accumulate ? *q + v : v;
  }
}
}  // namespace

extern "C" int carca_unpack_grads(const CarcaPackDesc* descs, int n, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(descs && n >= 1, "unpack_grads: no descriptors");
  for (int i = 0; i < n; ++i) {
    const CarcaPackDesc& d = descs[i];
    CARCA_CHECK_ARG(d.src && d.dst && d.rows >= 1 && d.cols >= 1 && d.dst_rows >= 1 && d.dst_cols >= 1,
                    "unpack_grads: descriptor %d malformed", i);
    CARCA_CHECK_ARG(!d.frag16 && !d.fold_vec, "unpack_grads: descriptor %d is a forward-only layout (fragment order / fold)", i);
  }
  for (int base = 0; base < n; base += PACK_CHUNK) {
    PackArgs pa{};
    const int m = min(PACK_CHUNK, n - base);
    for (int i = 0; i < m; ++i) pa.d[i] = descs[base + i];
    hipLaunchKernelGGL(unpack_kernel, dim3(8, m), dim3(256), 0, stream, pa, accumulate);
    CARCA_LAUNCH_CHECK();
  }
  return CARCA_OK;
}

extern "C" int carca_pack_weights(const CarcaPackDesc* descs, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(descs && n >= 1, "pack_weights: no descriptors");
  for (int i = 0; i < n; ++i) {
    const CarcaPackDesc& d = descs[i];
    CARCA_CHECK_ARG(d.src && d.dst && d.rows >= 1 && d.cols >= 1 && d.src_ld >= (d.transposed ? d.rows : d.cols) && d.dst_rows >= 1 &&
                        d.dst_cols >= 1,
                    "pack_weights: descriptor %d malformed", i);
    CARCA_CHECK_ARG((d.row_dh == 0) == (d.row_dhp == 0) && (d.col_dh == 0) == (d.col_dhp == 0) &&
                        d.row_dh <= d.row_dhp && d.col_dh <= d.col_dhp,
                    "pack_weights: descriptor %d has inconsistent head padding", i);
    CARCA_CHECK_ARG(!d.frag16 || (d.dst_rows % 16 == 0 && d.dst_cols % 16 == 0),
                    "pack_weights: descriptor %d: fragment order needs dst dims that are multiples of 16", i);
    CARCA_CHECK_ARG(!d.fold_vec || (d.fold_H >= 1 && d.rows % d.fold_H == 0 && d.row_dh == 0 && !d.transposed &&
                                    d.dst_rows >= d.fold_H),
                    "pack_weights: descriptor %d: malformed head fold", i);
  }
  for (int base = 0; base < n; base += PACK_CHUNK) {
    PackArgs pa{};
    const int m = min(PACK_CHUNK, n - base);
    for (int i = 0; i < m; ++i) pa.d[i] = descs[base + i];
    int big = 0;  // (an F x g transpose is 7.4 MB: 8 blocks took 450 us; the grid follows the largest descriptor)
    for (int i = 0; i < m; ++i) big = max(big, pa.d[i].dst_rows * pa.d[i].dst_cols);
    hipLaunchKernelGGL(pack_kernel, dim3(min(max(big / 2048, 8), 1024), m), dim3(256), 0, stream, pa);
    CARCA_LAUNCH_CHECK();
  }
  return CARCA_OK;
}

// ---- whole-forward host entry point + event helpers ---------------------------------------------------------
extern "C" int carca_event_create(void** ev_out) {
  CARCA_CHECK_ARG(ev_out, "event_create: null");
  hipEvent_t e;
  hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) {
    carca_set_error("hipEventCreate: %s", hipGetErrorString(rc));
    return (int)rc;
  }
  *ev_out = (void*)e;
  return CARCA_OK;
}
extern "C" int carca_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return CARCA_OK;
}
extern "C" int carca_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  CARCA_CHECK_ARG(start && stop && ms_out, "event_elapsed_ms: null");
  hipError_t rc = hipEventElapsedTime(ms_out, (hipEvent_t)start, (hipEvent_t)stop);
  if (rc != hipSuccess) {
    carca_set_error("hipEventElapsedTime: %s", hipGetErrorString(rc));
    return (int)rc;
  }
  return CARCA_OK;
}

extern "C" int carca_stream_wait_event(void* stream, void* event) {
  CARCA_CHECK_ARG(event, "stream_wait_event: null event");
  hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
  if (rc != hipSuccess) {
    carca_set_error("hipStreamWaitEvent: %s", hipGetErrorString(rc));
    return (int)rc;
  }
  return CARCA_OK;
}

namespace {
thread_local hipEvent_t g_armed_start = nullptr, g_armed_stop = nullptr;
}
bool carca_stream_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
// ---- memory the library owns on behalf of its kernels (carca_common.h) ------------------------------------------------
namespace {
std::mutex g_mem_mu;
struct ScratchSlot {
  int dev;
  hipStream_t stream;
  int tag;
  void* p;
  size_t bytes;
};
std::vector<ScratchSlot> g_scratch;
struct Retired {
  int dev;
  hipStream_t stream;
  void* p;
};
std::vector<Retired> g_retired;  // outgrown scratch: freed once its stream has drained (hipFree would block on it)
struct CaptureBlock {
  unsigned long long scope;
  void* p;
  size_t bytes;
  bool host;
};
std::vector<CaptureBlock> g_capture_blocks;
unsigned long long capture_id(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo(stream, &st, &id) != hipSuccess || st != hipStreamCaptureStatusActive) return 0;
  return id ? id : ~0ull;
}
}  // namespace

namespace {
// hipFree synchronises the device and is refused while a stream of this thread captures in global mode: outgrown buffers
// are therefore freed (a) only from a launch path that is not itself inside a capture, (b) outside g_mem_mu, (c) inside a
// relaxed capture-mode window, so that a capture running on ANOTHER thread does not turn the call into an error that
// invalidates it.
void free_outside_lock(const std::vector<void*>& ps) {
  if (ps.empty()) return;
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  for (void* p : ps) (void)hipFree(p);
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  (void)hipGetLastError();
}
}  // namespace

void* carca_stream_scratch(hipStream_t stream, int tag, size_t bytes, size_t zero_bytes, bool* fresh) {
  std::vector<void*> to_free;
  struct FreeAtExit {
    std::vector<void*>& v;
    ~FreeAtExit() { free_outside_lock(v); }
  } free_at_exit{to_free};  // (declared BEFORE the lock: runs after the mutex is released)
  std::lock_guard<std::mutex> lock(g_mem_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (fresh) *fresh = false;
  if (!carca_stream_capturing(stream)) {
    for (size_t i = 0; i < g_retired.size();) {  // (outgrown buffers whose stream has nothing left in flight)
      if (g_retired[i].dev == dev && !carca_stream_capturing(g_retired[i].stream) && hipStreamQuery(g_retired[i].stream) == hipSuccess) {
        to_free.push_back(g_retired[i].p);
        g_retired[i] = g_retired.back();
        g_retired.pop_back();
      } else {
        ++i;
      }
    }
    (void)hipGetLastError();  // (hipStreamQuery reports hipErrorNotReady through the sticky last-error too)
  }
  ScratchSlot* sl = nullptr;
  for (auto& s : g_scratch)
    if (s.dev == dev && s.stream == stream && s.tag == tag) sl = &s;
  if (!sl) {
    g_scratch.push_back(ScratchSlot{dev, stream, tag, nullptr, 0});
    sl = &g_scratch.back();
  }
  if (bytes <= sl->bytes) return sl->p;
  if (sl->p) g_retired.push_back(Retired{dev, stream, sl->p});  // launches already queued on `stream` may still use it
  sl->p = nullptr;
  sl->bytes = 0;
  const size_t want = bytes + bytes / 8;
  void* p = nullptr;
  hipError_t rc = hipMalloc(&p, want);
  if (rc == hipSuccess && zero_bytes) rc = hipMemsetAsync(p, 0, zero_bytes, stream);  // (stream-ordered: ahead of the first user)
  if (rc != hipSuccess) {
    if (p) (void)hipFree(p);
    carca_set_error("stream scratch: cannot allocate %zu B: %s", want, hipGetErrorString(rc));
    return nullptr;
  }
  sl->p = p;
  sl->bytes = want;
  if (fresh) *fresh = true;
  return p;
}

// The library's per-stream buffers of `stream` on the current device, handed back (a caller that destroys a stream it
// launched on: the slots are keyed by the raw handle and would otherwise stay for the life of the process).  The caller
// has synchronised the stream; refused while it is being captured.
extern "C" int carca_release_stream_scratch(void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(!carca_stream_capturing(stream), "release_stream_scratch: the stream is being captured");
  std::vector<void*> to_free;
  {
    std::lock_guard<std::mutex> lock(g_mem_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t kept = 0;
    for (size_t i = 0; i < g_scratch.size(); ++i) {
      if (g_scratch[i].dev == dev && g_scratch[i].stream == stream) {
        if (g_scratch[i].p) to_free.push_back(g_scratch[i].p);
      } else {
        g_scratch[kept++] = g_scratch[i];
      }
    }
    g_scratch.resize(kept);
    kept = 0;
    for (size_t i = 0; i < g_retired.size(); ++i) {
      if (g_retired[i].dev == dev && g_retired[i].stream == stream) to_free.push_back(g_retired[i].p);
      else g_retired[kept++] = g_retired[i];
    }
    g_retired.resize(kept);
  }
  free_outside_lock(to_free);
  return CARCA_OK;
}

void* carca_capture_alloc(hipStream_t stream, size_t bytes, bool host_mapped, void** device_view, size_t zero_bytes) {
  const unsigned long long scope = capture_id(stream);
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  void* p = nullptr;
  hipError_t rc;
  if (host_mapped) {
    rc = hipHostMalloc(&p, bytes, hipHostMallocMapped);
    if (rc == hipSuccess && device_view) rc = hipHostGetDevicePointer(device_view, p, 0);
  } else {
    rc = hipMalloc(&p, bytes);
    if (device_view) *device_view = p;
    // (state a kernel keeps clean itself -- the stream-K flags: every taker resets its own -- is cleared here, once, on a
    // stream of the library's own, inside the relaxed-mode window)
    if (rc == hipSuccess && zero_bytes) {
      static hipStream_t zs = nullptr;  // (a stream of our own: the legacy stream may not wait for a capturing blocking stream)
      if (!zs) rc = hipStreamCreateWithFlags(&zs, hipStreamNonBlocking);
      if (rc == hipSuccess) rc = hipMemsetAsync(p, 0, zero_bytes, zs);
      if (rc == hipSuccess) rc = hipStreamSynchronize(zs);
    }
  }
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (rc != hipSuccess) {
    carca_set_error("allocation of %zu B during stream capture failed: %s", bytes, hipGetErrorString(rc));
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_mem_mu);
  g_capture_blocks.push_back(CaptureBlock{scope, p, bytes, host_mapped});
  return p;
}

extern "C" int carca_capture_scope(void* stream, unsigned long long* id_out) {
  CARCA_CHECK_ARG(id_out, "capture_scope: null");
  *id_out = capture_id((hipStream_t)stream);
  return CARCA_OK;
}
extern "C" long long carca_capture_bytes(unsigned long long id) {
  std::lock_guard<std::mutex> lock(g_mem_mu);
  long long n = 0;
  for (const auto& b : g_capture_blocks)
    if (b.scope == id) n += (long long)b.bytes;
  return n;
}
extern "C" int carca_capture_release(unsigned long long id) {
  CARCA_CHECK_ARG(id != 0, "capture_release: 0 is not a capture");
  std::lock_guard<std::mutex> lock(g_mem_mu);
  size_t kept = 0;
  for (size_t i = 0; i < g_capture_blocks.size(); ++i) {
    const CaptureBlock& b = g_capture_blocks[i];
    if (b.scope == id) {
      (void)(b.host ? hipHostFree(b.p) : hipFree(b.p));
    } else {
      g_capture_blocks[kept++] = b;
    }
  }
  g_capture_blocks.resize(kept);
  return CARCA_OK;
}
void carca_arm_launch_events(void* start, void* stop) {
  g_armed_start = (hipEvent_t)start;
  g_armed_stop = (hipEvent_t)stop;
}
bool carca_take_launch_events(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_armed_start && !g_armed_stop) return false;
  *start = g_armed_start;
  *stop = g_armed_stop;
  g_armed_start = g_armed_stop = nullptr;
  return true;
}

extern "C" int carca_forward(const CarcaForwardDesc* D, void* const* ev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(D && D->ngroups >= 1 && D->ngroups <= CARCA_MAX_GROUPS && D->n_blocks >= 0 &&
                      D->n_blocks <= CARCA_MAX_BLOCKS,
                  "forward: bad group / block count");
  bool need_xw = false;  // the ping-pong buffers are only needed by blocks without an output of their own
  for (int i = 0; i < D->n_blocks; ++i) need_xw = need_xw || !D->x_out[i];
  CARCA_CHECK_ARG((!need_xw || (D->x_work[0] && D->x_work[1])) && (D->zq || D->fold_wc), "forward: null workspace");
  const int nseg = D->ngroups + 1;
  int rc;
#define CARCA_TRY(call) \
  do {                  \
    rc = (call);        \
    if (rc != CARCA_OK) return rc; \
  } while (0)
  if (!D->fold_wc) {
    // gather + feature GEMM in one call: the gather rides in the GEMM's launch when that leaves a CU idle (C2: 255 blocks)
    // ev[0], ev[1]: bound to the feature GEMM's own dispatch (start / end of that kernel, no packets of their own)
    if (ev && ev[0] && ev[1]) carca_arm_launch_events(ev[0], ev[1]);
    // (z_table: the item term comes out of the projected table in the joint product's epilogue -- no gather at all)
    const bool ztab = D->z_table != nullptr && !D->save_blocks && !D->save_cross;
    rc = carca_embed_fwd(D->segs, nseg, D->n_attrs, D->n_ctx, D->d, D->g, D->items_w, D->feats_w, D->feats_b,
                         D->joint_w, D->joint_b, D->pos, D->zq, D->ld_e,
                         ztab ? CARCA_EMBED_FEAT : (CARCA_EMBED_GATHER | CARCA_EMBED_FEAT), stream_);
    hipEvent_t left0, left1;
    if (carca_take_launch_events(&left0, &left1) && rc == CARCA_OK) {
      carca_set_error("forward: the feature GEMM's launch did not take the timing events");
      return CARCA_ERR_UNSUPPORTED;
    }
    if (rc != CARCA_OK) return rc;
    if (ev && D->n_events >= 8 && ev[6] && ev[7]) carca_arm_launch_events(ev[6], ev[7]);
    if (ztab)
      CARCA_TRY(carca_embed_joint_ztab(D->segs, nseg, D->d, D->g, D->joint_w, D->joint_b, D->pos, D->zq, D->ld_e,
                                       D->z_table, D->ld_z_table, stream_));
    else
      CARCA_TRY(carca_embed_fwd(D->segs, nseg, D->n_attrs, D->n_ctx, D->d, D->g, D->items_w, D->feats_w, D->feats_b,
                                D->joint_w, D->joint_b, D->pos, D->zq, D->ld_e, CARCA_EMBED_JOINT, stream_));
    {
      hipEvent_t l0, l1;
      (void)carca_take_launch_events(&l0, &l1);  // (a joint product on a kernel that does not take events: leave them unset)
    }
  } else {
    // folded embedding: e0 = sqrt(d) * E[ids] W_jz^T + bias_c ; e = ([attrs ; ctx] W_c^T + e0 (+ pos)) * mask
    CARCA_CHECK_ARG(D->fold_bias && D->fold_ldwc >= D->n_attrs + D->n_ctx, "forward: folded weights malformed");
    CarcaGemmDesc z{}, f{};
    z.nseg = f.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
      const CarcaRowSeg& sg = D->segs[s];
      CarcaGemmSeg& a = z.seg[s];
      a.a0 = D->items_w; a.a0_gather = 1; a.ids = sg.ids; a.c = sg.e_out; a.rows = sg.rows; a.T = sg.T;
      CarcaGemmSeg& b = f.seg[s];
      b.a0 = sg.attrs_table ? sg.attrs_table : sg.attrs; b.a0_gather = sg.attrs_table ? max(1, sg.attrs_table_rows) : 0;
      b.a0_bstride = sg.attrs_table ? 0 : sg.attrs_bstride;
      b.a1 = sg.ctx; b.a1_bstride = sg.ctx_bstride;
      b.ids = sg.ids; b.c = sg.e_out; b.add = sg.e_out; b.rows = sg.rows; b.T = sg.T; b.add_pos = sg.add_pos;
    }
    z.lda0 = D->d; z.K0 = D->d; z.bt0 = D->joint_w; z.ldb0 = D->d + D->g;
    z.N = D->d; z.ldc = D->ld_e; z.ncols_out = D->ld_e; z.bias = D->fold_bias; z.alpha = (float)sqrt((double)D->d);
    CARCA_TRY(carca_gemm_rows(&z, stream_));
    f.lda0 = D->n_attrs; f.lda1 = D->n_ctx; f.K0 = D->n_attrs; f.K1 = D->n_ctx;
    f.bt0 = D->fold_wc; f.ldb0 = D->fold_ldwc; f.bt1 = D->fold_wc + D->n_attrs; f.ldb1 = D->fold_ldwc;
    f.N = D->d; f.ldc = D->ld_e; f.ncols_out = D->ld_e; f.ld_add = D->ld_e; f.pos = D->pos; f.mask_rows = 1;
    if (ev && ev[0] && ev[1]) carca_arm_launch_events(ev[0], ev[1]);
    CARCA_TRY(carca_gemm_rows(&f, stream_));
  }
  if (D->p_embed > 0.f) {  // CARCA.dropout on the profile embedding (carca.py:416), in place
    CARCA_CHECK_ARG(D->m_embed && D->p_embed < 1.f, "forward: embedding dropout needs its mask buffer and p < 1");
    CarcaDropout dr{D->p_embed, D->seed, 1000u, D->seed_offset};
    CARCA_TRY(carca_dropout_fwd(D->segs[0].e_out, D->B * D->L, D->d, D->ld_e, &dr, D->m_embed, stream_));
  }
  const float* x = D->segs[0].e_out;
  for (int i = 0; i < D->n_blocks; ++i) {
    float* y = D->x_out[i] ? D->x_out[i] : D->x_work[i & 1];
    CarcaDropout dr{D->p_block, D->seed, (uint32_t)(4 * i), D->seed_offset};
    if (i == 0 && ev && D->n_events >= 8 && ev[4] && ev[5]) carca_arm_launch_events(ev[4], ev[5]);
    if (!D->save_blocks && !(D->p_block > 0.f) && !(D->p_embed > 0.f))
      // eval: the profile's leading pad rows are equal here (zeros out of the masked embedding, carca.py:94, then each
      // block's own image of them): the kernel computes one of them per user
      CARCA_TRY(carca_sa_block_eval(x, D->ld_e, D->segs[0].ids, y, D->ld_e, D->B, D->L, D->d, D->H, &D->sa[i],
                                    D->sa_residual[i], 1, stream_));
    else
      CARCA_TRY(carca_sa_block_fwd(x, D->ld_e, D->segs[0].ids, y, D->ld_e, D->B, D->L, D->d, D->H, &D->sa[i],
                                   D->sa_residual[i], D->save_blocks ? &D->sa_save[i] : nullptr,
                                   D->p_block > 0.f ? &dr : nullptr, stream_));
    x = y;
  }
  CarcaTargetGroup groups[CARCA_MAX_GROUPS];
  for (int gi = 0; gi < D->ngroups; ++gi) {
    groups[gi].o = D->segs[gi + 1].e_out;
    groups[gi].ids = D->segs[gi + 1].ids;
    groups[gi].y = D->y[gi];
    groups[gi].N = D->N[gi];
    groups[gi].ldy = D->ldy;
  }
  // ev[2], ev[3]: bound to the scoring kernel's own dispatch, like ev[0], ev[1] to the feature GEMM's
  if (ev && ev[2] && ev[3]) carca_arm_launch_events(ev[2], ev[3]);
  CarcaDropout drc{D->p_cross, D->seed, 2000u, D->seed_offset};
  CARCA_TRY(carca_cross_score_fwd(x, D->ld_e, D->segs[0].ids, D->p_normed, groups, D->ngroups, D->ld_e, D->B, D->L,
                                  D->d, D->H, &D->ca, D->ca_residual, D->training, D->save_cross ? &D->ca_save : nullptr,
                                  (D->save_cross && D->p_cross > 0.f) ? &drc : nullptr, stream_));
  {
    hipEvent_t left0, left1;
    if (carca_take_launch_events(&left0, &left1)) {
      carca_set_error("forward: the scoring kernel's launch did not take the timing events");
      return CARCA_ERR_UNSUPPORTED;
    }
  }
#undef CARCA_TRY
  return CARCA_OK;
}
