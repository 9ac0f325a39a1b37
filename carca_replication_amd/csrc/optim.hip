// f3: the optimizer step of the train driver as ONE launch.
// The reference trains with torch.optim.Adam(lr, betas=(0.9, 0.98), weight_decay) (scripts/training.py:174,
// src/train.py:96); its update for parameter p with gradient g at step t (no amsgrad, L2-style weight decay) is
//   g  = g + wd * p
//   m  = m + (1 - b1) (g - m)                  (torch writes it as lerp)
//   v  = b2 v + (1 - b2) g g
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// All of a model's tensors are walked by one grid: a table of (p, g, m, v, n) entries passed by value, blocks assigned
// to 1024-element chunks through a prefix of chunk counts.  HBM-bound: 7 floats moved per element (read p g m v, write
// p m v) = 28 B/element, 87 MB for the 3.09 M parameters of C2.
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

constexpr int ADAM_MAX = 48;     // tensors per launch (3.6 KB of kernel arguments)
constexpr int ADAM_CHUNK = 1024; // elements per block

struct AdamTable {
  CarcaAdamTensor t[ADAM_MAX];
  int chunk_start[ADAM_MAX + 1];
  int n;
};
struct AdamScalars {
  float lr_c1;      // lr / (1 - b1^t)
  float sqrt_c2;    // sqrt(1 - b2^t)
  float omb1, b2, omb2, eps, wd;  // 1 - b1 and 1 - b2 rounded from double like torch's python scalars
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& s) {
  g = g + s.wd * p;
  m = m + s.omb1 * (g - m);
  v = s.b2 * v + s.omb2 * g * g;
  p = p - s.lr_c1 * m / (sqrtf(v) / s.sqrt_c2 + s.eps);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tab, const AdamScalars s) {
  int ti = 0;
  for (int i = 1; i < tab.n; ++i)
    if ((int)blockIdx.x >= tab.chunk_start[i]) ti = i;
  const CarcaAdamTensor T = tab.t[ti];
  const int64_t base = (int64_t)((int)blockIdx.x - tab.chunk_start[ti]) * ADAM_CHUNK;
  const int64_t left = T.n - base;
  const bool aligned = (((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0;
  if (T.row_mask && !((T.row_len & 3) == 0 && aligned)) {  // rows or pointers off the 16-byte grid: one float per thread
    for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < T.n; i += 256) {
      if (!T.row_mask[i / T.row_len]) continue;
      float p = T.p[i], m = T.m[i], v = T.v[i];
      adam_one(p, T.g[i], m, v, s);
      T.p[i] = p;
      T.m[i] = m;
      T.v[i] = v;
    }
    return;
  }
  if (T.row_mask) {  // embedding table with a row mask: untouched rows are skipped (see the header)
    for (int64_t i = base + threadIdx.x * 4; i < base + ADAM_CHUNK && i < T.n; i += 1024) {
      if (!T.row_mask[i / T.row_len]) continue;
      f32x4 p = *reinterpret_cast<const f32x4*>(T.p + i), m = *reinterpret_cast<const f32x4*>(T.m + i);
      f32x4 v = *reinterpret_cast<const f32x4*>(T.v + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(T.g + i);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = p[r], mr = m[r], vr = v[r];
        adam_one(pr, g[r], mr, vr, s);
        p[r] = pr;
        m[r] = mr;
        v[r] = vr;
      }
      *reinterpret_cast<f32x4*>(T.p + i) = p;
      *reinterpret_cast<f32x4*>(T.m + i) = m;
      *reinterpret_cast<f32x4*>(T.v + i) = v;
    }
    return;
  }
  const bool vec = left >= ADAM_CHUNK && aligned;
  if (vec) {  // one float4 per thread
    const int64_t i = base + threadIdx.x * 4;
    f32x4 p = *reinterpret_cast<const f32x4*>(T.p + i), m = *reinterpret_cast<const f32x4*>(T.m + i);
    f32x4 v = *reinterpret_cast<const f32x4*>(T.v + i);
    const f32x4 g = *reinterpret_cast<const f32x4*>(T.g + i);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pr = p[r], mr = m[r], vr = v[r];
      adam_one(pr, g[r], mr, vr, s);
      p[r] = pr;
      m[r] = mr;
      v[r] = vr;
    }
    *reinterpret_cast<f32x4*>(T.p + i) = p;
    *reinterpret_cast<f32x4*>(T.m + i) = m;
    *reinterpret_cast<f32x4*>(T.v + i) = v;
  } else {
    for (int64_t i = base + threadIdx.x; i < base + ADAM_CHUNK && i < T.n; i += 256) {
      float p = T.p[i], m = T.m[i], v = T.v[i];
      adam_one(p, T.g[i], m, v, s);
      T.p[i] = p;
      T.m[i] = m;
      T.v[i] = v;
    }
  }
}

__global__ void mark_rows_kernel(const int32_t* __restrict__ ids, int64_t n, uint8_t* __restrict__ mask, int64_t n_rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int64_t r = ids[i];
    if (r >= 0 && r < n_rows) mask[r] = 1;
  }
}
struct IdLists {
  const int32_t* ids[CARCA_MAX_SEGS];
  long long start[CARCA_MAX_SEGS + 1];
  int n;
};
// one wave per id: its row is cleared with 16-byte stores (row_len % 4 == 0) or element stores
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ table, int64_t n_rows, int row_len, const IdLists L) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= L.start[L.n]) return;
  int s = 0;
  for (int j = 1; j < L.n; ++j)
    if (i >= L.start[j]) s = j;
  const int64_t r = L.ids[s][i - L.start[s]];
  if (r < 0 || r >= n_rows) return;
  float* row = table + r * row_len;
  if ((row_len & 3) == 0 && (((uintptr_t)table) & 15) == 0) {
    for (int c = lane * 4; c < row_len; c += 256) *reinterpret_cast<f32x4*>(row + c) = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    for (int c = lane; c < row_len; c += 64) row[c] = 0.f;
  }
}
__global__ void concat_ids_kernel(const IdLists L, int32_t* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.start[L.n]) return;
  int s = 0;
  for (int j = 1; j < L.n; ++j)
    if (i >= L.start[j]) s = j;
  out[i] = L.ids[s][i - L.start[s]];
}
static int make_lists(const int32_t* const* ids, const int64_t* counts, int nlists, IdLists* L, const char* who) {
  CARCA_CHECK_ARG(ids && counts && nlists >= 1 && nlists <= CARCA_MAX_SEGS, "%s: 1..%d id lists", who, CARCA_MAX_SEGS);
  long long t = 0;
  for (int i = 0; i < nlists; ++i) {
    CARCA_CHECK_ARG(ids[i] && counts[i] >= 0, "%s: list %d malformed", who, i);
    L->ids[i] = ids[i];
    L->start[i] = t;
    t += counts[i];
  }
  L->start[nlists] = t;
  L->n = nlists;
  return CARCA_OK;
}

}  // namespace

extern "C" int carca_mark_rows(const int32_t* ids, int64_t n, uint8_t* mask, int64_t n_rows, void* stream_) {
  CARCA_CHECK_ARG(ids && mask && n >= 0 && n_rows >= 1, "mark_rows: null pointer or bad sizes");
  if (n == 0) return CARCA_OK;
  hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, ids, n, mask, n_rows);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_zero_rows(float* table, int64_t n_rows, int row_len, const int32_t* const* ids, const int64_t* counts,
                               int nlists, void* stream_) {
  CARCA_CHECK_ARG(table && n_rows >= 1 && row_len >= 1, "zero_rows: null table or bad sizes");
  IdLists L{};
  int rc = make_lists(ids, counts, nlists, &L, "zero_rows");
  if (rc != CARCA_OK) return rc;
  if (L.start[L.n] == 0) return CARCA_OK;
  const long long blocks = (L.start[L.n] + 3) / 4;  // one wave per id, four waves per block
  CARCA_CHECK_SUPPORTED(blocks < (1ll << 31), "zero_rows: too many ids");
  hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, table, n_rows, row_len, L);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_concat_ids(const int32_t* const* ids, const int64_t* counts, int nlists, int32_t* out, void* stream_) {
  CARCA_CHECK_ARG(out, "concat_ids: null output");
  IdLists L{};
  int rc = make_lists(ids, counts, nlists, &L, "concat_ids");
  if (rc != CARCA_OK) return rc;
  if (L.start[L.n] == 0) return CARCA_OK;
  hipLaunchKernelGGL(concat_ids_kernel, dim3((unsigned)((L.start[L.n] + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, L, out);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_adam_step(const CarcaAdamTensor* tensors, int n, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int step, void* stream_) {
  CARCA_CHECK_ARG(tensors && n >= 1, "adam_step: no tensors");
  CARCA_CHECK_ARG(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. &&
                      weight_decay >= 0.,
                  "adam_step: bad hyper-parameters (step=%d lr=%g betas=(%g, %g) eps=%g wd=%g)", step, lr, beta1, beta2,
                  eps, weight_decay);
  for (int i = 0; i < n; ++i) {
    CARCA_CHECK_ARG(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n >= 0,
                    "adam_step: tensor %d malformed", i);
    CARCA_CHECK_ARG(!tensors[i].row_mask || (tensors[i].row_len >= 1 && weight_decay == 0.),
                    "adam_step: tensor %d: a row mask needs row_len >= 1 and weight_decay = 0 (decay moves untouched rows)", i);
  }
  AdamScalars s;
  s.lr_c1 = (float)(lr / (1.0 - pow(beta1, step)));  // the host arithmetic torch does in double, rounded once
  s.sqrt_c2 = (float)sqrt(1.0 - pow(beta2, step));
  s.omb1 = (float)(1.0 - beta1);
  s.b2 = (float)beta2;
  s.omb2 = (float)(1.0 - beta2);
  s.eps = (float)eps;
  s.wd = (float)weight_decay;
  for (int first = 0; first < n; first += ADAM_MAX) {
    AdamTable tab{};
    tab.n = n - first < ADAM_MAX ? n - first : ADAM_MAX;
    long long chunks = 0;
    for (int i = 0; i < tab.n; ++i) {
      tab.t[i] = tensors[first + i];
      tab.chunk_start[i] = (int)chunks;
      chunks += (tensors[first + i].n + ADAM_CHUNK - 1) / ADAM_CHUNK;
      CARCA_CHECK_SUPPORTED(chunks < (1ll << 31), "adam_step: too many elements in one launch");
    }
    tab.chunk_start[tab.n] = (int)chunks;
    if (chunks == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream_, tab, s);
    CARCA_LAUNCH_CHECK();
  }
  return CARCA_OK;
}
