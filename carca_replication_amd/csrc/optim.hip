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

constexpr int ADAM_MAX = 64;     // tensors per launch (3.5 KB of kernel arguments)
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
  const bool vec = left >= ADAM_CHUNK && (((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0;
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

}  // namespace

extern "C" int carca_adam_step(const CarcaAdamTensor* tensors, int n, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int step, void* stream_) {
  CARCA_CHECK_ARG(tensors && n >= 1, "adam_step: no tensors");
  CARCA_CHECK_ARG(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. &&
                      weight_decay >= 0.,
                  "adam_step: bad hyper-parameters (step=%d lr=%g betas=(%g, %g) eps=%g wd=%g)", step, lr, beta1, beta2,
                  eps, weight_decay);
  for (int i = 0; i < n; ++i)
    CARCA_CHECK_ARG(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n >= 0,
                    "adam_step: tensor %d malformed", i);
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
