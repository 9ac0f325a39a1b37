// K5: masked BinaryCrossEntropy (carca.py:441-444) and sort-free HR@k / NDCG@k (train.py:15-32).
#include "carca_common.h"
#include "../../include/carca_hip.h"

namespace {

// One 1024-thread block: fixed summation order, so the loss is bitwise reproducible run to run.
// Up to 16 elements per thread (n <= 16384: every batch of the reference's shapes) are loaded ONCE, all at the same time,
// through clamped indices (no load sits behind a branch: hipcc would wait for each before issuing the next), and the
// gradient pass reuses the registers -- the two-pass loop over global memory was 12.7 us of dependent round trips for
// 12,800 elements on one CU; larger batches keep the loop.
constexpr int BCE_PER = 16;
__global__ __launch_bounds__(1024) void bce_kernel(const float* __restrict__ y, const int32_t* __restrict__ y_true,
                                                   const int32_t* __restrict__ ids, int n, float eps,
                                                   float* __restrict__ scratch, float* __restrict__ loss_out,
                                                   float* __restrict__ dy, const float* __restrict__ denom) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool small = n <= 1024 * BCE_PER;
  float pr[BCE_PER], tr[BCE_PER], mr[BCE_PER];
  float sl = 0.f, sm = 0.f;
  if (small) {
#pragma unroll
    for (int j = 0; j < BCE_PER; ++j) {
      const int i = min(tid + 1024 * j, n - 1);
      pr[j] = y[i];
      tr[j] = (float)y_true[i];
      mr[j] = ids[i] != 0 ? 1.f : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BCE_PER; ++j) {  // (same order of additions as the loop below: element tid + 1024 j in turn)
      if (tid + 1024 * j >= n) mr[j] = 0.f;
      const float l = -(tr[j] * logf(pr[j] + eps) + (1.0f - tr[j]) * logf(1.0f - pr[j] + eps));
      if (tid + 1024 * j < n) {
        sl += l * mr[j];
        sm += mr[j];
      }
    }
  } else {
    for (int i = tid; i < n; i += 1024) {
      const float m = ids[i] != 0 ? 1.f : 0.f;
      const float t = (float)y_true[i];
      const float p = y[i];
      const float l = -(t * logf(p + eps) + (1.0f - t) * logf(1.0f - p + eps));
      sl += l * m;
      sm += m;
    }
  }
  sl = wave_sum(sl);
  sm = wave_sum(sm);
  if (lane == 0) {
    red[0][wave] = sl;
    red[1][wave] = sm;
  }
  __syncthreads();
  if (wave == 0) {
    float a = lane < 16 ? red[0][lane] : 0.f, b = lane < 16 ? red[1][lane] : 0.f;
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
      scratch[0] = a;
      scratch[1] = b;
      // denom: the mask count of the WHOLE batch when users are sharded over ranks (dist.py), else sum(mask)
      if (denom) b = denom[0];
      loss_out[0] = a / b;  // 0/0 = NaN for an all-pad batch, as in the reference
      red[0][0] = b;        // lane 0 alone holds the caller's normaliser: the other lanes' b is the local mask count
    }
  }
  if (dy) {
    __syncthreads();
    const float inv = 1.0f / red[0][0];
    // d/dp of -(t log(p+eps) + (1-t) log(1-p+eps)); NOT (p - t)/(p(1-p)) because of eps
    if (small) {
#pragma unroll
      for (int j = 0; j < BCE_PER; ++j)
        if (tid + 1024 * j < n)
          dy[tid + 1024 * j] = mr[j] * inv * (-(tr[j] / (pr[j] + eps)) + (1.0f - tr[j]) / (1.0f - pr[j] + eps));
    } else {
      for (int i = tid; i < n; i += 1024) {
        const float m = ids[i] != 0 ? 1.f : 0.f;
        const float t = (float)y_true[i];
        const float p = y[i];
        dy[i] = m * inv * (-(t / (p + eps)) + (1.0f - t) / (1.0f - p + eps));
      }
    }
  }
}

// one wave per user: rank of candidate 0 = number of strictly larger scores among candidates 1..N-1
__global__ void rank_kernel(const float* __restrict__ y, int B, int N, int k, const int32_t* __restrict__ pos,
                            int32_t* __restrict__ rank, float* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (u >= B) return;
  const float* yr = y + (size_t)u * N;
  const int pc = pos ? pos[u] : 0;  // column of the positive (0 in the reference's datasets, data.py:165,190)
  const float y0 = yr[pc];
  float gt = 0.f, eq = 0.f;
  for (int j = lane; j < N; j += 64) {
    if (j == pc) continue;
    const float v = yr[j];
    gt += v > y0 ? 1.f : 0.f;
    eq += v == y0 ? 1.f : 0.f;
  }
  gt = wave_sum(gt);
  eq = wave_sum(eq);
  if (lane == 0) {
    const int r = (int)gt;
    if (rank) rank[u] = r;
    if (r < k) {
      atomicAdd(&sums[0], 1.0f);
      atomicAdd(&sums[1], 1.0f / log2f((float)r + 2.0f));
    }
    if (eq > 0.f) atomicAdd(&sums[2], eq);
  }
}

// The same sums in a FIXED order (deterministic mode, carca_set_tuning(8, 1)): one block, wave w takes users w, w + 16, ...
// in turn, the sixteen partial sums are added in wave order and the block adds ONE term to each of the caller's sums --
// the NDCG sum of the kernel above follows the order in which the waves' atomics arrive (last-bit differences run to run).
__global__ __launch_bounds__(1024) void rank_ordered_kernel(const float* __restrict__ y, int B, int N, int k,
                                                            const int32_t* __restrict__ pos, int32_t* __restrict__ rank,
                                                            float* __restrict__ sums) {
  __shared__ float red[3][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float hr = 0.f, nd = 0.f, ties = 0.f;
  for (int u = wave; u < B; u += 16) {
    const float* yr = y + (size_t)u * N;
    const int pc = pos ? pos[u] : 0;
    const float y0 = yr[pc];
    float gt = 0.f, eq = 0.f;
    for (int j = lane; j < N; j += 64) {
      if (j == pc) continue;
      const float v = yr[j];
      gt += v > y0 ? 1.f : 0.f;
      eq += v == y0 ? 1.f : 0.f;
    }
    gt = wave_sum(gt);
    eq = wave_sum(eq);
    const int r = (int)gt;
    if (rank && lane == 0) rank[u] = r;
    if (r < k) {
      hr += 1.0f;
      nd += 1.0f / log2f((float)r + 2.0f);
    }
    ties += eq;
  }
  if (lane == 0) {
    red[0][wave] = hr;
    red[1][wave] = nd;
    red[2][wave] = ties;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[threadIdx.x][w];
    if (s != 0.f) sums[threadIdx.x] += s;  // (launches on one stream are ordered: a plain read-modify-write)
  }
}

// One evaluation batch's metrics in ONE launch (train.py:45-51: BCE loss, compute_HR, compute_NDCG): sums[0..2] as
// rank_ordered_kernel (fixed order), sums[3] += the batch's masked-mean BCE loss (bce_kernel's arithmetic and order),
// sums[4] += B.  The evaluation loop issued a rank kernel, a loss kernel and two one-element adds per batch -- ~15 us of
// launches behind a 610 us forward.  n = B N <= 16384.
__global__ __launch_bounds__(1024) void eval_metrics_kernel(const float* __restrict__ y, const int32_t* __restrict__ y_true,
                                                            const int32_t* __restrict__ ids, int B, int N, int k, float eps,
                                                            float* __restrict__ sums) {
  __shared__ float red[5][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = B * N;
  float pr[BCE_PER], tr[BCE_PER], mr[BCE_PER];
#pragma unroll
  for (int j = 0; j < BCE_PER; ++j) {
    const int i = min(tid + 1024 * j, n - 1);
    pr[j] = y[i];
    tr[j] = (float)y_true[i];
    mr[j] = ids[i] != 0 ? 1.f : 0.f;
  }
  float hr = 0.f, nd = 0.f, ties = 0.f;
  constexpr int UPW = 12;  // users per wave whose scores are all requested before the first is looked at (N <= 128)
  if (N <= 128 && B <= 16 * UPW) {
    // (the reference's evaluation shape, 1 + 100 candidates: a wave's <= 12 users are 24 loads in flight instead of a chain
    // of 12 dependent round trips)
    float v0[UPW], v1[UPW], y0[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const int u = min(wave + 16 * i, B - 1);
      const float* yr = y + (size_t)u * N;
      y0[i] = yr[0];
      v0[i] = yr[min(1 + lane, N - 1)];
      v1[i] = yr[min(65 + lane, N - 1)];
    }
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      const bool in0 = 1 + lane < N, in1 = 65 + lane < N;
      float gt = (in0 && v0[i] > y0[i] ? 1.f : 0.f) + (in1 && v1[i] > y0[i] ? 1.f : 0.f);
      float eq = (in0 && v0[i] == y0[i] ? 1.f : 0.f) + (in1 && v1[i] == y0[i] ? 1.f : 0.f);
      gt = wave_sum(gt);
      eq = wave_sum(eq);
      if (wave + 16 * i < B) {
        const int r = (int)gt;
        if (r < k) {
          hr += 1.0f;
          nd += 1.0f / log2f((float)r + 2.0f);
        }
        ties += eq;
      }
    }
  } else {
    for (int u = wave; u < B; u += 16) {  // (the positive is candidate 0: data.py:165,190)
      const float* yr = y + (size_t)u * N;
      const float y0 = yr[0];
      float gt = 0.f, eq = 0.f;
      for (int j = 1 + lane; j < N; j += 64) {
        const float v = yr[j];
        gt += v > y0 ? 1.f : 0.f;
        eq += v == y0 ? 1.f : 0.f;
      }
      gt = wave_sum(gt);
      eq = wave_sum(eq);
      const int r = (int)gt;
      if (r < k) {
        hr += 1.0f;
        nd += 1.0f / log2f((float)r + 2.0f);
      }
      ties += eq;
    }
  }
  float sl = 0.f, sm = 0.f;
#pragma unroll
  for (int j = 0; j < BCE_PER; ++j) {
    const float l = -(tr[j] * logf(pr[j] + eps) + (1.0f - tr[j]) * logf(1.0f - pr[j] + eps));
    if (tid + 1024 * j < n) {
      sl += l * mr[j];
      sm += mr[j];
    }
  }
  sl = wave_sum(sl);
  sm = wave_sum(sm);
  if (lane == 0) {
    red[0][wave] = hr;
    red[1][wave] = nd;
    red[2][wave] = ties;
    red[3][wave] = sl;
    red[4][wave] = sm;
  }
  __syncthreads();
  if (wave == 0) {
    float a = lane < 16 ? red[3][lane] : 0.f, b = lane < 16 ? red[4][lane] : 0.f;  // (bce_kernel's order: a wave reduction of the sixteen)
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane < 3) {
      float s = 0.f;
      for (int w = 0; w < 16; ++w) s += red[lane][w];
      if (s != 0.f) sums[lane] += s;
    }
    if (lane == 3) sums[3] += a / b;
    if (lane == 4) sums[4] += (float)B;
  }
}

}  // namespace

extern "C" int carca_eval_metrics(const float* y, const int32_t* y_true, const int32_t* ids, int B, int N, int k, float eps,
                                  float* sums, void* stream_) {
  CARCA_CHECK_ARG(y && y_true && ids && sums && B >= 1 && N >= 1 && k >= 1, "eval_metrics: null pointer or bad dims");
  CARCA_CHECK_SUPPORTED((long)B * N <= 1024L * BCE_PER, "eval_metrics: B x N = %ld > %d (use carca_rank_metrics + carca_bce_fwd)",
                        (long)B * N, 1024 * BCE_PER);
  hipLaunchKernelGGL(eval_metrics_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream_, y, y_true, ids, B, N, k, eps, sums);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_bce_fwd(const float* y, const int32_t* y_true, const int32_t* ids, int n, float eps,
                             float* scratch, float* loss_out, float* dy, const float* denom, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(y && y_true && ids && scratch && loss_out && n >= 1, "bce_fwd: null pointer or n < 1");
  hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(1024), 0, stream, y, y_true, ids, n, eps, scratch, loss_out, dy, denom);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}

extern "C" int carca_rank_metrics(const float* y, int B, int N, int k, const int32_t* pos, int32_t* rank, float* sums,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  CARCA_CHECK_ARG(y && sums && B >= 1 && N >= 1 && k >= 1, "rank_metrics: null pointer or bad dims");
  if (carca_tuning(CARCA_TUNE_DETERMINISTIC) != 0) {
    hipLaunchKernelGGL(rank_ordered_kernel, dim3(1), dim3(1024), 0, stream, y, B, N, k, pos, rank, sums);
    CARCA_LAUNCH_CHECK();
    return CARCA_OK;
  }
  const int blocks = (B + 3) / 4;
  hipLaunchKernelGGL(rank_kernel, dim3(blocks), dim3(256), 0, stream, y, B, N, k, pos, rank, sums);
  CARCA_LAUNCH_CHECK();
  return CARCA_OK;
}
