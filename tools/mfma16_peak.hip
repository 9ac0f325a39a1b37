// Microbench: sustained v_mfma_f32_16x16x4_f32 rate (the attention kernels' instruction) for 1..4 independent
// accumulators per wave and 1..4 waves per SIMD, no memory traffic; cycles per MFMA per SIMD from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks, int threads, int iters, float* out, unsigned long long* cyc) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f, 0.25f);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.5f, 0.25f);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)blocks * (threads / 64) * iters * 16.0 * NACC * 2048.0 / ms / 1e9;
  unsigned long long h[1];
  (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
  const int waves_per_simd = threads / 256;
  const double per = (double)h[0] / ((double)iters * 16 * NACC * waves_per_simd);
  printf("acc=%d waves/SIMD=%d: %.1f cycles per MFMA per SIMD by s_memtime; wall %.3f ms = %.1f TFLOP/s\n", NACC,
         waves_per_simd, per, ms, tf);
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
  run<1>(256, 256, 2000, out, cyc);
  run<2>(256, 256, 2000, out, cyc);
  run<4>(256, 256, 2000, out, cyc);
  run<1>(256, 512, 2000, out, cyc);
  run<1>(256, 768, 2000, out, cyc);
  run<1>(256, 1024, 2000, out, cyc);
  run<4>(256, 1024, 1000, out, cyc);
  return 0;
}
