// Microbench: sustained v_mfma_f32_32x32x2_f32 rate at the feat GEMM's occupancy (no memory traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
    if (rep == 2) printf("NACC=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, ms, flops / ms / 1e9);
  }
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  run<3>(256, 2000, out);
  run<3>(512, 2000, out);
  run<3>(755, 2000, out);
  run<3>(768, 2000, out);
  run<1>(768, 6000, out);
  run<4>(256, 2000, out);
  return 0;
}
