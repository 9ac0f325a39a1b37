// Microbench: do v_mfma_f32_16x16x4_f32 / 32x32x2_f32 (fp32-input MFMA) and plain fp32 VALU work of ANOTHER wave on the same SIMD
// overlap?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run v_fma_f32 / v_exp_f32 / v_max.
// Three launches: MFMA waves only, VALU waves only, both.  If fp32 MFMA had a pipe of its own, "both" would take
// max(t_mfma, t_valu); if it runs on the vector ALUs, the sum.  (bf16 MFMA row for comparison.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>  // 0: 16x16x4 f32, 1: 32x32x2 f32, 2: 16x16x32 bf16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int n_mfma, int n_valu, int vkind, float a0) {
  const int wave = threadIdx.x >> 6;
  float a = a0 + threadIdx.x * 1e-3f, b = 0.25f + threadIdx.x * 2e-3f;
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if constexpr (KIND == 0) {
      f32x4 acc[4] = {};
      for (int it = 0; it < n_mfma / 16; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    } else if constexpr (KIND == 1) {
      f32x16 acc[2] = {};
      for (int it = 0; it < n_mfma / 16; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][15];
    } else {
      f32x4 acc[4] = {};
      bf16x8 va, vb;
      for (int j = 0; j < 8; ++j) { va[j] = (__bf16)(a + j); vb[j] = (__bf16)(b - j); }
      for (int it = 0; it < n_mfma / 16; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[i], 0, 0, 0);
      for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    }
  } else {
    float x0 = a, x1 = b, x2 = a + b, x3 = a - b, x4 = a * 2, x5 = b * 2, x6 = a * 3, x7 = b * 3;
    for (int it = 0; it < n_valu / 16; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (vkind == 0) {
          x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
          x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (vkind == 1) {
          x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_exp2f(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_exp2f(x3);
          x4 = __builtin_amdgcn_exp2f(x4); x5 = __builtin_amdgcn_exp2f(x5); x6 = __builtin_amdgcn_exp2f(x6); x7 = __builtin_amdgcn_exp2f(x7);
        } else {
          int i0 = __float_as_int(x0), i1 = __float_as_int(x1), i2 = __float_as_int(x2), i3 = __float_as_int(x3);
          int i4 = __float_as_int(x4), i5 = __float_as_int(x5), i6 = __float_as_int(x6), i7 = __float_as_int(x7);
          i0 = (i0 ^ i1) + 3; i1 = (i1 ^ i2) + 5; i2 = (i2 ^ i3) + 7; i3 = (i3 ^ i4) + 9; i4 = (i4 ^ i5) + 11; i5 = (i5 ^ i6) + 13; i6 = (i6 ^ i7) + 15; i7 = (i7 ^ i0) + 17;
          x0 = __int_as_float(i0); x1 = __int_as_float(i1); x2 = __int_as_float(i2); x3 = __int_as_float(i3);
          x4 = __int_as_float(i4); x5 = __int_as_float(i5); x6 = __int_as_float(i6); x7 = __int_as_float(i7);
        }
      }
    }
    s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int KIND>
void run(const char* name, int nm, int nv, int vkind, float* out, unsigned long long* cyc) {
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, nm, nv, vkind, 0.5f);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, cyc, nm, nv, vkind, 0.5f);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-28s n_mfma=%6d n_valu=%6d vkind=%d: MFMA wave %8llu cycles, VALU wave %8llu cycles\n", name, nm, nv, vkind, h[0], h[4]);
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  unsigned long long* cyc; (void)hipMalloc(&cyc, 256 * 8 * 8);
  const char* vk[3] = {"v_fma_f32", "v_exp_f32", "int xor/add"};
  for (int vkind = 0; vkind < 3; ++vkind) {
    printf("--- VALU kind: %s\n", vk[vkind]);
    run<0>("f32 16x16x4  mfma only", 16000, 0, vkind, out, cyc);
    run<0>("f32 16x16x4  valu only", 0, 64000, vkind, out, cyc);
    run<0>("f32 16x16x4  both", 16000, 64000, vkind, out, cyc);
    run<0>("f32 16x16x4  both (2x valu)", 16000, 128000, vkind, out, cyc);
    run<1>("f32 32x32x2  mfma only", 8000, 0, vkind, out, cyc);
    run<1>("f32 32x32x2  both", 8000, 64000, vkind, out, cyc);
    run<2>("bf16 16x16x32 mfma only", 32000, 0, vkind, out, cyc);
    run<2>("bf16 16x16x32 both", 32000, 64000, vkind, out, cyc);
  }
  return 0;
}
