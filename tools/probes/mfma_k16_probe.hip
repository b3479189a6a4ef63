// issue rate of v_mfma_f32_16x16x16_bf16 against v_mfma_f32_16x16x32_bf16 on gfx950 (is a half-K step half the time?): one wave per SIMD,
// 8 independent accumulators, 4096 MFMAs each.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_k16_probe mfma_k16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int K> __global__ __launch_bounds__(256) void k(float* out, int n) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a8, b8; bf16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (short)(threadIdx.x + i); b8[i] = (short)(threadIdx.x * 3 + i); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (K == 32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 4096;
  for (int K : {32, 16, 32, 16}) {
    if (K == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, 16); else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (K == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, out, n); else hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 256.0 * 4 * n * 8;
    printf("16x16x%d bf16: %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA and SIMD, %.0f TFLOP/s\n", K, ms, n * 8, ms * 1e6 / (n * 8), mf * 16 * 16 * K * 2 / ms / 1e9);
  }
  return 0;
}
