// sustained MFMA rate under the socket power limit: v_mfma_f32_16x16x32_bf16 (12 accumulators per wave, conv48's shape) against
// v_mfma_f32_32x32x16_bf16 (3 accumulators of 16 registers: same accumulator footprint, a quarter of the operand-register reads per flop).
// Operands are N(0,1) bf16 fragments that rotate every instruction.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[8], b[6];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(i * 64 + lane)];
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = src[((8 + i) * 64 + lane)];
  float sum = 0.f;
  if (MODE == 0) {
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)          // 8 x 12 MFMAs, operands rotate
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[m * 3 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(m + s) & 7], b[(n + s) % 6], acc[m * 3 + n], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)          // 8 x 6 MFMAs of twice the flops = the same work per iteration
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int n = 0; n < 3; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(2 * q + s) & 7], b[(n + s + q) % 6], acc[n], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += acc[i][j];
  }
  if (sum == 12345.678f) out[0] = sum;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  std::vector<unsigned short> h(14 * 64 * 8);
  srand(1);
  for (auto& v : h) {   // N(0,1) as bf16
    double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    float f = (float)(sqrt(-2 * log(u1)) * cos(6.283185307 * u2));
    unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16);
  }
  bf16x8* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, o, iters); else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, o, iters); };
  launch(); hipDeviceSynchronize();
  const double flops = 2.0 * 16 * 16 * 32 * 96.0 * iters * 8 * blocks;   // per launch
  double elapsed = 0; int n = 0; float last = 0;
  while (elapsed < secs) {
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); elapsed += ms * 1e-3; n += 20; last = ms / 20;
  }
  printf("mode %d (%s): last %.3f ms/launch = %.1f TFLOP/s after %.1f s\n", mode, mode ? "32x32x16" : "16x16x32", last, flops / last * 1e-9, elapsed);
  return 0;
}
