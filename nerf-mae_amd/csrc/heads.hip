// Kernels of the dense-prediction heads fine-tuned on the MAE-pretrained encoder + decoder:
// SwinTransformer_VoxelSR_Pretrained_Skip / SwinTransformer_VoxelSemantics_Pretrained_Skip (nerf_rpn/model/feature_extractor.py:
// 1898-2244, 2521-2848).  The convolutions, norms and GEMMs of both heads are the kernels of the MAE path; what is new here is data
// movement around them and the two losses:
//   grid -> 8-channel channels-last input of `encoder1` (4 data channels + 4 zero channels: the GEMM engine contracts in units of 8);
//   nearest-neighbour upsampling of the head output with the channels-last -> NCDHW transposition folded in, and its adjoint
//     (nn.Upsample(scale_factor) followed by a 1x1x1 conv == the 1x1x1 conv followed by the upsampling: the head GEMM runs at the
//     decoder resolution, 4.1x fewer voxels for 160 -> 256);
//   masked RGB MSE of the super-resolution head and the masked, class-weighted cross entropy (+ soft-IoU sums) of the semantics head.
#include "common.hpp"
#include "kernels.hpp"

static inline unsigned hd_blocks(long n) { long b = (n + 255) / 256; return (unsigned)(b > 65535L * 16 ? 65535L * 16 : (b < 1 ? 1 : b)); }

// ---- (B,4,V) fp32 NCDHW grid -> [B*V][8] channels-last in the compute dtype, channels 4..7 zero --------------------------------
template <typename T> __global__ void grid_to_cl8_kernel(const float* __restrict__ src, T* __restrict__ dst, long V, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / V, v = i - b * V;
    const float* s = src + b * 4 * V + v;
    float o[8] = {s[0], s[V], s[2 * V], s[3 * V], 0.f, 0.f, 0.f, 0.f};
    Vec8<T>::store(dst + i * 8, o);
  }
}
int k_grid_to_cl8(int dt, const float* src, void* dst, int B, long V, hipStream_t st) {
  const long total = (long)B * V;
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(grid_to_cl8_kernel<bf16_t>, dim3(hd_blocks(total)), dim3(256), 0, st, src, (bf16_t*)dst, V, total);
  else hipLaunchKernelGGL(grid_to_cl8_kernel<float>, dim3(hd_blocks(total)), dim3(256), 0, st, src, (float*)dst, V, total);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ATen's nearest source index for an explicit scale_factor (upsample_nearest3d with scales): min(floorf(dst * (float)(1/scale)), in - 1)
__device__ __forceinline__ int nn_src(int d, float inv_scale, int in) {
  const int s = (int)floorf((float)d * inv_scale);
  return s < in - 1 ? s : in - 1;
}

// ---- head output: src [B*R^3][Cp] (compute dtype, channels-last, first Co columns valid) -> pred (B,Co,Ro,Ro,Ro) fp32, nearest ----
template <typename T> __global__ void cl_to_ncdhw_up_kernel(const T* __restrict__ src, float* __restrict__ dst, int Co, int Cp, int R, int Ro, float inv_scale, long total) {
  const long Vo = (long)Ro * Ro * Ro;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {   // i over (b, zo, yo, xo)
    const long b = i / Vo, vo = i - b * Vo;
    const int xo = (int)(vo % Ro), yo = (int)((vo / Ro) % Ro), zo = (int)(vo / ((long)Ro * Ro));
    const long row = ((b * R + nn_src(zo, inv_scale, R)) * R + nn_src(yo, inv_scale, R)) * R + nn_src(xo, inv_scale, R);
    const T* s = src + row * Cp;
    for (int c = 0; c < Co; ++c) dst[(b * Co + c) * Vo + vo] = to_f<T>(s[c]);
  }
}
int k_cl_to_ncdhw_up(int dt, const void* src, float* dst, int B, int Co, int Cp, int R, int Ro, float inv_scale, hipStream_t st) {
  const long total = (long)B * Ro * Ro * Ro;
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(cl_to_ncdhw_up_kernel<bf16_t>, dim3(hd_blocks(total)), dim3(256), 0, st, (const bf16_t*)src, dst, Co, Cp, R, Ro, inv_scale, total);
  else hipLaunchKernelGGL(cl_to_ncdhw_up_kernel<float>, dim3(hd_blocks(total)), dim3(256), 0, st, (const float*)src, dst, Co, Cp, R, Ro, inv_scale, total);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- adjoint: g[B*R^3][Cp] = sum over the output voxels that read each source voxel of dpred (B,Co,Ro^3) fp32; columns >= Co zero ----
__device__ __forceinline__ void nn_copies(int s, float inv_scale, int in, int out, int& lo, int& hi) {
  // [lo, hi): the output indices d with nn_src(d) == s.  Start from the real-valued estimate and correct for float rounding.
  int d = (int)ceilf((float)s / inv_scale);
  if (d > out) d = out;
  while (d > 0 && nn_src(d - 1, inv_scale, in) >= s) --d;
  while (d < out && nn_src(d, inv_scale, in) < s) ++d;
  lo = d;
  while (d < out && nn_src(d, inv_scale, in) == s) ++d;
  hi = d;
}
template <typename T> __global__ void ncdhw_up_adjoint_kernel(const float* __restrict__ dpred, T* __restrict__ g, int Co, int Cp, int R, int Ro, float inv_scale, long total) {
  const long Vo = (long)Ro * Ro * Ro, V = (long)R * R * R;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {   // i over (b, z, y, x) source voxels
    const long b = i / V, v = i - b * V;
    const int x = (int)(v % R), y = (int)((v / R) % R), z = (int)(v / ((long)R * R));
    int z0, z1, y0, y1, x0, x1;
    nn_copies(z, inv_scale, R, Ro, z0, z1);
    nn_copies(y, inv_scale, R, Ro, y0, y1);
    nn_copies(x, inv_scale, R, Ro, x0, x1);
    T* o = g + i * Cp;
    for (int c = 0; c < Cp; ++c) {
      float s = 0.f;
      if (c < Co) {
        const float* p = dpred + (b * Co + c) * Vo;
        for (int zo = z0; zo < z1; ++zo)
          for (int yo = y0; yo < y1; ++yo)
            for (int xo = x0; xo < x1; ++xo) s += p[((long)zo * Ro + yo) * Ro + xo];
      }
      o[c] = from_f<T>(s);
    }
  }
}
int k_ncdhw_up_adjoint(int dt, const float* dpred, void* g, int B, int Co, int Cp, int R, int Ro, float inv_scale, hipStream_t st) {
  const long total = (long)B * R * R * R;
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(ncdhw_up_adjoint_kernel<bf16_t>, dim3(hd_blocks(total)), dim3(256), 0, st, dpred, (bf16_t*)g, Co, Cp, R, Ro, inv_scale, total);
  else hipLaunchKernelGGL(ncdhw_up_adjoint_kernel<float>, dim3(hd_blocks(total)), dim3(256), 0, st, dpred, (float*)g, Co, Cp, R, Ro, inv_scale, total);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- dst[m][0:C] += src[m][0:C] (fp32, row strides in elements): un-padding of weight gradients computed on padded operands ----
__global__ void add_cols_f32_kernel(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long M, int C) {
  const long n = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C;
    const int c = (int)(i - m * C);
    dst[m * ldd + c] += src[m * lds + c];
  }
}
int k_add_cols_f32(const float* src, long lds, float* dst, long ldd, long M, int C, hipStream_t st) {
  hipLaunchKernelGGL(add_cols_f32_kernel, dim3(hd_blocks(M * C)), dim3(256), 0, st, src, lds, dst, ldd, M, C);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- VoxelSR loss (feature_extractor.py:2133-2160): sums[0] = sum_v m (p_rgb - t_rgb)^2, sums[1] = sum_v m, m = t_alpha > 0.01 ----
__global__ __launch_bounds__(256) void sr_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, long V, long total, double* __restrict__ sums,
                                                      float* __restrict__ dpred, const double* __restrict__ sums_in, float gscale) {
  // dpred == nullptr: forward sums; else backward: dpred = gscale * 2 m (p - t) / sums_in[1] on the RGB planes, 0 on the alpha plane
  __shared__ float sh[8];
  float s0 = 0.f, s1 = 0.f;
  const float k = dpred ? gscale * 2.0f / (float)sums_in[1] : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / V, v = i - b * V;
    const float* p = pred + b * 4 * V + v;
    const float* t = tgt + b * 4 * V + v;
    const float m = t[3 * V] > 0.01f ? 1.f : 0.f;
    if (dpred) {
      float* d = dpred + b * 4 * V + v;
      d[0] = k * m * (p[0] - t[0]); d[V] = k * m * (p[V] - t[V]); d[2 * V] = k * m * (p[2 * V] - t[2 * V]); d[3 * V] = 0.f;
    } else {
      const float e0 = p[0] - t[0], e1 = p[V] - t[V], e2 = p[2 * V] - t[2 * V];
      s0 += m * (e0 * e0 + e1 * e1 + e2 * e2);
      s1 += m;
    }
  }
  if (dpred) return;
  s0 = wave_sum(s0); s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sh[(threadIdx.x >> 6) * 2] = s0; sh[(threadIdx.x >> 6) * 2 + 1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums, (double)(sh[0] + sh[2] + sh[4] + sh[6]));
    atomicAdd(sums + 1, (double)(sh[1] + sh[3] + sh[5] + sh[7]));
  }
}
__global__ void sr_loss_finalize_kernel(const double* sums, float* loss) { loss[0] = (float)(sums[0] / sums[1]); }
int k_sr_loss_fwd(const float* pred, const float* tgt, int B, long V, double* sums, float* loss, hipStream_t st) {
  hipError_t e = nmh_zero_async(sums, 2 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  const long total = (long)B * V;
  unsigned nb = hd_blocks(total);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(sr_loss_kernel, dim3(nb), dim3(256), 0, st, pred, tgt, V, total, sums, (float*)nullptr, (const double*)nullptr, 0.f);
  hipLaunchKernelGGL(sr_loss_finalize_kernel, dim3(1), dim3(1), 0, st, sums, loss);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_sr_loss_bwd(const float* pred, const float* tgt, int B, long V, const double* sums, float gscale, float* dpred, hipStream_t st) {
  const long total = (long)B * V;
  hipLaunchKernelGGL(sr_loss_kernel, dim3(hd_blocks(total)), dim3(256), 0, st, pred, tgt, V, total, (double*)nullptr, dpred, sums, gscale);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- VoxelSemantics loss (metrics.py:540-553 + nn.CrossEntropyLoss(weight); soft-IoU sums of mIoULoss_new, metrics.py:194-245) ----
// logits (B,K,V) fp32, labels (B,V) fp32 class ids (0 = unlabelled), cw [K] or null.  mask m = label > 0; the cross entropy runs over
// ALL voxels on (logits*m, label*m).  sums[0] = sum_v w nll, sums[1] = sum_v w; iou[b][k-1][3] = {sum_v m p_k, #(label == k), sum_{label==k} p_k}.
template <int KMAX>
__global__ __launch_bounds__(256) void masked_ce_kernel(const float* __restrict__ logits, const float* __restrict__ labels, const float* __restrict__ cw, int K, long V,
                                                        double* __restrict__ sums, double* __restrict__ iou, float* __restrict__ dlogits, const double* __restrict__ sums_in,
                                                        float gscale) {
  extern __shared__ float shm[];      // [2 + 3*(K-1)]
  const int b = blockIdx.y;
  const int nacc = 2 + 3 * (K - 1);
  if (!dlogits) {
    for (int i = threadIdx.x; i < nacc; i += 256) shm[i] = 0.f;
    __syncthreads();
  }
  const float inv_w = dlogits ? gscale / (float)sums_in[1] : 0.f;
  float s_nll = 0.f, s_w = 0.f;
  float sp[KMAX];                      // per-thread sums of m * p_k (256 threads adding to the same LDS words would serialise)
#pragma unroll
  for (int k = 0; k < KMAX; ++k) sp[k] = 0.f;
  const float* lg = logits + (long)b * K * V;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long)gridDim.x * blockDim.x) {
    const int t = (int)labels[(long)b * V + v];
    const float m = t > 0 ? 1.f : 0.f;
    float l[KMAX], mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) { l[k] = lg[(long)k * V + v]; mx = fmaxf(mx, l[k] * m); }
    float se = 0.f, sraw = 0.f, mraw = -3.0e38f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) { se += __expf(l[k] * m - mx); mraw = fmaxf(mraw, l[k]); }
    const int tm = t > 0 ? t : 0;
    const float w = cw ? cw[tm] : 1.f;
    if (dlogits) {
      float* dl = dlogits + (long)b * K * V + v;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) dl[(long)k * V] = m * w * inv_w * (__expf(l[k] * m - mx) / se - (k == tm ? 1.f : 0.f));
      continue;
    }
    const float lse = mx + __logf(se);
    float lt = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && k == tm) lt = l[k] * m;
    s_nll += w * (lse - lt);
    s_w += w;
    // soft IoU terms use the softmax of the RAW logits (mIoULoss_new: F.softmax(inputs) * mask)
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) sraw += __expf(l[k] - mraw);
    if (t > 0) {
      const float rs = 1.f / sraw;
#pragma unroll
      for (int k = 1; k < KMAX; ++k)
        if (k < K) {
          const float p = __expf(l[k] - mraw) * rs;
          sp[k] += p;
          if (k == t) { atomicAdd(&shm[2 + (k - 1) * 3 + 1], 1.f); atomicAdd(&shm[2 + (k - 1) * 3 + 2], p); }   // one class per voxel: spread over K-1 words
        }
    }
  }
  if (dlogits) return;
  s_nll = wave_sum(s_nll); s_w = wave_sum(s_w);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&shm[0], s_nll); atomicAdd(&shm[1], s_w); }
#pragma unroll
  for (int k = 1; k < KMAX; ++k)
    if (k < K) {
      const float v = wave_sum(sp[k]);
      if ((threadIdx.x & 63) == 0) atomicAdd(&shm[2 + (k - 1) * 3], v);
    }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(sums + threadIdx.x, (double)shm[threadIdx.x]);
  for (int i = threadIdx.x; i < 3 * (K - 1); i += 256) atomicAdd(iou + (long)b * 3 * (K - 1) + i, (double)shm[2 + i]);
}
__global__ void masked_ce_finalize_kernel(const double* sums, const double* iou, int B, int K, float* out) {
  // out[0] = weighted CE; out[1] = mean over (sample, class >= 1) of inter / (union + 1e-8), union = sum m p + count - inter
  if (threadIdx.x == 0) {
    out[0] = (float)(sums[0] / sums[1]);
    double acc = 0.0;
    for (int i = 0; i < B * (K - 1); ++i) {
      const double sp = iou[i * 3], cnt = iou[i * 3 + 1], inter = iou[i * 3 + 2];
      acc += inter / (sp + cnt - inter + 1e-8);
    }
    out[1] = (float)(acc / (B * (K - 1)));
  }
}
int k_masked_ce_fwd(const float* logits, const float* labels, const float* cw, int B, int K, long V, double* sums, double* iou, float* out, hipStream_t st) {
  if (K < 2 || K > 32) return -2;
  hipError_t e = nmh_zero_async(sums, 2 * sizeof(double), st);
  if (e == hipSuccess) e = nmh_zero_async(iou, sizeof(double) * 3 * (K - 1) * B, st);
  if (e != hipSuccess) return (int)e;
  unsigned nb = hd_blocks(V);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(masked_ce_kernel<32>, dim3(nb, B), dim3(256), (2 + 3 * (K - 1)) * sizeof(float), st, logits, labels, cw, K, V, sums, iou, (float*)nullptr, (const double*)nullptr, 0.f);
  hipLaunchKernelGGL(masked_ce_finalize_kernel, dim3(1), dim3(64), 0, st, sums, iou, B, K, out);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_masked_ce_bwd(const float* logits, const float* labels, const float* cw, int B, int K, long V, const double* sums, float gscale, float* dlogits, hipStream_t st) {
  if (K < 2 || K > 32) return -2;
  hipLaunchKernelGGL(masked_ce_kernel<32>, dim3(hd_blocks(V), B), dim3(256), 0, st, logits, labels, cw, K, V, (double*)nullptr, (double*)nullptr, dlogits, sums, gscale);
  NMH_CHECK_LAUNCH();
  return 0;
}
