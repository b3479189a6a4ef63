// 3-D shifted-window attention core for 4x4x4 windows (64 tokens), head_dim 32  (reference:
// swin_mae3d.py:103-172 QK^T*scale + relative-position bias + shift mask(-100) -> softmax -> AV, and
// :200-211,257-280 for the bias index).  One wave per (window, head):
//   * head_dim 32 is exactly one MFMA k-step, so Q/K/V/dO fragments are 16-byte rows straight from HBM;
//   * S^T = K.Q^T is computed "swapped" so that the probabilities of one query sit in one lane column and the
//     C-layout registers ARE the A-operand of the P.V MFMA (slot map of common.hpp) -- P never leaves registers;
//   * V (and K, Q, dO in backward) are staged once in LDS as loaded and consumed through ds_read_b64_tr_b16;
//   * backward recomputes P from the saved log-sum-exp, builds both S and S^T layouts (2x16 cheap MFMAs) instead
//     of transposing through LDS, and reduces d(bias table) in LDS before 343 global atomics per wave.
// Pad tokens (zero rows -> qkv = bias) participate exactly as in the reference.
#include "common.hpp"
#include "kernels.hpp"

template <int X> struct OddRS32 { static constexpr int v = ((X + 31) / 32 % 2 == 1) ? (X + 31) / 32 * 32 : ((X + 31) / 32 + 1) * 32; };

template <typename T> __device__ __forceinline__ Frag<T> gfrag(const T* base, long ld, int row, int g);
template <> __device__ __forceinline__ Frag<bf16_t> gfrag<bf16_t>(const bf16_t* base, long ld, int row, int g) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(base + row * ld + 8 * g);
  return f;
}
template <> __device__ __forceinline__ Frag<float> gfrag<float>(const float* base, long ld, int row, int g) {
  Frag<float> f;
  float4 a = *reinterpret_cast<const float4*>(base + row * ld + 8 * g), b = *reinterpret_cast<const float4*>(base + row * ld + 8 * g + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
// pack two C-layout accumulators (4 regs each) into an A-operand fragment following the tr slot map
__device__ __forceinline__ Frag<bf16_t> pack_frag(const f32x4& lo, const f32x4& hi, bf16_t*) {
  Frag<bf16_t> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = (short)f2bf(lo[j]); f.v[4 + j] = (short)f2bf(hi[j]); }
  return f;
}
__device__ __forceinline__ Frag<float> pack_frag(const f32x4& lo, const f32x4& hi, float*) {
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = lo[j]; f.v[4 + j] = hi[j]; }
  return f;
}
__device__ __forceinline__ int relidx(int i, int j) {
  return ((i >> 4) - (j >> 4) + 3) * 49 + (((i >> 2) & 3) - ((j >> 2) & 3) + 3) * 7 + ((i & 3) - (j & 3) + 3);
}
__device__ __forceinline__ int axis_region(int p, int P, int s) { return s == 0 ? 0 : (p < P - 4 ? 0 : (p < P - s ? 1 : 2)); }
__device__ __forceinline__ int token_region(const WinMap& w, int winl, int t) {
  const int nwy = w.PW >> 2, nwx = w.PD >> 2;
  int wx = winl % nwx, wy = (winl / nwx) % nwy, wz = winl / (nwx * nwy);
  return axis_region(wz * 4 + (t >> 4), w.PH, w.s0) * 9 + axis_region(wy * 4 + ((t >> 2) & 3), w.PW, w.s1) * 3 + axis_region(wx * 4 + (t & 3), w.PD, w.s2);
}
// stage a [64][32] tile (row stride ld in global) into LDS rows of RS bytes
template <typename T, int RS> __device__ __forceinline__ void stage_tile(char* dst, const T* src, long ld, int lane) {
  constexpr int CPR = 32 * (int)sizeof(T) / 16, CE = 16 / (int)sizeof(T);
#pragma unroll
  for (int i = 0; i < CPR; ++i) {
    int it = lane + 64 * i, r = it / CPR, c = it - r * CPR;
    *reinterpret_cast<uint4*>(dst + r * RS + c * 16) = *reinterpret_cast<const uint4*>(src + r * ld + c * CE);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table, T* __restrict__ out, float* __restrict__ lse,
                                                        int heads, int C, WinMap wm, long npairs) {
  constexpr int RS = OddRS32<32 * (int)sizeof(T)>::v;
  __shared__ __attribute__((aligned(16))) char sV[4][64 * RS];
  __shared__ float sB[4][344];
  __shared__ int sR[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const long pair = (long)blockIdx.x * 4 + wave;
  if (pair >= npairs) return;
  const long win = pair / heads;
  const int h = (int)(pair - win * heads);
  const int nW = (wm.PH >> 2) * (wm.PW >> 2) * (wm.PD >> 2);
  const bool shifted = (wm.s0 + wm.s1 + wm.s2) > 0;
  const long ld = 3L * C;
  const T* qb = qkv + win * 64 * ld + h * 32;
  const T* kb = qb + C;
  const T* vb = qb + 2 * C;
  stage_tile<T, RS>(sV[wave], vb, ld, lane);
  for (int t = lane; t < 343; t += 64) sB[wave][t] = table[t * heads + h];
  sR[wave][lane] = shifted ? token_region(wm, (int)(win % nW), lane) : 0;
  Frag<T> kf[4], qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { kf[t] = gfrag<T>(kb, ld, t * 16 + li, g); qf[t] = gfrag<T>(qb, ld, t * 16 + li, g); }
  f32x4 s[4][4];  // [jt][it]: key j = 16jt+4g+r, query i = 16it+li
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < 4; ++it) { s[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(s[jt][it], kf[jt], qf[it]); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float scale = 0.17677669529663689f;  // 32^-0.5
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = 16 * it + li;
    const int ri = sR[wave][i];
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * g + r;
        float v = s[jt][it][r] * scale + sB[wave][relidx(i, j)];
        if (shifted && sR[wave][j] != ri) v += -100.0f;
        s[jt][it][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { float e = __expf(s[jt][it][r] - mx); s[jt][it][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[jt][it][r] *= inv;
    if (g == 0) lse[pair * 64 + i] = mx + __logf(sum);
  }
  // O = P.V
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag<T> pa = pack_frag(s[2 * ks][it], s[2 * ks + 1][it], (T*)nullptr);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        Frag<T> vfr = lds_frag_t(sV[wave], RS, ks * 32, dt * 16, lane, (T*)nullptr);
        mma(o[dt], pa, vfr);
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(win * 64 + 16 * it + 4 * g + r) * C + h * 32 + 16 * dt + li] = from_f<T>(o[dt][r]);
  }
}

int k_attn_fwd(int dt, const void* qkv, const float* table, void* out, float* lse, int heads, int C, const WinMap& wm, hipStream_t st) {
  if (C != heads * 32) return -2;
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  const long npairs = nwin * heads;
  dim3 grid((unsigned)((npairs + 3) / 4));
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, table, (bf16_t*)out, lse, heads, C, wm, npairs);
  else hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, table, (float*)out, lse, heads, C, wm, npairs);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table, const T* __restrict__ dout, const float* __restrict__ lse,
                                                        T* __restrict__ dqkv, float* __restrict__ dtable, int heads, int C, WinMap wm, long nwin) {
  constexpr int RS = OddRS32<32 * (int)sizeof(T)>::v;
  __shared__ __attribute__((aligned(16))) char sQ[NW][64 * RS];
  __shared__ __attribute__((aligned(16))) char sK[NW][64 * RS];
  __shared__ __attribute__((aligned(16))) char sO[NW][64 * RS];
  __shared__ float sB[NW][344], sDB[NW][344], sD[NW][64], sL[NW][64];
  // d(bias) of one (query,key) pair always comes from the same lane/register, so it is accumulated over all windows of this
  // wave in a private 64x64 LDS tile (conflict-free adds) and folded into the 343 bins once at the end (was: 4096 colliding
  // LDS atomics per window)
  __shared__ float sDS[NW][64 * 64];
  __shared__ int sR[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  const int h = blockIdx.y;
  const int nW = (wm.PH >> 2) * (wm.PW >> 2) * (wm.PD >> 2);
  const bool shifted = (wm.s0 + wm.s1 + wm.s2) > 0;
  const long ld = 3L * C;
  const float scale = 0.17677669529663689f;
  for (int t = lane; t < 344; t += 64) { sB[wave][t] = t < 343 ? table[t * heads + h] : 0.f; sDB[wave][t] = 0.f; }
  for (int t = lane; t < 64 * 64; t += 64) sDS[wave][t] = 0.f;

  for (long win = (long)blockIdx.x * NW + wave; win < nwin; win += (long)gridDim.x * NW) {
    const T* qb = qkv + win * 64 * ld + h * 32;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = dout + win * 64 * C + h * 32;
    T* dqb = dqkv + win * 64 * ld + h * 32;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stage_tile<T, RS>(sQ[wave], qb, ld, lane);
    stage_tile<T, RS>(sK[wave], kb, ld, lane);
    stage_tile<T, RS>(sO[wave], dob, (long)C, lane);
    sR[wave][lane] = shifted ? token_region(wm, (int)(win % nW), lane) : 0;
    sL[wave][lane] = lse[(win * heads + h) * 64 + lane];
    Frag<T> kf[4], qf[4], vf[4], df[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kf[t] = gfrag<T>(kb, ld, t * 16 + li, g); qf[t] = gfrag<T>(qb, ld, t * 16 + li, g);
      vf[t] = gfrag<T>(vb, ld, t * 16 + li, g); df[t] = gfrag<T>(dob, (long)C, t * 16 + li, g);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase A: S^T layout (lane <-> query) : D_i, d(bias), dQ ----------------
    {
      f32x4 p[4][4], dp[4][4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          p[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(p[jt][it], kf[jt], qf[it]);
          dp[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(dp[jt][it], vf[jt], df[it]);
        }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int i = 16 * it + li;
        const int ri = sR[wave][i];
        const float L = sL[wave][i];
        float dsum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            float v = p[jt][it][r] * scale + sB[wave][relidx(i, j)];
            if (shifted && sR[wave][j] != ri) v += -100.0f;
            const float e = __expf(v - L);
            p[jt][it][r] = e;
            dsum += e * dp[jt][it][r];
          }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        if (g == 0) sD[wave][i] = dsum;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            const float ds = p[jt][it][r] * (dp[jt][it][r] - dsum);
            dp[jt][it][r] = ds;
            sDS[wave][j * 64 + i] += ds;  // lanes li = consecutive i: conflict-free
          }
        // dQ rows of this query tile: sum_j dS[i][j] K[j][d]
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Frag<T> a = pack_frag(dp[2 * ks][it], dp[2 * ks + 1][it], (T*)nullptr);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            Frag<T> b = lds_frag_t(sK[wave], RS, ks * 32, dt * 16, lane, (T*)nullptr);
            mma(o[dt], a, b);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dqb[(16 * it + 4 * g + r) * ld + 16 * dt + li] = from_f<T>(o[dt][r] * scale);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase B: S layout (lane <-> key) : dV, dK ----------------
    {
      f32x4 p[4][4], dp[4][4];  // [it][jt]: query i = 16it+4g+r, key j = 16jt+li
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          p[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(p[it][jt], qf[it], kf[jt]);
          dp[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(dp[it][jt], df[it], vf[jt]);
        }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int j = 16 * jt + li;
        const int rj = sR[wave][j];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * it + 4 * g + r;
            float v = p[it][jt][r] * scale + sB[wave][relidx(i, j)];
            if (shifted && sR[wave][i] != rj) v += -100.0f;
            const float e = __expf(v - sL[wave][i]);
            p[it][jt][r] = e;
            dp[it][jt][r] = e * (dp[it][jt][r] - sD[wave][i]);
          }
        f32x4 ov[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, ok[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Frag<T> ap = pack_frag(p[2 * ks][jt], p[2 * ks + 1][jt], (T*)nullptr);
          Frag<T> ad = pack_frag(dp[2 * ks][jt], dp[2 * ks + 1][jt], (T*)nullptr);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            Frag<T> bo = lds_frag_t(sO[wave], RS, ks * 32, dt * 16, lane, (T*)nullptr);
            mma(ov[dt], ap, bo);
            Frag<T> bq = lds_frag_t(sQ[wave], RS, ks * 32, dt * 16, lane, (T*)nullptr);
            mma(ok[dt], ad, bq);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long ro = (long)(16 * jt + 4 * g + r) * ld + 16 * dt + li;
            dqb[ro + 2 * C] = from_f<T>(ov[dt][r]);
            dqb[ro + C] = from_f<T>(ok[dt][r] * scale);
          }
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int t = lane; t < 64 * 64; t += 64) atomicAdd(&sDB[wave][relidx(t & 63, t >> 6)], sDS[wave][t]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int t = lane; t < 343; t += 64) atomicAdd(dtable + t * heads + h, sDB[wave][t]);
}

int k_attn_bwd(int dt, const void* qkv, const float* table, const void* dout, const float* lse, void* dqkv, float* dtable, int heads, int C, const WinMap& wm, hipStream_t st) {
  if (C != heads * 32) return -2;
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  const int nw = dt == NMH_DT_BF16 ? 2 : 1;
  long gx = (nwin + nw * 4 - 1) / (nw * 4);
  long cap = 2048 / heads;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  dim3 grid((unsigned)gx, heads);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, 2>), grid, dim3(128), 0, st, (const bf16_t*)qkv, table, (const bf16_t*)dout, lse, (bf16_t*)dqkv, dtable, heads, C, wm, nwin);
  else hipLaunchKernelGGL((attn_bwd_kernel<float, 1>), grid, dim3(64), 0, st, (const float*)qkv, table, (const float*)dout, lse, (float*)dqkv, dtable, heads, C, wm, nwin);
  NMH_CHECK_LAUNCH();
  return 0;
}
