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
  const unsigned w0 = pk_bf16(lo[0], lo[1]), w1 = pk_bf16(lo[2], lo[3]), w2 = pk_bf16(hi[0], hi[1]), w3 = pk_bf16(hi[2], hi[3]);
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v u = {w0, w1, w2, w3};
  f.v = __builtin_bit_cast(bf16x8, u);
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
// NIT: query tiles (16 rows) per wave.  NIT = 4 is one wave per (window, head); with few windows (1-2 grids per GPU: 324 / 648 pairs on
// 1024 SIMDs) a pair is split over 4 / NIT waves, each with its own copy of K / V and NIT of the four query tiles: the kernel is one
// dependent chain per wave, and the chain gets shorter
template <typename T, int NIT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table, T* __restrict__ out, float* __restrict__ lse,
                                                        int heads, int C, WinMap wm, long npairs, int tok_out) {
  // tok_out: `out` is TOKEN-ordered [T][C] -- the row of a window goes to its token, pad rows are dropped (the proj GEMM and its weight gradient then run on
  // the real tokens: nmh_window_attn_fwd_tokens)
  constexpr int RS = OddRS32<32 * (int)sizeof(T)>::v;
  __shared__ __attribute__((aligned(16))) char sV[4][64 * RS];
  __shared__ float sB[4][344];
  __shared__ int sR[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
  constexpr int QS = 4 / NIT;
  const long unit = (long)blockIdx.x * 4 + wave;
  const long pair = unit / QS;
  const int it0 = (int)(unit % QS) * NIT;
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
  Frag<T> kf[4], qf[NIT];
#pragma unroll
  for (int t = 0; t < 4; ++t) kf[t] = gfrag<T>(kb, ld, t * 16 + li, g);
#pragma unroll
  for (int t = 0; t < NIT; ++t) qf[t] = gfrag<T>(qb, ld, (it0 + t) * 16 + li, g);
  f32x4 s[4][NIT];  // [jt][it - it0]: key j = 16jt+4g+r, query i = 16it+li
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int it = 0; it < NIT; ++it) { s[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(s[jt][it], kf[jt], qf[it]); }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float scale = 0.17677669529663689f;  // 32^-0.5
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = 16 * (it0 + it) + li;
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
  for (int it = 0; it < NIT; ++it) {
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
      for (int r = 0; r < 4; ++r) {
        long orow = win * 64 + 16 * (it0 + it) + 4 * g + r;
        if (tok_out) orow = win_to_tok(wm, orow);
        if (orow >= 0) out[orow * C + h * 32 + 16 * dt + li] = from_f<T>(o[dt][r]);
      }
  }
}

int k_attn_fwd(int dt, const void* qkv, const float* table, void* out, float* lse, int heads, int C, const WinMap& wm, hipStream_t st, int tok_out) {
  if (C != heads * 32) return -2;
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  const long npairs = nwin * heads;
  // waves per pair (NMH_ATTN_QSPLIT=1/2/4 forces; read per call: the tests switch it).  Measured (bf16, 12 heads x 27 windows per grid at
  // 10^3 tokens): 324 pairs 10.6 / 7.3 / 6.3 us with 1 / 2 / 4 waves, 1296 pairs 12.8 / 10.3 / 14.6, 2592 pairs 16.6 / 18.0 (the chip holds
  // ~5120 of these waves: 2 x 2592 starts a second round), 6000 pairs 32.9 / 27.9, 24000 pairs 121 / 119
  const char* qs_s = getenv("NMH_ATTN_QSPLIT");
  const int qs_env = qs_s ? atoi(qs_s) : 0;
  const int qs = qs_env ? qs_env : (npairs <= 768 ? 4 : (npairs <= 2560 ? 2 : (npairs <= 4000 ? 1 : 2)));
#define ATTN_FWD(T, NIT) hipLaunchKernelGGL((attn_fwd_kernel<T, NIT>), dim3((unsigned)((npairs * (4 / NIT) + 3) / 4)), dim3(256), 0, st, (const T*)qkv, table, (T*)out, lse, heads, C, wm, npairs, tok_out)
  if (dt == NMH_DT_BF16) { if (qs == 4) ATTN_FWD(bf16_t, 1); else if (qs == 2) ATTN_FWD(bf16_t, 2); else ATTN_FWD(bf16_t, 4); }
  else { if (qs == 4) ATTN_FWD(float, 1); else if (qs == 2) ATTN_FWD(float, 2); else ATTN_FWD(float, 4); }
#undef ATTN_FWD
  NMH_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of the window attention core, one wave per (window, head), NW waves per block sharing the head's bias table.
//   phase A (S^T layout, lane <-> query): P, dP -> D_i = sum_j P dP, dS, d(bias), dQ = dS K
//   phase B (S layout,   lane <-> key)  : P, dS recomputed -> dV = P^T dO, dK = dS^T Q
// Register/LDS budget is what bounds this kernel (everything is latency-chained inside a wave, so occupancy is the lever):
//   * only one query tile (phase A) / key tile (phase B) of P and dP is live at a time (32 accumulators, was 256);
//   * the K and Q tiles share one LDS buffer (K is only needed transposed in phase A, Q only in phase B) and both LDS tiles
//     are written from the fragment registers (no second global read); dO row fragments are re-read from its LDS tile;
//   * d(bias): within a lane the bin depends only on (it - jt, r), so 28 register accumulators replace the 64x64 LDS tile;
//   * products are formed transposed (mma(acc, B, A)) so that every lane owns 4 consecutive head-dim elements -> 8-byte stores.
template <typename T> __device__ __forceinline__ void frag_to_lds(char* tile, int RS, int row, int g, const Frag<T>& f);
template <> __device__ __forceinline__ void frag_to_lds<bf16_t>(char* tile, int RS, int row, int g, const Frag<bf16_t>& f) {
  *reinterpret_cast<bf16x8*>(tile + row * RS + 16 * g) = f.v;
}
template <> __device__ __forceinline__ void frag_to_lds<float>(char* tile, int RS, int row, int g, const Frag<float>& f) {
  float4* d = reinterpret_cast<float4*>(tile + row * RS + 32 * g);
  d[0] = float4{f.v[0], f.v[1], f.v[2], f.v[3]};
  d[1] = float4{f.v[4], f.v[5], f.v[6], f.v[7]};
}
template <typename T> __device__ __forceinline__ Frag<T> lds_row_frag(const char* tile, int RS, int row, int g);
template <> __device__ __forceinline__ Frag<bf16_t> lds_row_frag<bf16_t>(const char* tile, int RS, int row, int g) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(tile + row * RS + 16 * g);
  return f;
}
template <> __device__ __forceinline__ Frag<float> lds_row_frag<float>(const char* tile, int RS, int row, int g) {
  const float4* d = reinterpret_cast<const float4*>(tile + row * RS + 32 * g);
  const float4 a = d[0], b = d[1];
  Frag<float> f;
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v, float s) {
  uint2 u;
  u.x = pk_bf16(v[0] * s, v[1] * s);
  u.y = pk_bf16(v[2] * s, v[3] * s);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void store4(float* p, const f32x4& v, float s) { *reinterpret_cast<float4*>(p) = float4{v[0] * s, v[1] * s, v[2] * s, v[3] * s}; }

template <typename T, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ table, const T* __restrict__ dout, const float* __restrict__ lse,
                                                        T* __restrict__ dqkv, float* __restrict__ dtable, int heads, int C, WinMap wm, long nwin, T* __restrict__ dqkv_tok) {
  // dqkv_tok != nullptr (token mode): dout is TOKEN-ordered [T][C] (pad rows of a window read as zero) and the rows of dqkv that belong to a token go to
  // dqkv_tok[token] -- window order then lives inside this kernel only, and the GEMMs either side of it run on the real tokens.  Pad rows are still
  // written to dqkv at their window row: the qkv bias gradient sums them too (k_attn_pad_rows_colsum).
  constexpr int RS = OddRS32<32 * (int)sizeof(T)>::v;
  __shared__ __attribute__((aligned(16))) char sA[NW][64 * RS];   // K tile (phase A), then Q tile (phase B)
  __shared__ __attribute__((aligned(16))) char sO[NW][64 * RS];   // dO tile
  __shared__ float sB[344], sDB[344];
  __shared__ __attribute__((aligned(16))) float sD[NW][64];
  __shared__ __attribute__((aligned(16))) float sL[NW][64];
  __shared__ unsigned sR[NW][16];                                  // region id of the 64 tokens, one byte each
  // bf16: the probabilities of phase A stay in LDS for phase B, transposed (key-major rows of 64 queries, 136-byte rows: conflict-free 2-byte writes, 8-byte
  // reads) -- phase B then needs neither its 16 score MFMAs nor 64 bias reads, mask tests and exponentials per lane.  P is rounded to bf16 there, as it is
  // for the dV product anyway; dS = P (dP - D) takes one extra rounding before its own.  8.5 KB per wave: six waves per CU instead of eight (measured equal).
  constexpr bool PC = sizeof(T) == 2;
  constexpr int PRS = 136;
  __shared__ __attribute__((aligned(16))) char sP[PC ? NW : 1][PC ? 64 * PRS : 16];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* const tP = sP[PC ? wave : 0];
  const int h = blockIdx.y;
  const int nW = (wm.PH >> 2) * (wm.PW >> 2) * (wm.PD >> 2);
  const bool shifted = (wm.s0 + wm.s1 + wm.s2) > 0;
  const long ld = 3L * C;
  const float scale = 0.17677669529663689f;
  for (int t = threadIdx.x; t < 344; t += 64 * NW) { sB[t] = t < 343 ? table[t * heads + h] : 0.f; sDB[t] = 0.f; }
  __syncthreads();
  float dsacc[7][4];
#pragma unroll
  for (int q = 0; q < 7; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) dsacc[q][r] = 0.f;
  char* tA = sA[wave];
  char* tO = sO[wave];
  const unsigned char* sRb = reinterpret_cast<const unsigned char*>(sR[wave]);

  for (long win = (long)blockIdx.x * NW + wave; win < nwin; win += (long)gridDim.x * NW) {
    const T* qb = qkv + win * 64 * ld + h * 32;
    const T* kb = qb + C;
    const T* vb = qb + 2 * C;
    const T* dob = dout + win * 64 * C + h * 32;
    T* dqb = dqkv + win * 64 * ld + h * 32;
    // opaque lane id: everything derived from it (24 store offsets, 16 load offsets, LDS addresses, the 56 bias values a lane
    // uses) is loop-invariant and would otherwise be hoisted into ~200 registers that stay live across the whole loop
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, li = lane & 15;
    // relative-position bin of (query i, key j), i = 16 i0 + 4 i1 + i2: (i0-j0+3)*49 + (i1-j1+3)*7 + (i2-j2+3)
    const int c1 = (li >> 2) - g, c2 = li & 3;
    const int binA = (c1 + 3) * 7 + c2 + 3;  // phase A (i = 16it+li, j = 16jt+4g+r): binA + (it-jt+3)*49 - r
    const int binB = (3 - c1) * 7 + 3 - c2;  // phase B (i = 16it+4g+r, j = 16jt+li): binB + (it-jt+3)*49 + r
    Frag<T> kf[4], qf[4], vf[4];
    int tk[4] = {0, 0, 0, 0};   // token mode: token of window row 16 t + li, -1 = pad
    if (dqkv_tok) {
#pragma unroll
      for (int t = 0; t < 4; ++t) tk[t] = (int)win_to_tok(wm, win * 64 + 16 * t + li);
    }
    {
      Frag<T> df[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        kf[t] = gfrag<T>(kb, ld, t * 16 + li, g); qf[t] = gfrag<T>(qb, ld, t * 16 + li, g);
        vf[t] = gfrag<T>(vb, ld, t * 16 + li, g);
        if (dqkv_tok) {
          if (tk[t] >= 0) df[t] = gfrag<T>(dout + h * 32, (long)C, tk[t], g);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) df[t].v[j] = 0;
          }
        } else df[t] = gfrag<T>(dob, (long)C, t * 16 + li, g);
      }
      sL[wave][lane] = lse[(win * heads + h) * 64 + lane];
      if (shifted) reinterpret_cast<unsigned char*>(sR[wave])[lane] = (unsigned char)token_region(wm, (int)(win % nW), lane);
#pragma unroll
      for (int t = 0; t < 4; ++t) { frag_to_lds<T>(tA, RS, t * 16 + li, g, kf[t]); frag_to_lds<T>(tO, RS, t * 16 + li, g, df[t]); }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase A ----------------
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4 p[4], dp[4];
      {
        const Frag<T> dfi = lds_row_frag<T>(tO, RS, it * 16 + li, g);
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          p[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(p[jt], kf[jt], qf[it]);
          dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(dp[jt], vf[jt], dfi);
        }
      }
      const int i = 16 * it + li;
      const unsigned ri = shifted ? sRb[i] : 0u;
      const float L = sL[wave][i];
      float dsum = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const unsigned rw = shifted ? sR[wave][4 * jt + g] : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = p[jt][r] * scale + sB[binA + (it - jt + 3) * 49 - r];
          if (shifted && ((rw >> (8 * r)) & 255u) != ri) v += -100.0f;
          const float e = __expf(v - L);
          p[jt][r] = e;
          dsum += e * dp[jt][r];
        }
      }
      if constexpr (PC) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const unsigned w01 = pk_bf16(p[jt][0], p[jt][1]), w23 = pk_bf16(p[jt][2], p[jt][3]);
          char* const q = tP + (16 * jt + 4 * g) * PRS + i * 2;
          *reinterpret_cast<unsigned short*>(q) = (unsigned short)(w01 & 0xffffu);
          *reinterpret_cast<unsigned short*>(q + PRS) = (unsigned short)(w01 >> 16);
          *reinterpret_cast<unsigned short*>(q + 2 * PRS) = (unsigned short)(w23 & 0xffffu);
          *reinterpret_cast<unsigned short*>(q + 3 * PRS) = (unsigned short)(w23 >> 16);
        }
      }
      dsum += __shfl_xor(dsum, 16, 64);
      dsum += __shfl_xor(dsum, 32, 64);
      if (g == 0) sD[wave][i] = dsum;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = p[jt][r] * (dp[jt][r] - dsum);
          dp[jt][r] = ds;
          dsacc[it - jt + 3][r] += ds;
        }
      // dQ^T[d][i] = sum_j K[j][d] dS^T[j][i]
      f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Frag<T> a = pack_frag(dp[2 * ks], dp[2 * ks + 1], (T*)nullptr);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const Frag<T> b = lds_frag_t(tA, RS, ks * 32, dt * 16, lane, (T*)nullptr);
          mma(o[dt], b, a);
        }
      }
      {
        T* const rowp = (dqkv_tok && tk[it] >= 0) ? dqkv_tok + (long)tk[it] * ld + h * 32 : dqb + (long)i * ld;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) store4(rowp + 16 * dt + 4 * g, o[dt], scale);
      }
      __builtin_amdgcn_sched_barrier(0);   // keeps the query tiles sequential (interleaved, they need 4x the registers)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) frag_to_lds<T>(tA, RS, t * 16 + li, g, qf[t]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase B ----------------
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      f32x4 p[4], dp[4];  // [it]: query i = 16it+4g+r, key j = 16jt+li
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const Frag<T> dfi = lds_row_frag<T>(tO, RS, it * 16 + li, g);
        if constexpr (!PC) { p[it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(p[it], qf[it], kf[jt]); }
        dp[it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(dp[it], dfi, vf[jt]);
      }
      const int j = 16 * jt + li;
      const unsigned rj = shifted ? sRb[j] : 0u;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const float4 D4 = *reinterpret_cast<const float4*>(&sD[wave][16 * it + 4 * g]);
        const float Dr[4] = {D4.x, D4.y, D4.z, D4.w};
        if constexpr (PC) {   // P[i = 16 it + 4 g + r][j] of phase A
          const uint2 w = *reinterpret_cast<const uint2*>(tP + j * PRS + (16 * it + 4 * g) * 2);
          const float e4[4] = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
#pragma unroll
          for (int r = 0; r < 4; ++r) { p[it][r] = e4[r]; dp[it][r] = e4[r] * (dp[it][r] - Dr[r]); }
        } else {
          const unsigned rw = shifted ? sR[wave][4 * it + g] : 0u;
          const float4 L4 = *reinterpret_cast<const float4*>(&sL[wave][16 * it + 4 * g]);
          const float Lr[4] = {L4.x, L4.y, L4.z, L4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = p[it][r] * scale + sB[binB + (it - jt + 3) * 49 + r];
            if (shifted && ((rw >> (8 * r)) & 255u) != rj) v += -100.0f;
            const float e = __expf(v - Lr[r]);
            p[it][r] = e;
            dp[it][r] = e * (dp[it][r] - Dr[r]);
          }
        }
      }
      // dV^T[d][j] = sum_i dO[i][d] P[i][j];  dK^T[d][j] = sum_i Q[i][d] dS[i][j]
      f32x4 ov[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, ok[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Frag<T> ap = pack_frag(p[2 * ks], p[2 * ks + 1], (T*)nullptr);
        const Frag<T> ad = pack_frag(dp[2 * ks], dp[2 * ks + 1], (T*)nullptr);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const Frag<T> bo = lds_frag_t(tO, RS, ks * 32, dt * 16, lane, (T*)nullptr);
          mma(ov[dt], bo, ap);
          const Frag<T> bq = lds_frag_t(tA, RS, ks * 32, dt * 16, lane, (T*)nullptr);
          mma(ok[dt], bq, ad);
        }
      }
      {
        T* const rowp = (dqkv_tok && tk[jt] >= 0) ? dqkv_tok + (long)tk[jt] * ld + h * 32 : dqb + (long)j * ld;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          store4(rowp + 2 * C + 16 * dt + 4 * g, ov[dt], 1.0f);
          store4(rowp + C + 16 * dt + 4 * g, ok[dt], scale);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int binA0 = ((((lane0 & 15) >> 2) - (lane0 >> 4)) + 3) * 7 + (lane0 & 3) + 3;
  // lanes of one 16-lane group hit 16 distinct bins, lanes of different groups may share one: one group at a time keeps every
  // ds_add_f32 conflict-free (same-address LDS float atomics inside an instruction cost ~300 ns each on this part)
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    if ((lane0 >> 4) == ph) {
#pragma unroll
      for (int q = 0; q < 7; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(&sDB[binA0 + q * 49 - r], dsacc[q][r]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 343; t += 64 * NW) atomicAdd(dtable + t * heads + h, sDB[t]);
}

int k_attn_bwd(int dt, const void* qkv, const float* table, const void* dout, const float* lse, void* dqkv, float* dtable, int heads, int C, const WinMap& wm, hipStream_t st,
               void* dqkv_tok) {
  if (C != heads * 32) return -2;
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  // one wave per (window, head); a wave loops over several windows only when there are more waves than the chip holds at once: the grid is ONE round of
  // resident workgroups -- six waves per CU (248 VGPRs: two per SIMD by registers; 45 KB of LDS per 2-wave workgroup with the bf16 probability cache:
  // three workgroups per CU).  Rounds 3-5 measured 8 waves per CU and 4-wave workgroups for mid-size launches: within 5 % of this everywhere
  // (profiles/r5m_attn_bwd_waves_per_cu_sweep_not_kept.txt).
  const int nw = 2;
  long gx = (nwin + nw - 1) / nw;
  long cap = (256L * 6 / nw) / heads;
  if (cap < 1) cap = 1;
  if (gx > cap) gx = cap;
  dim3 grid((unsigned)gx, heads);
  if (dt == NMH_DT_BF16) {
    hipLaunchKernelGGL((attn_bwd_kernel<bf16_t, 2>), grid, dim3(128), 0, st, (const bf16_t*)qkv, table, (const bf16_t*)dout, lse, (bf16_t*)dqkv, dtable, heads, C, wm, nwin, (bf16_t*)dqkv_tok);
  } else hipLaunchKernelGGL((attn_bwd_kernel<float, 2>), dim3((unsigned)gx, heads), dim3(128), 0, st, (const float*)qkv, table, (const float*)dout, lse, (float*)dqkv, dtable, heads, C, wm, nwin, (float*)dqkv_tok);
  NMH_CHECK_LAUNCH();
  return 0;
}

// out[n] += sum over the PAD rows (window rows without a token) of x[row][n]: the part of the qkv bias gradient that the token-ordered weight-gradient
// GEMM does not see (pad tokens enter the attention as keys / values with q = k = v = bias, swin_mae3d.py:62-112, so their d(qkv) is not zero).
// A workgroup walks every gridDim.x-th window: thread = (8-column chunk, row slice); the window's pad rows are listed once per window (wave ballot).
// Grouped form: blockIdx.y = item (one Swin block's buffer and bias gradient).  A stage's blocks share the launch: as one launch per block the 18 stage-2 items
// were 18 nodes on the side branch of the captured step, and at 1 grid per GPU the host-driven graph launch left the chip idle between them (0.8 ms of an 11-ms step).
constexpr int PADSUM_MAX = 24;
struct PadSumArgs { const void* x[PADSUM_MAX]; float* out[PADSUM_MAX]; };
template <typename T>
__global__ __launch_bounds__(512) void attn_pad_rows_colsum_kernel(PadSumArgs pa, int N, WinMap wm, long nwin) {
  const T* __restrict__ x = reinterpret_cast<const T*>(pa.x[blockIdx.y]);
  float* __restrict__ out = pa.out[blockIdx.y];
  extern __shared__ float ssum[];   // [N]
  __shared__ unsigned char list[64];   // the window's pad rows, compacted
  __shared__ int npad;
  const int nch = N >> 3, S = 512 / nch > 0 ? 512 / nch : 1;   // nch <= 512 (launcher)
  const int c = threadIdx.x % nch, sl = threadIdx.x / nch;
  for (int i = threadIdx.x; i < N; i += 512) ssum[i] = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long win = blockIdx.x; win < nwin; win += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < 64) {   // wave 0: ballot of the pad rows -> compact list
      const bool p = win_to_tok(wm, win * 64 + threadIdx.x) < 0;
      const unsigned long long m = __ballot(p);
      if (p) list[__popcll(m & ((1ull << threadIdx.x) - 1ull))] = (unsigned char)threadIdx.x;
      if (threadIdx.x == 0) npad = __popcll(m);
    }
    __syncthreads();
    const int n = npad;
    if (sl < S) {
      const T* const base = x + win * 64 * (long)N + c * 8;
      int i = sl;
      for (; i + 3 * S < n; i += 4 * S) {   // four independent loads in flight
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) Vec8<T>::load(base + (long)list[i + u * S] * N, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v[u][j];
      }
      for (; i < n; i += S) {
        float v[8];
        Vec8<T>::load(base + (long)list[i] * N, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += v[j];
      }
    }
  }
  if (sl < S) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&ssum[c * 8 + j], acc[j]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 512) atomicAdd(out + i, ssum[i]);
}
int k_attn_pad_rows_colsum_grouped(int dt, const void* const* xs, float* const* outs, int n, int N, const WinMap& wm, hipStream_t st) {
  if (N % 8 || n < 0) return -2;
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  if ((long)wm.PH * wm.PW * wm.PD == (long)wm.H * wm.W * wm.D) return 0;   // no pad rows
  if (N > 4096) return -2;   // (one chunk per thread: N / 8 <= 512)
  const unsigned nb = (unsigned)(nwin < 256 ? nwin : 256);   // (one window per workgroup at 8 grids of stage 2: 216 windows)
  for (int base = 0; base < n; base += PADSUM_MAX) {
    const int cnt = n - base < PADSUM_MAX ? n - base : PADSUM_MAX;
    PadSumArgs pa{};
    for (int i = 0; i < cnt; ++i) {
      if (!xs[base + i] || !outs[base + i]) return -4;
      pa.x[i] = xs[base + i]; pa.out[i] = outs[base + i];
    }
    if (dt == NMH_DT_BF16) hipLaunchKernelGGL(attn_pad_rows_colsum_kernel<bf16_t>, dim3(nb, cnt), dim3(512), N * sizeof(float), st, pa, N, wm, nwin);
    else hipLaunchKernelGGL(attn_pad_rows_colsum_kernel<float>, dim3(nb, cnt), dim3(512), N * sizeof(float), st, pa, N, wm, nwin);
    NMH_CHECK_LAUNCH();
  }
  return 0;
}
int k_attn_pad_rows_colsum(int dt, const void* x, int N, const WinMap& wm, float* out, hipStream_t st) {
  return k_attn_pad_rows_colsum_grouped(dt, &x, &out, 1, N, wm, st);
}
