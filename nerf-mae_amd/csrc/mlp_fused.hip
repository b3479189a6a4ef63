// Fused MLP branch of a Swin block (reference: swin_mae3d.py:352-358 torchvision MLP, :368 `x = x + stochastic_depth(mlp(norm2(x)))`),
// bf16, gfx950.  SURVEY 2a "K2": LayerNorm -> Linear(C,4C) -> exact-erf GELU -> Linear(4C,C) -> row-scale -> + residual in ONE kernel,
// and its backward (recomputes LN and the hidden activations, never reads a stored pre-activation) in one kernel.
//
// Work decomposition (both kernels): a 256-thread workgroup owns 64*MT consecutive token rows, wave w the rows 16*MT*w .. +16*MT.
//   * The wave's rows never touch LDS: lane (i = lane&15, g = lane>>4) loads X[row i][32 s + 8 g ..+7] straight into the MFMA operand
//     fragment of k-step s, so LayerNorm is a reduction over the lane's own elements and the 4 lanes that share a row (two shuffles).
//   * The hidden dimension is walked in chunks of HC units.  Per chunk the rows W1[chunk][C] and W2^T[chunk][C] (both K-contiguous
//     packs that already exist: "fc1.w" and "fc2.wT") are staged through a double-buffered LDS ring (register prefetch of chunk j+1
//     under the MFMAs of chunk j, one barrier per chunk).
//   * Products are formed transposed (weights as the MFMA A operand): a lane then holds, for token `i`, four consecutive hidden
//     units 16 b + 4 g + r of n-tile b.  Two such tiles ARE the 8 k-slots of the next contraction in the slot order of
//     ds_read_b64_tr_b16 (common.hpp: slot (g,j) <-> 4g+j | 16+4g+(j-4)), so GELU(h) feeds the second GEMM from registers and the
//     contraction-major operand (W2^T rows = hidden units for the forward, W1 rows = hidden units for the backward's dX) is read
//     from the SAME row-major LDS tile with transpose reads.  Row stride 2C+32 bytes is conflict-free for both read kinds.
//   * Forward saves nothing but the block's input; backward writes gelu(h) and dh (the operands of the two weight-gradient GEMMs,
//     which stay on the grouped weight-gradient path) and x1n = LN2(x1).
// HBM traffic per token at C = 96: forward 2C*2 B (was 2C*2 + 2*4C*2 + ...), backward 3C*2 + 2*4C*2 + C*2.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>

namespace {

template <int C> struct MlpCfg { static constexpr int RS = 2 * C + 32; };   // LDS row stride (bytes) of a staged weight row

// 8 consecutive bf16 -> floats
__device__ __forceinline__ void unpack8(const uint4& u, float (&v)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v u = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ Frag<bf16_t> pack_tr(const f32x4& lo, const f32x4& hi) {   // two C-layout tiles -> operand fragment in tr slot order
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v u = {pk_bf16(lo[0], lo[1]), pk_bf16(lo[2], lo[3]), pk_bf16(hi[0], hi[1]), pk_bf16(hi[2], hi[3])};
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8, u);
  return f;
}
__device__ __forceinline__ float quad_row_sum(float v) {   // sum over the 4 lanes (g = 0..3) that share row lane&15
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// LayerNorm of the wave's row `row` (clamped) in fragment layout: returns the normalised bf16 fragments, mean and rstd
template <int C>
__device__ __forceinline__ void ln_rows(const bf16_t* __restrict__ x, long row, int g, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        Frag<bf16_t> (&af)[C / 32], float& mean, float& rstd) {
  constexpr int KS = C / 32;
  float xv[KS][8];
  float s = 0.f;
  const bf16_t* xr = x + row * C + 8 * g;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    unpack8(*reinterpret_cast<const uint4*>(xr + 32 * k), xv[k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[k][j];
  }
  mean = quad_row_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[k][j] - mean; q += d * d; }
  rstd = rsqrtf(quad_row_sum(q) * (1.0f / C) + eps);
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + 32 * k + 8 * g), g1 = *reinterpret_cast<const float4*>(gamma + 32 * k + 8 * g + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + 32 * k + 8 * g), b1 = *reinterpret_cast<const float4*>(beta + 32 * k + 8 * g + 4);
    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (xv[k][j] - mean) * rstd * gm[j] + bt[j];
    af[k].v = pack8(o);
  }
}

// the same from rows and affine parameters already in registers (the lane's C / 32 16-byte pieces, requested a tile earlier; gamma / beta of its k-groups)
template <int C>
__device__ __forceinline__ void ln_rows_raw(const uint4 (&raw)[C / 32], const float (&gm)[C / 32][8], const float (&bt)[C / 32][8], float eps,
                                            Frag<bf16_t> (&af)[C / 32], float& mean, float& rstd) {
  constexpr int KS = C / 32;
  float xv[KS][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    unpack8(raw[k], xv[k]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += xv[k][j];
  }
  mean = quad_row_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = xv[k][j] - mean; q += d * d; }
  rstd = rsqrtf(quad_row_sum(q) * (1.0f / C) + eps);
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (xv[k][j] - mean) * rstd * gm[k][j] + bt[k][j];
    af[k].v = pack8(o);
  }
}

// register-staged copy of chunk j of two [4C][C] row-major weight matrices into the LDS ring (rows padded to RS bytes)
template <int C, int HC> struct WStage {
  static constexpr int RS = MlpCfg<C>::RS, CPR = C / 8, PER = HC * CPR, TOTAL = 2 * PER, NI = (TOTAL + 255) / 256;
  uint4 r[NI];
  __device__ __forceinline__ void gload(const bf16_t* __restrict__ Wa, const bf16_t* __restrict__ Wb, int j, int tid) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = tid + 256 * i;
      if (TOTAL % 256 == 0 || p < TOTAL) {
        const int op = p >= PER, rem = p - op * PER, row = rem / CPR, c16 = rem - row * CPR;
        r[i] = *reinterpret_cast<const uint4*>((op ? Wb : Wa) + ((long)j * HC + row) * C + c16 * 8);
      }
    }
  }
  __device__ __forceinline__ void sstore(char* buf, int tid) const {   // buf: [W a chunk: HC rows][W b chunk: HC rows]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int p = tid + 256 * i;
      if (TOTAL % 256 == 0 || p < TOTAL) {
        const int op = p >= PER, rem = p - op * PER, row = rem / CPR, c16 = rem - row * CPR;
        *reinterpret_cast<uint4*>(buf + (op * HC + row) * RS + c16 * 16) = r[i];
      }
    }
  }
};

// row-major operand fragment of staged weight row `row`, k-step s (16-byte read)
__device__ __forceinline__ Frag<bf16_t> wrow_frag(const char* tile, int RS, int row, int s, int g) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(tile + row * RS + (32 * s + 8 * g) * 2);
  return f;
}

struct MlpFwdArgs {
  const bf16_t* x1; const float* gamma; const float* beta; const bf16_t* W1; const float* b1; const bf16_t* W2T; const float* b2;
  const float* rowscale; int rows_per_scale; bf16_t* x2; float* mean; float* rstd; long M; float eps;
};

// ------------------------------------------------------------------------------------------------
// forward: x2 = x1 + s_row * (gelu(LN(x1) W1^T + b1) W2^T + b2)
// ------------------------------------------------------------------------------------------------
template <int C, int MT, int HC>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpFwdArgs a) {
  constexpr int KS = C / 32, CT = C / 16, NT1 = HC / 16, KS2 = HC / 32, NCH = 4 * C / HC, RS = MlpCfg<C>::RS, BUF = 2 * HC * RS, SLD = C + 4;
  static_assert(2 * BUF >= 4 * 16 * SLD * 4, "the epilogue slabs alias the weight ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const long rbase = (long)blockIdx.x * (64 * MT) + wave * (16 * MT);

  Frag<bf16_t> af[MT][KS];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const long row = rbase + 16 * m + li;
    float mean, rstd;
    ln_rows<C>(a.x1, row < a.M ? row : a.M - 1, g, a.gamma, a.beta, a.eps, af[m], mean, rstd);
    if (a.mean && g == 0 && row < a.M) { a.mean[row] = mean; a.rstd[row] = rstd; }
  }
  f32x4 acc[MT][CT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  WStage<C, HC> ws;
  ws.gload(a.W1, a.W2T, 0, tid);
  ws.sstore(smem, tid);
  __syncthreads();
  for (int j = 0; j < NCH; ++j) {
    const char* w1 = smem + (j & 1) * BUF;
    const char* w2 = w1 + HC * RS;
    // (vector-memory operations retire in order: a load issued AFTER the weight prefetch would make its consumer wait for the whole prefetch)
    float4 bias4[NT1];
#pragma unroll
    for (int b = 0; b < NT1; ++b) bias4[b] = *reinterpret_cast<const float4*>(a.b1 + j * HC + 16 * b + 4 * g);
    if (j + 1 < NCH) ws.gload(a.W1, a.W2T, j + 1, tid);
    f32x4 h[MT][NT1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int b = 0; b < NT1; ++b) h[m][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int b = 0; b < NT1; ++b) {
        const Frag<bf16_t> wf = wrow_frag(w1, RS, 16 * b + li, s, g);
#pragma unroll
        for (int m = 0; m < MT; ++m) mma(h[m][b], wf, af[m][s]);     // h[m][b]: token li, hidden units 16 b + 4 g + r
      }
#pragma unroll
    for (int b = 0; b < NT1; ++b) {
      const float bv[4] = {bias4[b].x, bias4[b].y, bias4[b].z, bias4[b].w};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[m][b][r] = gelu_fast_f(h[m][b][r] + bv[r]);
    }
#pragma unroll
    for (int q = 0; q < KS2; ++q) {
      Frag<bf16_t> pa[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) pa[m] = pack_tr(h[m][2 * q], h[m][2 * q + 1]);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const Frag<bf16_t> wf = lds_frag_t(w2, RS, 32 * q, 16 * c, lane, (bf16_t*)nullptr);
#pragma unroll
        for (int m = 0; m < MT; ++m) mma(acc[m][c], wf, pa[m]);       // acc[m][c]: token li, output channels 16 c + 4 g + r
      }
    }
    if (j + 1 < NCH) ws.sstore(smem + ((j + 1) & 1) * BUF, tid);
    __syncthreads();
  }
  // epilogue through a wave-private fp32 slab: coalesced 16-byte residual loads and stores
  float* stg = reinterpret_cast<float*>(smem) + wave * 16 * SLD;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
#pragma unroll
    for (int c = 0; c < CT; ++c) *reinterpret_cast<float4*>(stg + li * SLD + 16 * c + 4 * g) = make_float4(acc[m][c][0], acc[m][c][1], acc[m][c][2], acc[m][c][3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int it = lane; it < 16 * (C / 8); it += 64) {
      const int rr = it / (C / 8), cc = it - rr * (C / 8);
      const long row = rbase + 16 * m + rr;
      if (row < a.M) {
        const float4 p0 = *reinterpret_cast<const float4*>(stg + rr * SLD + cc * 8), p1 = *reinterpret_cast<const float4*>(stg + rr * SLD + cc * 8 + 4);
        const float4 c0 = *reinterpret_cast<const float4*>(a.b2 + cc * 8), c1 = *reinterpret_cast<const float4*>(a.b2 + cc * 8 + 4);
        float v[8] = {p0.x + c0.x, p0.y + c0.y, p0.z + c0.z, p0.w + c0.w, p1.x + c1.x, p1.y + c1.y, p1.z + c1.z, p1.w + c1.w};
        const float sc = a.rowscale ? a.rowscale[row / a.rows_per_scale] : 1.0f;
        float xr[8];
        unpack8(*reinterpret_cast<const uint4*>(a.x1 + row * C + cc * 8), xr);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc + xr[j];
        *reinterpret_cast<bf16x8*>(a.x2 + row * C + cc * 8) = pack8(v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

struct MlpBwdArgs {
  const bf16_t* x1; const bf16_t* dx2; const float* gamma; const float* beta; const bf16_t* W1; const float* b1; const bf16_t* W2T;
  const float* rowscale; int rows_per_scale;
  bf16_t* dx1; bf16_t* x1n; bf16_t* hact; bf16_t* dh; float* dgamma; float* dbeta;
  bf16_t* dyw; const float* dyw_scale; WinMap wm; int dyw_pads;
  long M; float eps;
};

// ------------------------------------------------------------------------------------------------
// backward: given dx2 = dL/dx2,
//   x1n = LN(x1);  hp = x1n W1^T + b1;  hact = gelu(hp)                       (recomputed)
//   dh  = s_row * (dx2 W2) * gelu'(hp)                                        [M, 4C]   (written: A operand of dW1 = dh^T x1n)
//   dxn = dh W1                                                               [M, C]
//   dx1 = dx2 + LN_backward(dxn);  dgamma += sum_rows dxn * xhat;  dbeta += sum_rows dxn
//   dyw[window row of token] = dyw_scale[sample] * dx1   (optional: the attention branch's window-ordered gradient)
// ------------------------------------------------------------------------------------------------
template <int C, int MT, int HC>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpBwdArgs a) {
  constexpr int KS = C / 32, CT = C / 16, NT1 = HC / 16, KS2 = HC / 32, NCH = 4 * C / HC, RS = MlpCfg<C>::RS, BUF = 2 * HC * RS, SLD = C + 4;
  constexpr int SRS = 2 * HC + 16;                       // row stride (bytes) of the per-wave store slabs of hact / dh
  constexpr int SLAB = 2 * 16 * SRS;                     // one wave: [hact | dh][16 rows]
  constexpr int AUX = (4 * SLAB > 32 * C ? 4 * SLAB : 32 * C);   // also holds the [4 waves][2][C] dgamma / dbeta partials at the end
  static_assert(2 * BUF >= 4 * 16 * SLD * 4, "the dx1 slabs alias the weight ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* aux = smem + 2 * BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const long rbase = (long)blockIdx.x * (64 * MT) + wave * (16 * MT);

  if (a.dyw && a.dyw_pads) {   // pad rows of the window-ordered output receive no token: zeroed here
    const long wrows = (long)a.wm.B * a.wm.PH * a.wm.PW * a.wm.PD;
    for (long i = (long)blockIdx.x * 256 + tid; i < wrows * (C / 8); i += (long)gridDim.x * 256) {
      const unsigned m = (unsigned)i / (unsigned)(C / 8), c = (unsigned)i - m * (unsigned)(C / 8);
      if (win_to_tok(a.wm, (long)m) < 0) *reinterpret_cast<uint4*>(a.dyw + (long)m * C + c * 8) = make_uint4(0, 0, 0, 0);
    }
  }

  Frag<bf16_t> xf[MT][KS], df[MT][KS];
  float mean[MT], rstd[MT], sc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const long row = rbase + 16 * m + li;
    const long rc = row < a.M ? row : a.M - 1;
    ln_rows<C>(a.x1, rc, g, a.gamma, a.beta, a.eps, xf[m], mean[m], rstd[m]);
    sc[m] = a.rowscale ? a.rowscale[rc / a.rows_per_scale] : 1.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      df[m][s].v = *reinterpret_cast<const bf16x8*>(a.dx2 + rc * C + 32 * s + 8 * g);
      if (row < a.M) *reinterpret_cast<bf16x8*>(a.x1n + row * C + 32 * s + 8 * g) = xf[m][s].v;
    }
  }
  f32x4 acc[MT][CT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  WStage<C, HC> ws;
  ws.gload(a.W1, a.W2T, 0, tid);
  ws.sstore(smem, tid);
  __syncthreads();
  char* slab = aux + wave * SLAB;
  for (int j = 0; j < NCH; ++j) {
    const char* w1 = smem + (j & 1) * BUF;
    const char* w2 = w1 + HC * RS;
    float4 bias4[NT1];
#pragma unroll
    for (int b = 0; b < NT1; ++b) bias4[b] = *reinterpret_cast<const float4*>(a.b1 + j * HC + 16 * b + 4 * g);
    if (j + 1 < NCH) ws.gload(a.W1, a.W2T, j + 1, tid);
    f32x4 hp[MT][NT1], dh[MT][NT1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int b = 0; b < NT1; ++b) { hp[m][b] = f32x4{0.f, 0.f, 0.f, 0.f}; dh[m][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int b = 0; b < NT1; ++b) {
        const Frag<bf16_t> wf1 = wrow_frag(w1, RS, 16 * b + li, s, g);
        const Frag<bf16_t> wf2 = wrow_frag(w2, RS, 16 * b + li, s, g);
#pragma unroll
        for (int m = 0; m < MT; ++m) { mma(hp[m][b], wf1, xf[m][s]); mma(dh[m][b], wf2, df[m][s]); }
      }
#pragma unroll
    for (int b = 0; b < NT1; ++b) {
      const float bv[4] = {bias4[b].x, bias4[b].y, bias4[b].z, bias4[b].w};
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = hp[m][b][r] + bv[r];
          float er, e;
          erf_as_parts(x * 0.70710678118654752f, er, e);
          const float cdf = 0.5f * (1.0f + er);
          hp[m][b][r] = x * cdf;                                                           // gelu(x)
          dh[m][b][r] = dh[m][b][r] * (cdf + x * 0.39894228040143268f * e) * sc[m];        // dL/d(pre-activation)
        }
    }
    // the two [rows][HC] chunks leave through the wave's slab: 16-byte row-contiguous stores
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int b = 0; b < NT1; ++b) {
        uint2 u, v;
        u.x = pk_bf16(hp[m][b][0], hp[m][b][1]); u.y = pk_bf16(hp[m][b][2], hp[m][b][3]);
        v.x = pk_bf16(dh[m][b][0], dh[m][b][1]); v.y = pk_bf16(dh[m][b][2], dh[m][b][3]);
        *reinterpret_cast<uint2*>(slab + li * SRS + (16 * b + 4 * g) * 2) = u;
        *reinterpret_cast<uint2*>(slab + 16 * SRS + li * SRS + (16 * b + 4 * g) * 2) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = lane; it < 2 * 16 * (HC / 8); it += 64) {
        const int t = it / (16 * (HC / 8)), rem = it - t * (16 * (HC / 8)), rr = rem / (HC / 8), cc = rem - rr * (HC / 8);
        const long row = rbase + 16 * m + rr;
        if (row < a.M) {
          const uint4 val = *reinterpret_cast<const uint4*>(slab + t * 16 * SRS + rr * SRS + cc * 16);
          *reinterpret_cast<uint4*>((t ? a.dh : a.hact) + row * (4L * C) + j * HC + cc * 8) = val;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int q = 0; q < KS2; ++q) {
      Frag<bf16_t> pa[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) pa[m] = pack_tr(dh[m][2 * q], dh[m][2 * q + 1]);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const Frag<bf16_t> wf = lds_frag_t(w1, RS, 32 * q, 16 * c, lane, (bf16_t*)nullptr);   // W1[hidden][channel], contraction over hidden
#pragma unroll
        for (int m = 0; m < MT; ++m) mma(acc[m][c], wf, pa[m]);       // acc[m][c]: token li, channels 16 c + 4 g + r
      }
    }
    if (j + 1 < NCH) ws.sstore(smem + ((j + 1) & 1) * BUF, tid);
    __syncthreads();
  }

  // ---- LayerNorm backward on acc = dL/d(x1n) (token li, channels 16 c + 4 g + r) ----
  float* part = reinterpret_cast<float*>(aux) + wave * 2 * C;     // this wave's dgamma / dbeta partial sums
  float* stg = reinterpret_cast<float*>(smem) + wave * 16 * SLD;
  float pg[CT][4], pb[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { pg[c][r] = 0.f; pb[c][r] = 0.f; }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const long row = rbase + 16 * m + li;
    const bool ok = row < a.M;
    const long rc = ok ? row : a.M - 1;
    float s1 = 0.f, s2 = 0.f;
    float xh[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const uint2 xr = *reinterpret_cast<const uint2*>(a.x1 + rc * C + 16 * c + 4 * g);
      const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u), __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
      const float4 gm4 = *reinterpret_cast<const float4*>(a.gamma + 16 * c + 4 * g);
      const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = ok ? acc[m][c][r] : 0.f;
        const float h = (xv[r] - mean[m]) * rstd[m];
        xh[c][r] = h;
        pg[c][r] += d * h;
        pb[c][r] += d;
        const float gg = d * gm[r];
        acc[m][c][r] = gg;
        s1 += gg;
        s2 += gg * h;
      }
    }
    const float m1 = quad_row_sum(s1) * (1.0f / C), m2 = quad_row_sum(s2) * (1.0f / C);
#pragma unroll
    for (int c = 0; c < CT; ++c)
      *reinterpret_cast<float4*>(stg + li * SLD + 16 * c + 4 * g) = make_float4(rstd[m] * (acc[m][c][0] - m1 - xh[c][0] * m2), rstd[m] * (acc[m][c][1] - m1 - xh[c][1] * m2),
                                                                                 rstd[m] * (acc[m][c][2] - m1 - xh[c][2] * m2), rstd[m] * (acc[m][c][3] - m1 - xh[c][3] * m2));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int it = lane; it < 16 * (C / 8); it += 64) {
      const int rr = it / (C / 8), cc = it - rr * (C / 8);
      const long row2 = rbase + 16 * m + rr;
      if (row2 < a.M) {
        const float4 p0 = *reinterpret_cast<const float4*>(stg + rr * SLD + cc * 8), p1 = *reinterpret_cast<const float4*>(stg + rr * SLD + cc * 8 + 4);
        float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float rv[8];
        unpack8(*reinterpret_cast<const uint4*>(a.dx2 + row2 * C + cc * 8), rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
        *reinterpret_cast<bf16x8*>(a.dx1 + row2 * C + cc * 8) = pack8(v);
        if (a.dyw) {
          const float s = a.dyw_scale ? a.dyw_scale[(unsigned long)row2 / (unsigned long)a.rows_per_scale] : 1.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= s;
          *reinterpret_cast<bf16x8*>(a.dyw + tok_to_win(a.wm, row2) * C + cc * 8) = pack8(v);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // column sums over the wave's rows: butterfly over the 16 lanes of a row group (DPP), lane li == 0 of each g owns columns 16 c + 4 g + r
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vg = pg[c][r], vb = pb[c][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { vg += __shfl_xor(vg, o, 64); vb += __shfl_xor(vb, o, 64); }
      if (li == 0) { part[16 * c + 4 * g + r] = vg; part[C + 16 * c + 4 * g + r] = vb; }
    }
  __syncthreads();
  {
    const float* p = reinterpret_cast<const float*>(aux);
    for (int i = tid; i < 2 * C; i += 256) {
      const float v = p[i] + p[2 * C + i] + p[4 * C + i] + p[6 * C + i];
      atomicAdd((i < C ? a.dgamma : a.dbeta - C) + i, v);
    }
  }
}

// ================================================================================================
// C = 96 (stage 0 of swin_t/s/l: 40^3 tokens per grid, 58 % of the encoder's token rows): both weight matrices (2 x 72 KB) stay RESIDENT in
// LDS in MFMA-fragment order, the workgroups are persistent (one per CU) and every wave walks its own 16*MT-row tiles without a single
// barrier after the set-up.  Fragment (n-tile b, k-step s) of a [384][96] matrix is one lane-linear KB: lane (li, g) holds
// W[16 b + li][32 s + 8 g .. +7] -- a conflict-free ds_read_b128 per fragment.  The contraction-major operand of the SECOND product of
// each chain (hidden units as k) is read from the same image with ds_read_b64_tr_b16 at computed addresses: the 32 lanes of a transpose
// read cover two aligned 128-byte runs, i.e. every bank exactly once.
// ================================================================================================
constexpr int W96_BYTES = 24 * 3 * 1024;

// byte offset of lane (li, g)'s 16 bytes inside a fragment's KB: [g >> 1][li][g & 1] -- row reads stay lane-distinct (every bank once per
// b128 service group), and the eight rows x 32 bytes of a transpose read are one contiguous 256-byte run (the plain lane order g*16 + li puts
// them into two 128-byte runs 256 bytes apart = the same 32 banks twice: PMC showed 31 % of the LDS cycles as conflicts)
__device__ __forceinline__ int w96_lane_off(int li, int g) { return (g >> 1) * 512 + li * 32 + (g & 1) * 16; }

__device__ __forceinline__ void w96_stage(char* dst, const bf16_t* __restrict__ W, int wave, int nwaves, int lane) {
  const int g = lane >> 4, li = lane & 15;
  for (int f = wave; f < 72; f += nwaves) {
    const int b = f / 3, s = f - 3 * b;
    *reinterpret_cast<uint4*>(dst + f * 1024 + w96_lane_off(li, g)) = *reinterpret_cast<const uint4*>(W + (16 * b + li) * 96 + 32 * s + 8 * g);
  }
}
__device__ __forceinline__ Frag<bf16_t> w96_row(const char* W, int b, int s, int lane) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(W + (b * 3 + s) * 1024 + w96_lane_off(lane & 15, lane >> 4));
  return f;
}
// lane-dependent part of a transposed fragment address: rows m0 + 4 g + (p >> 2), columns 16 ct + 4 (p & 3) .. +3, i.e. element W[n][c] of
// fragment (n >> 4, c >> 5) at [(c >> 4) & 1][n & 15][(c >> 3) & 1][c & 7]
__device__ __forceinline__ int w96_tr_lane(int lane) {
  const int g = lane >> 4, p = lane & 15;
  return (4 * g + (p >> 2)) * 32 + ((p & 3) >> 1) * 16 + (p & 1) * 8;
}
// k-step over hidden rows [32 kq, 32 kq + 32), operand rows = channels 16 ct .. +15
__device__ __forceinline__ Frag<bf16_t> w96_tr(const char* W, int kq, int ct, int trl) {
  const char* a = W + ((2 * kq) * 3 + (ct >> 1)) * 1024 + (ct & 1) * 512 + trl;
  const bf16x4 lo = ds_read_tr16(a), hi = ds_read_tr16(a + 3 * 1024);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}

template <int MT>
__global__ __launch_bounds__(512) void mlp96_fwd_kernel(MlpFwdArgs a) {
  constexpr int C = 96, KS = 3, CT = 6, NW = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* W1s = smem;
  char* W2s = smem + W96_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  w96_stage(W1s, a.W1, wave, NW, lane);
  w96_stage(W2s, a.W2T, wave, NW, lane);
  __syncthreads();
  const int trl = w96_tr_lane(lane);
  const long ntile = (a.M + 16 * MT - 1) / (16 * MT);
  for (long t = (long)blockIdx.x * NW + wave; t < ntile; t += (long)gridDim.x * NW) {
    const long rbase = t * (16 * MT);
    Frag<bf16_t> af[MT][KS];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long row = rbase + 16 * m + li;
      float mean, rstd;
      ln_rows<C>(a.x1, row < a.M ? row : a.M - 1, g, a.gamma, a.beta, a.eps, af[m], mean, rstd);
      if (a.mean && g == 0 && row < a.M) { a.mean[row] = mean; a.rstd[row] = rstd; }
    }
    f32x4 acc[MT][CT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int hb = 0; hb < 6; ++hb) {   // 64 hidden units per pass
      f32x4 h[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) h[m][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const Frag<bf16_t> wf = w96_row(W1s, 4 * hb + b, s, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) mma(h[m][b], wf, af[m][s]);
        }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float4 bb = *reinterpret_cast<const float4*>(a.b1 + 64 * hb + 16 * b + 4 * g);
        const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[m][b][r] = gelu_fast_f(h[m][b][r] + bv[r]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        Frag<bf16_t> pa[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) pa[m] = pack_tr(h[m][2 * q], h[m][2 * q + 1]);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const Frag<bf16_t> wf = w96_tr(W2s, 2 * hb + q, c, trl);
#pragma unroll
          for (int m = 0; m < MT; ++m) mma(acc[m][c], wf, pa[m]);
        }
      }
    }
    // epilogue straight from the accumulators: lane (li, g) owns channels 16 c + 4 g .. +3 of token li (8-byte residual loads / stores;
    // the four lanes of a token cover a 32-byte run, the six channel tiles complete the token's 192-byte row in L2)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long row = rbase + 16 * m + li;
      if (row < a.M) {
        const float sc = a.rowscale ? a.rowscale[row / a.rows_per_scale] : 1.0f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float4 b2 = *reinterpret_cast<const float4*>(a.b2 + 16 * c + 4 * g);
          const uint2 xr = *reinterpret_cast<const uint2*>(a.x1 + row * C + 16 * c + 4 * g);
          const float x0 = __uint_as_float(xr.x << 16), x1v = __uint_as_float(xr.x & 0xffff0000u), x2v = __uint_as_float(xr.y << 16), x3 = __uint_as_float(xr.y & 0xffff0000u);
          uint2 o;
          o.x = pk_bf16((acc[m][c][0] + b2.x) * sc + x0, (acc[m][c][1] + b2.y) * sc + x1v);
          o.y = pk_bf16((acc[m][c][2] + b2.z) * sc + x2v, (acc[m][c][3] + b2.w) * sc + x3);
          *reinterpret_cast<uint2*>(a.x2 + row * C + 16 * c + 4 * g) = o;
        }
      }
    }
  }
}

template <int MT>
__global__ __launch_bounds__(256) void mlp96_bwd_kernel(MlpBwdArgs a) {
  constexpr int C = 96, KS = 3, CT = 6, NW = 4, SLABB = 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* W1s = smem;
  char* W2s = smem + W96_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  char* slab = smem + 2 * W96_BYTES + wave * SLABB;   // [hact: 16 rows x 128 B | dh: 16 rows x 128 B], 8-byte granules XOR-swizzled by the row
  w96_stage(W1s, a.W1, wave, NW, lane);
  w96_stage(W2s, a.W2T, wave, NW, lane);
  if (a.dyw && a.dyw_pads) {
    const long wrows = (long)a.wm.B * a.wm.PH * a.wm.PW * a.wm.PD;
    for (long i = (long)blockIdx.x * 256 + tid; i < wrows * (C / 8); i += (long)gridDim.x * 256) {
      const unsigned m = (unsigned)i / (unsigned)(C / 8), c = (unsigned)i - m * (unsigned)(C / 8);
      if (win_to_tok(a.wm, (long)m) < 0) *reinterpret_cast<uint4*>(a.dyw + (long)m * C + c * 8) = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  const int trl = w96_tr_lane(lane);
  float pg[CT][4], pb[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) { pg[c][r] = 0.f; pb[c][r] = 0.f; }
  const long ntile = (a.M + 16 * MT - 1) / (16 * MT);
  // One wave per SIMD (the two weight matrices fill the LDS): nothing hides a load's latency but the wave's own instruction stream (PMC: waves waiting
  // 46-56 %).  So nothing is loaded where it is used: the affine parameters of both layouts sit in registers for the whole kernel (the wave has 512 to
  // itself), every row a tile reads -- x1 and dx2 in the operand layout (k-group pieces) and in the accumulator layout (4-channel pieces), the row
  // scales -- is requested one tile ahead, the fc1 bias of hidden block hb + 1 during block hb.
  float gmo[KS][8], bto[KS][8], gmc[CT][4];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) { gmo[k][j] = a.gamma[32 * k + 8 * g + j]; bto[k][j] = a.beta[32 * k + 8 * g + j]; }
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) gmc[c][r] = a.gamma[16 * c + 4 * g + r];
  struct Pre { uint4 nx[MT][KS], nd[MT][KS]; uint2 xc[MT][CT], dc[MT][CT]; float sc[MT], dsc[MT]; };
  auto fetch = [&](long t, Pre& q) {
    const long rb = t * (16 * MT);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long row = rb + 16 * m + li;
      const long rc = row < a.M ? row : a.M - 1;          // (rows behind the end, a tile behind the last one: valid addresses, values unused)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        q.nx[m][s] = *reinterpret_cast<const uint4*>(a.x1 + rc * C + 32 * s + 8 * g);
        q.nd[m][s] = *reinterpret_cast<const uint4*>(a.dx2 + rc * C + 32 * s + 8 * g);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        q.xc[m][c] = *reinterpret_cast<const uint2*>(a.x1 + rc * C + 16 * c + 4 * g);
        q.dc[m][c] = *reinterpret_cast<const uint2*>(a.dx2 + rc * C + 16 * c + 4 * g);
      }
      q.sc[m] = a.rowscale ? a.rowscale[rc / a.rows_per_scale] : 1.0f;
      q.dsc[m] = (a.dyw && a.dyw_scale) ? a.dyw_scale[rc / a.rows_per_scale] : 1.0f;
    }
  };
  Pre cur;
  const long tstride = (long)gridDim.x * NW;
  fetch((long)blockIdx.x * NW + wave, cur);
  float4 bnext[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) bnext[b] = *reinterpret_cast<const float4*>(a.b1 + 16 * b + 4 * g);
  for (long t = (long)blockIdx.x * NW + wave; t < ntile; t += tstride) {
    const long rbase = t * (16 * MT);
    Pre nxt;
    fetch(t + tstride, nxt);
    Frag<bf16_t> xf[MT][KS], df[MT][KS];
    float mean[MT], rstd[MT], sc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long row = rbase + 16 * m + li;
      ln_rows_raw<C>(cur.nx[m], gmo, bto, a.eps, xf[m], mean[m], rstd[m]);
      sc[m] = cur.sc[m];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        df[m][s].v = __builtin_bit_cast(bf16x8, cur.nd[m][s]);
        if (row < a.M) *reinterpret_cast<bf16x8*>(a.x1n + row * C + 32 * s + 8 * g) = xf[m][s].v;
      }
    }
    f32x4 acc[MT][CT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[m][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int hb = 0; hb < 6; ++hb) {
      f32x4 hp[MT][4], dh[MT][4];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) { hp[m][b] = f32x4{0.f, 0.f, 0.f, 0.f}; dh[m][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const Frag<bf16_t> wf1 = w96_row(W1s, 4 * hb + b, s, lane), wf2 = w96_row(W2s, 4 * hb + b, s, lane);
#pragma unroll
          for (int m = 0; m < MT; ++m) { mma(hp[m][b], wf1, xf[m][s]); mma(dh[m][b], wf2, df[m][s]); }
        }
      float4 bcur[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bcur[b] = bnext[b];
      {
        const int hn = hb == 5 ? 0 : hb + 1;       // (block 0 of the next tile behind the last one)
#pragma unroll
        for (int b = 0; b < 4; ++b) bnext[b] = *reinterpret_cast<const float4*>(a.b1 + 64 * hn + 16 * b + 4 * g);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float bv[4] = {bcur[b].x, bcur[b].y, bcur[b].z, bcur[b].w};
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = hp[m][b][r] + bv[r];
            float er, e;
            erf_as_parts(x * 0.70710678118654752f, er, e);
            const float cdf = 0.5f * (1.0f + er);
            hp[m][b][r] = x * cdf;
            dh[m][b][r] = dh[m][b][r] * (cdf + x * 0.39894228040143268f * e) * sc[m];
          }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          uint2 u, v;
          u.x = pk_bf16(hp[m][b][0], hp[m][b][1]); u.y = pk_bf16(hp[m][b][2], hp[m][b][3]);
          v.x = pk_bf16(dh[m][b][0], dh[m][b][1]); v.y = pk_bf16(dh[m][b][2], dh[m][b][3]);
          const int off = li * 128 + (((4 * b + g) ^ li) << 3);
          *reinterpret_cast<uint2*>(slab + off) = u;
          *reinterpret_cast<uint2*>(slab + 2048 + off) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int it = lane + 64 * k, tsel = it >> 7, rem = it & 127, rr = rem >> 3, cc = rem & 7;
          const long row = rbase + 16 * m + rr;
          uint4 val = *reinterpret_cast<const uint4*>(slab + tsel * 2048 + rr * 128 + ((((2 * cc) ^ rr) >> 1) << 4));
          if (rr & 1) val = make_uint4(val.z, val.w, val.x, val.y);
          if (row < a.M) *reinterpret_cast<uint4*>((tsel ? a.dh : a.hact) + row * 384L + 64 * hb + cc * 8) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        Frag<bf16_t> pa[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) pa[m] = pack_tr(dh[m][2 * q], dh[m][2 * q + 1]);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const Frag<bf16_t> wf = w96_tr(W1s, 2 * hb + q, c, trl);
#pragma unroll
          for (int m = 0; m < MT; ++m) mma(acc[m][c], wf, pa[m]);
        }
      }
    }
    // LayerNorm backward on acc = dL/d(x1n); dx1 (and its window-ordered copy) leave through the slab as bf16 rows of 208 bytes
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const long row = rbase + 16 * m + li;
      const bool ok = row < a.M;
      const long rc = ok ? row : a.M - 1;
      float s1 = 0.f, s2 = 0.f;
      float xh[CT][4];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const uint2 xr = cur.xc[m][c];
        const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xffff0000u), __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xffff0000u)};
        const float gm[4] = {gmc[c][0], gmc[c][1], gmc[c][2], gmc[c][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = ok ? acc[m][c][r] : 0.f;
          const float h = (xv[r] - mean[m]) * rstd[m];
          xh[c][r] = h;
          pg[c][r] += d * h;
          pb[c][r] += d;
          const float gg = d * gm[r];
          acc[m][c][r] = gg;
          s1 += gg;
          s2 += gg * h;
        }
      }
      const float m1 = quad_row_sum(s1) * (1.0f / C), m2 = quad_row_sum(s2) * (1.0f / C);
      const float dsc = cur.dsc[m];
      float v[CT][4];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const uint2 rr2 = cur.dc[m][c];
        const float rv[4] = {__uint_as_float(rr2.x << 16), __uint_as_float(rr2.x & 0xffff0000u), __uint_as_float(rr2.y << 16), __uint_as_float(rr2.y & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[c][r] = rstd[m] * (acc[m][c][r] - m1 - xh[c][r] * m2) + rv[r];
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !a.dyw) break;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float f = pass ? dsc : 1.0f;
          uint2 o;
          o.x = pk_bf16(v[c][0] * f, v[c][1] * f); o.y = pk_bf16(v[c][2] * f, v[c][3] * f);
          *reinterpret_cast<uint2*>(slab + li * 208 + (16 * c + 4 * g) * 2) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int it = lane + 64 * k, rr = it / 12, cc = it - rr * 12;
          const long row2 = rbase + 16 * m + rr;
          if (row2 < a.M) {
            const uint4 val = *reinterpret_cast<const uint4*>(slab + rr * 208 + cc * 16);
            if (pass == 0) *reinterpret_cast<uint4*>(a.dx1 + row2 * C + cc * 8) = val;
            else *reinterpret_cast<uint4*>(a.dyw + tok_to_win(a.wm, row2) * C + cc * 8) = val;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    cur = nxt;
  }
  // dgamma / dbeta: butterfly over the 16 token lanes, per-wave partials through LDS (the slabs), one set of atomics per workgroup
  __syncthreads();
  float* part = reinterpret_cast<float*>(smem + 2 * W96_BYTES) + wave * 2 * C;
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vg = pg[c][r], vb = pb[c][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { vg += __shfl_xor(vg, o, 64); vb += __shfl_xor(vb, o, 64); }
      if (li == 0) { part[16 * c + 4 * g + r] = vg; part[C + 16 * c + 4 * g + r] = vb; }
    }
  __syncthreads();
  {
    const float* p = reinterpret_cast<const float*>(smem + 2 * W96_BYTES);
    for (int i = tid; i < 2 * C; i += 256) {
      const float v = p[i] + p[2 * C + i] + p[4 * C + i] + p[6 * C + i];
      atomicAdd((i < C ? a.dgamma : a.dbeta - C) + i, v);
    }
  }
}

template <int MT> int launch96_fwd(const MlpFwdArgs& a, hipStream_t st) {
  constexpr int lds = 2 * W96_BYTES;
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp96_fwd_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  long nb = (a.M + 8 * 16 * MT - 1) / (8 * 16 * MT);
  static const long cap = getenv("NMH_MLP96_WGS") ? atol(getenv("NMH_MLP96_WGS")) : 256;   // persistent workgroups (one per CU: 144 KB of LDS each)
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL((mlp96_fwd_kernel<MT>), dim3((unsigned)nb), dim3(512), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}
template <int MT> int launch96_bwd(const MlpBwdArgs& a, hipStream_t st) {
  constexpr int lds = 2 * W96_BYTES + 4 * 4096;
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp96_bwd_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  long nb = (a.M + 4 * 16 * MT - 1) / (4 * 16 * MT);
  static const long cap = getenv("NMH_MLP96_WGS") ? atol(getenv("NMH_MLP96_WGS")) : 256;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL((mlp96_bwd_kernel<MT>), dim3((unsigned)nb), dim3(256), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

template <int C, int MT, int HC> constexpr int mlp_fwd_lds() { return 2 * 2 * HC * MlpCfg<C>::RS; }
template <int C, int MT, int HC> constexpr int mlp_bwd_lds() {
  constexpr int slab = 4 * 2 * 16 * (2 * HC + 16);
  return 2 * 2 * HC * MlpCfg<C>::RS + (slab > 32 * C ? slab : 32 * C);
}

template <int C, int MT, int HC> int launch_fwd(const MlpFwdArgs& a, hipStream_t st) {
  constexpr int lds = mlp_fwd_lds<C, MT, HC>();
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<C, MT, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  const long nb = (a.M + 64 * MT - 1) / (64 * MT);
  hipLaunchKernelGGL((mlp_fwd_kernel<C, MT, HC>), dim3((unsigned)nb), dim3(256), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}
template <int C, int MT, int HC> int launch_bwd(const MlpBwdArgs& a, hipStream_t st) {
  constexpr int lds = mlp_bwd_lds<C, MT, HC>();
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<C, MT, HC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  const long nb = (a.M + 64 * MT - 1) / (64 * MT);
  hipLaunchKernelGGL((mlp_bwd_kernel<C, MT, HC>), dim3((unsigned)nb), dim3(256), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

int env_mt(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }

}  // namespace

// supported widths: C in {96, 128, 192, 256, 384} (swin_t/s stages 0-2, swin_b* stages 0-1); returns -1 otherwise (callers keep the unfused path)
int k_mlp_fused_supported(int C) { return C == 96 || C == 128 || C == 192 || C == 256 || C == 384; }

int k_mlp_fused_fwd(const void* x1, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* b2, const float* rowscale,
                    int rows_per_scale, void* x2, float* mean, float* rstd, long M, int C, float eps, hipStream_t st) {
  MlpFwdArgs a{(const bf16_t*)x1, gamma, beta, (const bf16_t*)W1, b1, (const bf16_t*)W2T, b2, rowscale, rows_per_scale > 0 ? rows_per_scale : 1, (bf16_t*)x2, mean, rstd, M, eps};
  const int mt = env_mt("NMH_MLP_FWD_MT");   // tuning override: rows per wave / 16
  static const int p96 = getenv("NMH_MLP96") ? atoi(getenv("NMH_MLP96")) : 1;   // 0: the chunk-ring kernel for C = 96 too
  const int p96mt = env_mt("NMH_MLP96_FWD_MT");
  switch (C) {
    case 96:
      if (p96 && mt == 0) return p96mt == 1 ? launch96_fwd<1>(a, st) : launch96_fwd<2>(a, st);
      if (mt == 1) return launch_fwd<96, 1, 64>(a, st);
      if (mt == 2) return launch_fwd<96, 2, 64>(a, st);
      return launch_fwd<96, 4, 64>(a, st);
    case 128:
      if (mt == 1) return launch_fwd<128, 1, 64>(a, st);
      return launch_fwd<128, 2, 64>(a, st);
    case 192:
      if (mt == 1 || (mt == 0 && M < 32768)) return launch_fwd<192, 1, 32>(a, st);
      return launch_fwd<192, 2, 32>(a, st);
    case 256:
      if (mt == 2) return launch_fwd<256, 2, 32>(a, st);
      return launch_fwd<256, 1, 32>(a, st);
    case 384:
      if (mt == 2) return launch_fwd<384, 2, 32>(a, st);
      return launch_fwd<384, 1, 32>(a, st);
  }
  return -1;
}

int k_mlp_fused_bwd(const void* x1, const void* dx2, const float* gamma, const float* beta, const void* W1, const float* b1, const void* W2T, const float* rowscale,
                    int rows_per_scale, void* dx1, void* x1n, void* hact, void* dh, float* dgamma, float* dbeta, void* dyw, const float* dyw_scale, const WinMap* wm,
                    long M, int C, float eps, hipStream_t st) {
  MlpBwdArgs a{(const bf16_t*)x1, (const bf16_t*)dx2, gamma, beta, (const bf16_t*)W1, b1, (const bf16_t*)W2T, rowscale, rows_per_scale > 0 ? rows_per_scale : 1,
               (bf16_t*)dx1, (bf16_t*)x1n, (bf16_t*)hact, (bf16_t*)dh, dgamma, dbeta, (bf16_t*)dyw, dyw_scale, WinMap{}, 0, M, eps};
  if (dyw) {
    if (!wm) return -4;
    a.wm = *wm;
    if ((long)wm->B * wm->PH * wm->PW * wm->PD * (C / 8) >= (1L << 32)) return -2;
    a.dyw_pads = (long)wm->PH * wm->PW * wm->PD != (long)wm->H * wm->W * wm->D;
  }
  const int mt = env_mt("NMH_MLP_BWD_MT");
  static const int p96 = getenv("NMH_MLP96") ? atoi(getenv("NMH_MLP96")) : 1;
  const int p96mt = env_mt("NMH_MLP96_BWD_MT");
  switch (C) {
    case 96:
      if (p96 && mt == 0) return p96mt == 2 ? launch96_bwd<2>(a, st) : launch96_bwd<1>(a, st);
      if (mt == 1) return launch_bwd<96, 1, 64>(a, st);
      return launch_bwd<96, 2, 64>(a, st);
    case 128:
      if (mt == 2) return launch_bwd<128, 2, 64>(a, st);
      return launch_bwd<128, 1, 64>(a, st);
    case 192:
      if (mt == 2) return launch_bwd<192, 2, 32>(a, st);
      return launch_bwd<192, 1, 32>(a, st);
    case 256: return launch_bwd<256, 1, 32>(a, st);
    case 384: return launch_bwd<384, 1, 32>(a, st);
  }
  return -1;
}
