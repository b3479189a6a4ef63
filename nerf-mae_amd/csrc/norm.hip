// LayerNorm (with the window-partition / patch-merge gathers folded into its addressing), the
// window scatter/gather elementwise kernels, and InstanceNorm+LeakyReLU(+residual) forward/backward
// over channels-last volumes.  All HBM-bound: 16-byte vector accesses, fp32 statistics.
// Reference semantics: swin_mae3d.py:62-101,176-196 (pad/roll/partition), :341-369 (LN), :390-414 (merge),
// unetr_block.py:57-71 (IN eps=1e-5, no affine; LeakyReLU 0.01).
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// window maps.  Window-ordered row m = ((b*nwz+wz)*nwy+wy)*nwx+wx)*64 + tz*16+ty*4+tx.
// rolled[p] = padded[(p + shift) mod P]  (torch.roll(x, -shift)), padded rows beyond the real dims are zero.
// ------------------------------------------------------------------------------------------------
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// patch-merge source of chunk (8 elements starting at column col) of output row `row`; nullptr = zero pad
template <typename T> __device__ __forceinline__ const T* merge_src(const T* x, const WinMap& w, long row, int col, int Cin, long* tok_out) {
  const int H2 = (w.H + 1) >> 1, W2 = (w.W + 1) >> 1, D2 = (w.D + 1) >> 1;
  unsigned r = (unsigned)row;   // (32-bit div/mod: rows < 2^31)
  unsigned q = r / (unsigned)D2;
  const int x2 = (int)(r - q * (unsigned)D2);
  r = q; q = r / (unsigned)W2;
  const int y2 = (int)(r - q * (unsigned)W2);
  r = q; q = r / (unsigned)H2;
  const int z2 = (int)(r - q * (unsigned)H2);
  const long b = (long)q;
  int seg = col / Cin, off = col - seg * Cin;
  int z = 2 * z2 + (seg & 1), y = 2 * y2 + ((seg >> 1) & 1), xx = 2 * x2 + (seg >> 2);
  if (z >= w.H || y >= w.W || xx >= w.D) { *tok_out = -1; return nullptr; }
  long tok = ((b * w.H + z) * w.W + y) * w.D + xx;
  *tok_out = tok;
  return x + tok * Cin + off;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR, int NCH, int MODE>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnArgs a) {
  const int sub = threadIdx.x % LPR;
  const long gstride = (long)gridDim.x * (256 / LPR);
  const int C = a.C, nch = C >> 3, Cin = C >> 3;  // Cin only meaningful for MODE 2 (C = 8*Cin)
  const T* x = (const T*)a.x;
  T* out = (T*)a.out;
  const float invC = 1.0f / (float)C;
  for (long row = (long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR; row < a.rows; row += gstride) {
    long tok = row;
    bool valid = true;
    if (MODE == 1) { tok = win_to_tok(a.wm, row); valid = tok >= 0; }
    bool masked = false;
    long tl = 0, srow = tok;
    if (MODE == 0 && a.mask) { tl = (long)((unsigned)row % (unsigned)a.tokens_per_sample); masked = a.mask[tl] != 0; }
    if (MODE == 0 && a.rowmap) {   // compact rows of the kept tokens
      const int rm = masked ? -1 : a.rowmap[tl];
      srow = rm < 0 ? -1 : (long)((unsigned)row / (unsigned)a.tokens_per_sample) * a.cap_rows + rm;
    }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = sub + i * LPR;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
      if (c < nch && valid && srow >= 0) {
        const T* p;
        if (MODE == 2) { long tk; p = merge_src<T>(x, a.wm, row, c * 8, Cin, &tk); }
        else p = x + srow * C + c * 8;
        if (p) Vec8<T>::load(p, v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[i][j];
      }
    }
    const float mean = group_sum<LPR>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (sub + i * LPR < nch)
#pragma unroll
        for (int j = 0; j < 8; ++j) { float d = v[i][j] - mean; q += d * d; }
    const float rstd = rsqrtf(group_sum<LPR>(q) * invC + a.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = sub + i * LPR;
      if (c < nch) {
        float o[8];
        if (!valid) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
        } else if (masked) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = a.mask_token[c * 8 + j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * a.gamma[c * 8 + j] + a.beta[c * 8 + j];
          if (MODE == 0 && a.pos) {
            const long tp = a.mask ? tl : (long)((unsigned)row % (unsigned)a.tokens_per_sample);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += a.pos[tp * C + c * 8 + j];
          }
        }
        Vec8<T>::store(out + row * C + c * 8, o);
        if (MODE == 1 && a.out_tok && valid) Vec8<T>::store((T*)a.out_tok + tok * C + c * 8, o);
      }
    }
    if (sub == 0 && valid) { a.mean[tok] = mean; a.rstd[tok] = rstd; }
  }
}

template <typename T, int MODE> static int ln_fwd_dispatch(const LnArgs& a, hipStream_t st) {
  const int nch = a.C / 8;
  if (a.C % 8) return -2;
  long rows = a.rows;
#define LN_LAUNCH(LPR, NCH)                                                                   \
  {                                                                                           \
    long nb = (rows + (256 / LPR) - 1) / (256 / LPR);                                         \
    if (nb > 8192) nb = 8192;                                                                 \
    hipLaunchKernelGGL((ln_fwd_kernel<T, LPR, NCH, MODE>), dim3((unsigned)nb), dim3(256), 0, st, a); \
  }
  if (nch <= 16) LN_LAUNCH(16, 1)
  else if (nch <= 32) LN_LAUNCH(16, 2)
  else if (nch <= 48) LN_LAUNCH(16, 3)
  else if (nch <= 128) LN_LAUNCH(64, 2)
  else if (nch <= 256) LN_LAUNCH(64, 4)
  else if (nch <= 512) LN_LAUNCH(64, 8)
  else return -3;
#undef LN_LAUNCH
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_ln_fwd(const LnArgs& a, hipStream_t st) {
  if (a.dt == NMH_DT_BF16) {
    if (a.src_mode == 0) return ln_fwd_dispatch<bf16_t, 0>(a, st);
    if (a.src_mode == 1) return ln_fwd_dispatch<bf16_t, 1>(a, st);
    return ln_fwd_dispatch<bf16_t, 2>(a, st);
  }
  if (a.src_mode == 0) return ln_fwd_dispatch<float, 0>(a, st);
  if (a.src_mode == 1) return ln_fwd_dispatch<float, 1>(a, st);
  return ln_fwd_dispatch<float, 2>(a, st);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: dx = dres + rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;
// dgamma += dy*xhat, dbeta += dy (register partials -> LDS -> fp32 atomics).
// ------------------------------------------------------------------------------------------------
template <typename T, int LPR, int NCH, int MODE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) {
  extern __shared__ float sacc[];  // [3][C]: dgamma, dbeta, dmask_token
  const int sub = threadIdx.x % LPR;
  const long gstride = (long)gridDim.x * (256 / LPR);
  const int C = a.C, nch = C >> 3, Cin = C >> 3;
  const T* x = (const T*)a.x;
  const T* dy = (const T*)a.dy;
  T* dx = (T*)a.dx;
  const float invC = 1.0f / (float)C;
  // small rows keep dgamma/dbeta partials in registers; wide rows (NCH >= 4) flush per row to LDS
  constexpr bool FLUSH = NCH >= 4;
  for (int i = threadIdx.x; i < (FLUSH ? 3 : 12) * C; i += 256) sacc[i] = 0.f;   // [4 waves][3][C] (one slab when flushing per row)
  __syncthreads();
  constexpr int PN = FLUSH ? 1 : NCH, PMN = (MODE == 0 && !FLUSH) ? NCH : 1;
  float pg[PN][8], pb[PN][8], pm[PMN][8];
#pragma unroll
  for (int i = 0; i < PN; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; }
#pragma unroll
  for (int i = 0; i < PMN; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pm[i][j] = 0.f;

  // gamma of the lane's columns: loop-invariant, but the compiler cannot hoist the loads over the stores to dx.  Only for the narrow rows
  // (C <= 256: stages 0 / 1, tens of row groups per workgroup; 512000 x 96: 126 -> 118 us) -- a workgroup of the C = 384 launches handles
  // one row group, and the preloads would only lengthen its chain (1000 x 384: 10.8 -> 12.5 us)
  constexpr bool HOIST = NCH <= 2;
  float gam[HOIST ? NCH : 1][8];
  if constexpr (HOIST) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = sub + i * LPR;
#pragma unroll
      for (int j = 0; j < 8; ++j) gam[i][j] = c < nch ? a.gamma[c * 8 + j] : 0.f;
    }
  }
  if (MODE == 0 && a.dyw && a.dyw_pads) {   // pad rows of the window-ordered output (they receive no token): zeroed here, not by a launch of their own
    const long wrows = (long)a.wm.B * a.wm.PH * a.wm.PW * a.wm.PD;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < wrows * nch; i += (long)gridDim.x * 256) {
      const unsigned m = (unsigned)i / (unsigned)nch, c = (unsigned)i - m * (unsigned)nch;
      if (win_to_tok(a.wm, (long)m) < 0) { const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; Vec8<T>::store((T*)a.dyw + (long)m * C + c * 8, z8); }
    }
  }
  if (MODE == 0 && a.rowmap) {   // compact dx: the rows behind the kept count are operands of the weight gradient -- zero
    const long tps = a.tokens_per_sample, nsamp = a.rows / tps, K = a.rowmap[tps], tail = a.cap_rows - K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nsamp * tail * nch; i += (long)gridDim.x * 256) {
      const long r = i / nch, c = i - r * nch, b = r / tail, rr = K + (r - b * tail);
      const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      Vec8<T>::store(dx + (b * a.cap_rows + rr) * C + c * 8, z8);
    }
  }
  // U consecutive rows per lane group and iteration: all of their loads are issued before the first row is reduced
  constexpr int U = (!FLUSH && MODE != 2 && NCH == 2) ? 2 : 1;   // measured: 64000 x 192 57 -> 52 us with 2; 512000 x 96 (NCH = 1) 118 -> 154 us with 4 (registers)
  for (long rowb = ((long)blockIdx.x * (256 / LPR) + threadIdx.x / LPR) * U; rowb < a.rows; rowb += gstride * U) {   // U consecutive rows
    float dv[U][NCH][8], xr[U][NCH][8], rv[U][NCH][8];
    long toks[U][NCH];
    float mean[U], rstd[U];
    bool ok[U], maskedr[U];
    long xrow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = rowb + u;
      ok[u] = row < a.rows;
      maskedr[u] = false;
      xrow[u] = row;
      mean[u] = 0.f; rstd[u] = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        toks[u][i] = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dv[u][i][j] = 0.f; xr[u][i][j] = 0.f; rv[u][i][j] = 0.f; }
      }
      if (!ok[u]) continue;
      long dyrow = row;
      if (MODE == 1) dyrow = tok_to_win(a.wm, row);
      if (MODE == 0 && a.mask) maskedr[u] = a.mask[(unsigned)row % (unsigned)a.tokens_per_sample] != 0;
      xrow[u] = row;
      if (MODE == 0 && a.rowmap) {   // compact rows of the kept tokens (x and dx)
        const int rm = maskedr[u] ? -1 : a.rowmap[(unsigned)row % (unsigned)a.tokens_per_sample];
        if (rm < 0) maskedr[u] = true;   // (a kept token without a row: the host never lets the kept count exceed the capacity)
        xrow[u] = rm < 0 ? -1 : (long)((unsigned)row / (unsigned)a.tokens_per_sample) * a.cap_rows + rm;
      }
      mean[u] = a.mean[row]; rstd[u] = a.rstd[row];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = sub + i * LPR;
        if (c < nch) {
          Vec8<T>::load(dy + dyrow * C + c * 8, dv[u][i]);
          if (!maskedr[u]) {
            const T* p;
            if (MODE == 2) p = merge_src<T>(x, a.wm, row, c * 8, Cin, &toks[u][i]);
            else p = x + xrow[u] * C + c * 8;
            if (p) Vec8<T>::load(p, xr[u][i]);
          }
          if (MODE != 2 && a.dres) Vec8<T>::load((const T*)a.dres + row * C + c * 8, rv[u][i]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const long row = rowb + u;
      const bool masked = maskedr[u];
      float xv[NCH][8], gv[NCH][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = sub + i * LPR;
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[i][j] = 0.f; gv[i][j] = 0.f; }
        if (c < nch) {
          if (masked) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (FLUSH) atomicAdd(&sacc[2 * C + c * 8 + j], dv[u][i][j]);
              else pm[MODE == 0 ? i : 0][j] += dv[u][i][j];
            }
            continue;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = dv[u][i][j];
            const float xh = (xr[u][i][j] - mean[u]) * rstd[u];
            xv[i][j] = xh;
            if (FLUSH) { atomicAdd(&sacc[c * 8 + j], d * xh); atomicAdd(&sacc[C + c * 8 + j], d); }
            else { pg[i][j] += d * xh; pb[i][j] += d; }
            const float g = d * (HOIST ? gam[HOIST ? i : 0][j] : a.gamma[c * 8 + j]);
            gv[i][j] = g;
            s1 += g;
            s2 += g * xh;
          }
        }
      }
      const float m1 = group_sum<LPR>(s1) * invC, m2 = group_sum<LPR>(s2) * invC;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = sub + i * LPR;
        if (c < nch) {
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = masked ? 0.f : rstd[u] * (gv[i][j] - m1 - xv[i][j] * m2);
          if (MODE == 2) {
            if (toks[u][i] >= 0) {
              const int seg = (c * 8) / Cin, off = c * 8 - seg * Cin;
              Vec8<T>::store(dx + toks[u][i] * Cin + off, o);
            }
          } else {
            if (a.dres) {
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] += rv[u][i][j];
            }
            if (xrow[u] >= 0) Vec8<T>::store(dx + xrow[u] * C + c * 8, o);
            if (MODE == 0 && a.dyw) {   // adjoint of the window scatter fused here: dyw[win(row)] = s_b * dx[row] (pad rows pre-zeroed)
              const float sc = a.dyw_scale ? a.dyw_scale[(unsigned)row / (unsigned)a.tokens_per_sample] : 1.0f;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] *= sc;
              Vec8<T>::store((T*)a.dyw + tok_to_win(a.wm, row) * C + c * 8, o);
            }
          }
        }
      }
    }
  }
  if (!FLUSH) {
    // the 64/LPR rows of a wave hold partials for the SAME columns: sum them with shuffles first, then one row-lane group issues
    // conflict-free LDS adds (64/LPR-way same-address ds_add_f32 measured ~300 ns per instruction: 15 of this kernel's 27 us)
    const bool lead = (threadIdx.x & 63) < LPR;
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      const int c = sub + i * LPR;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float vg = pg[i][j], vb = pb[i][j], vm = (MODE == 0 && a.mask) ? pm[MODE == 0 ? i : 0][j] : 0.f;
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
          vg += __shfl_xor(vg, o, 64);
          vb += __shfl_xor(vb, o, 64);
          if (MODE == 0 && a.mask) vm += __shfl_xor(vm, o, 64);
        }
        if (lead && c < nch) {   // per-wave slab: plain stores, summed over the 4 waves below
          float* sw = sacc + (threadIdx.x >> 6) * 3 * C;
          sw[c * 8 + j] = vg;
          sw[C + c * 8 + j] = vb;
          if (MODE == 0 && a.mask) sw[2 * C + c * 8 + j] = vm;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float g = sacc[i], b = sacc[C + i], m = sacc[2 * C + i];
    if (!FLUSH) {
#pragma unroll
      for (int w = 1; w < 4; ++w) { g += sacc[w * 3 * C + i]; b += sacc[w * 3 * C + C + i]; m += sacc[w * 3 * C + 2 * C + i]; }
    }
    if (a.part) {   // (two coalesced stores per workgroup instead of 2C atomics that 256 workgroups aim at the same 2C addresses: 3 of this kernel's 24 us alone,
      //               0.26 ms of the step at 8 grids)
      a.part[(long)blockIdx.x * 2 * C + i] = g;
      a.part[(long)blockIdx.x * 2 * C + C + i] = b;
    } else {
      atomicAdd(a.dgamma + i, g);
      atomicAdd(a.dbeta + i, b);
    }
    if (MODE == 0 && a.mask) atomicAdd(a.dmask_token + i, m);
  }
}
// dgamma[c] += sum_b part[b][c], dbeta[c] += sum_b part[b][C + c] for every queued LayerNorm of a stage in ONE launch (the descriptors travel in the
// kernel arguments): 64 columns x 4 row slices per workgroup, grid (column blocks of the widest item, items)
struct LnReduceArgs { LnReduceItem it[LN_REDUCE_MAX]; };
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(LnReduceArgs ra) {
  __shared__ float s[4][64];
  const LnReduceItem& d = ra.it[blockIdx.y];
  const int C = d.C;
  const long nb = d.nb;
  const float* __restrict__ part = d.part;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= 2 * C) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < 2 * C) {
    long b = sl;
    for (; b + 12 < nb; b += 16) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += part[(b + 4 * u) * 2 * C + col];
    }
    for (; b < nb; b += 4) acc[0] += part[b * 2 * C + col];
  }
  s[sl][threadIdx.x & 63] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (sl == 0 && col < 2 * C) {
    const float t = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
    if (col < C) d.dgamma[col] += t; else d.dbeta[col - C] += t;
  }
}
int k_ln_param_reduce(const LnReduceItem* items, int n, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += LN_REDUCE_MAX) {
    LnReduceArgs ra;
    const int m = n - i0 < LN_REDUCE_MAX ? n - i0 : LN_REDUCE_MAX;
    int cmax = 0;
    for (int i = 0; i < m; ++i) {
      ra.it[i] = items[i0 + i];
      if (!ra.it[i].part || !ra.it[i].dgamma || !ra.it[i].dbeta || ra.it[i].C <= 0 || ra.it[i].nb < 0) return -2;
      if (ra.it[i].C > cmax) cmax = ra.it[i].C;
    }
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((unsigned)((2 * cmax + 63) / 64), (unsigned)m), dim3(256), 0, st, ra);
    NMH_CHECK_LAUNCH();
  }
  return 0;
}
static long lnb_max_blocks() { static const long v = getenv("NMH_LN_BWD_BLOCKS") ? atol(getenv("NMH_LN_BWD_BLOCKS")) : 1024; return v; }
// workgroups of the backward launch for `rows` rows of width C (also the rows of LnBwdArgs::part)
long k_ln_bwd_blocks(long rows, int C) {
  const int nch = C / 8, lpr = nch <= 48 ? 16 : 64;
  long nb = (rows + (256 / lpr) - 1) / (256 / lpr);
  /* every workgroup ends with 2C same-address fp32 atomics: measured optimum ~256 workgroups, ~512 from 16k row groups on */
  long cap = nb / 32 < 256 ? 256 : nb / 32;
  if (cap > lnb_max_blocks()) cap = lnb_max_blocks();
  return nb > cap ? cap : nb;
}
template <typename T, int MODE> static int ln_bwd_dispatch(const LnBwdArgs& a, hipStream_t st) {
  const int nch = a.C / 8;
  if (a.C % 8) return -2;
  const long nb = k_ln_bwd_blocks(a.rows, a.C);
#define LNB_LAUNCH(LPR, NCH)                                                                        \
  {                                                                                                 \
    const size_t lds = ((NCH) >= 4 ? 3 : 12) * (size_t)a.C * sizeof(float);                         \
    hipLaunchKernelGGL((ln_bwd_kernel<T, LPR, NCH, MODE>), dim3((unsigned)nb), dim3(256), lds, st, a); \
  }
  if (nch <= 16) LNB_LAUNCH(16, 1)
  else if (nch <= 32) LNB_LAUNCH(16, 2)
  else if (nch <= 48) LNB_LAUNCH(16, 3)
  else if (nch <= 128) LNB_LAUNCH(64, 2)
  else if (nch <= 256) LNB_LAUNCH(64, 4)
  else if (nch <= 512) LNB_LAUNCH(64, 8)
  else return -3;
#undef LNB_LAUNCH
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_ln_bwd(const LnBwdArgs& a0, hipStream_t st) {
  LnBwdArgs a = a0;
  a.dyw_pads = 0;
  if (a.dyw) {
    if (a.src_mode != 0) return -2;
    const WinMap& w = a.wm;
    if ((long)w.B * w.PH * w.PW * w.PD * (a.C / 8) >= (1L << 32)) return -2;
    a.dyw_pads = (long)w.PH * w.PW * w.PD != (long)w.H * w.W * w.D;   // pad rows of the window-ordered tensor receive no token
  }
  if (a.dt == NMH_DT_BF16) {
    if (a.src_mode == 0) return ln_bwd_dispatch<bf16_t, 0>(a, st);
    if (a.src_mode == 1) return ln_bwd_dispatch<bf16_t, 1>(a, st);
    return ln_bwd_dispatch<bf16_t, 2>(a, st);
  }
  if (a.src_mode == 0) return ln_bwd_dispatch<float, 0>(a, st);
  if (a.src_mode == 1) return ln_bwd_dispatch<float, 1>(a, st);
  return ln_bwd_dispatch<float, 2>(a, st);
}

// ------------------------------------------------------------------------------------------------
// out[tok] = x[tok] + s_b * yw[win(tok)]       (window reverse + un-roll + un-pad + residual + stochastic depth)
// dyw[m]   = s_b * dx[tok(m)] or 0 for pad rows (the adjoint gather)
// ------------------------------------------------------------------------------------------------
template <typename T> __global__ void win_scatter_kernel(const T* yw, const T* x, T* out, const float* rs, int C, WinMap wm) {
  const int nch = C >> 3;
  const long total = (long)wm.B * wm.H * wm.W * wm.D * nch;
  const long tps = (long)wm.H * wm.W * wm.D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long tok = i / nch;
    const int c = (int)(i - tok * nch);
    const long m = tok_to_win(wm, tok);
    float a[8], b[8];
    Vec8<T>::load(yw + m * C + c * 8, a);
    Vec8<T>::load(x + tok * C + c * 8, b);
    const float s = rs ? rs[tok / tps] : 1.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] += s * a[j];
    Vec8<T>::store(out + tok * C + c * 8, b);
  }
}
template <typename T> __global__ void win_gather_kernel(const T* dx, T* dyw, const float* rs, int C, WinMap wm) {
  const int nch = C >> 3;
  const long rows = (long)wm.B * (wm.PH >> 2) * (wm.PW >> 2) * (wm.PD >> 2) * 64;
  const long total = rows * nch;
  const long tps = (long)wm.H * wm.W * wm.D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / nch;
    const int c = (int)(i - m * nch);
    const long tok = win_to_tok(wm, m);
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (tok >= 0) {
      Vec8<T>::load(dx + tok * C + c * 8, a);
      if (rs) {
        const float s = rs[tok / tps];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= s;
      }
    }
    Vec8<T>::store(dyw + m * C + c * 8, a);
  }
}
static inline unsigned ew_blocks(long total) { long nb = (total + 255) / 256; return (unsigned)(nb > 16384 ? 16384 : (nb < 1 ? 1 : nb)); }

int k_window_scatter_residual(int dt, const void* yw, const void* x, void* out, const float* rs, int C, const WinMap& wm, hipStream_t st) {
  long total = (long)wm.B * wm.H * wm.W * wm.D * (C / 8);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(win_scatter_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, st, (const bf16_t*)yw, (const bf16_t*)x, (bf16_t*)out, rs, C, wm);
  else hipLaunchKernelGGL(win_scatter_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, st, (const float*)yw, (const float*)x, (float*)out, rs, C, wm);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_window_gather_scale(int dt, const void* dx, void* dyw, const float* rs, int C, const WinMap& wm, hipStream_t st) {
  long total = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4) * 64 * (C / 8);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(win_gather_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, st, (const bf16_t*)dx, (bf16_t*)dyw, rs, C, wm);
  else hipLaunchKernelGGL(win_gather_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, st, (const float*)dx, (float*)dyw, rs, C, wm);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm over channels-last [B][V][C].  Reductions: each thread owns one 8-channel chunk and
// strides over voxels; block partials -> LDS atomics -> fp64 global atomics (scratch [B][C][2]).
// ------------------------------------------------------------------------------------------------
#define IN_VOX_PER_BLOCK 4096  /* minimum; the launchers grow it so that <= ~256 blocks per sample contend on the 2*C fp64 atomics */

template <typename T, int BWD>
__global__ __launch_bounds__(256) void in_reduce_kernel(const T* x, const T* dout, const T* outp, const float* stats, const T* r, const float* stats_r, int rmode,
                                                        double* acc, double* acc_r, long V, int C, float slope, long vpb) {
  extern __shared__ float sred[];  // [4][C]
  const int CL = C >> 3, NV = 256 / CL;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL;
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * C; i += 256) sred[i] = 0.f;
  __syncthreads();
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  float s1[8], s2[8], t1[8], t2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = s2[j] = t1[j] = t2[j] = 0.f; }
  float mu[8], rs_[8], mur[8], rsr[8], cref[8];
  if (!BWD && vl < NV) Vec8<T>::load(x + ((long)b * V) * C + cl * 8, cref);  // per-channel shift kills the E[x^2]-mean^2 cancellation
  if (BWD && vl < NV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = stats[((long)b * C + cl * 8 + j) * 2]; rs_[j] = stats[((long)b * C + cl * 8 + j) * 2 + 1];
      if (rmode == 2) { mur[j] = stats_r[((long)b * C + cl * 8 + j) * 2]; rsr[j] = stats_r[((long)b * C + cl * 8 + j) * 2 + 1]; }
    }
  }
  if (vl < NV) {
    // U independent voxels per iteration: keeps several 16-B loads per tensor in flight per thread (HBM latency-bound otherwise)
    constexpr int U = 4;
    for (long vb = v0 + vl; vb < v1; vb += (long)NV * U) {
      float xv[U][8], dv[U][8], ov[U][8], rv[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vb + (long)u * NV;
        if (v < v1) {
          const long o = ((long)b * V + v) * C + cl * 8;
          Vec8<T>::load(x + o, xv[u]);
          if (BWD) {
            Vec8<T>::load(dout + o, dv[u]);
            if (outp) Vec8<T>::load(outp + o, ov[u]);
            if (rmode == 2) Vec8<T>::load(r + o, rv[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vb + (long)u * NV;
        if (v < v1) {
          if (!BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = xv[u][j] - cref[j]; s1[j] += d; s2[j] += d * d; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              // without a residual sign(out) == sign(x - mean): `out` need not be read at all
              const float g = dv[u][j] * ((outp ? ov[u][j] : xv[u][j] - mu[j]) > 0.f ? 1.0f : slope);
              s1[j] += g;
              s2[j] += g * (xv[u][j] - mu[j]) * rs_[j];
              if (rmode == 2) { t1[j] += g; t2[j] += g * (rv[u][j] - mur[j]) * rsr[j]; }
            }
          }
        }
      }
    }
  }
  // block reduction without LDS atomics (the NV voxel lanes of a channel all hit the same address: same-address ds_add_f32 costs
  // ~300 ns per instruction, which dominated the short blocks of the coarse decoder levels): every thread parks its partials in a
  // private column of sp[value][thread]; thread (k, c) then sums the NV entries of its channel
  float* sp = sred + 4 * C;   // [32][256]
  const int nk = (BWD && rmode == 2) ? 4 : 2;
  if (vl < NV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sp[(0 * 8 + j) * 256 + threadIdx.x] = s1[j];
      sp[(1 * 8 + j) * 256 + threadIdx.x] = s2[j];
      if (BWD && rmode == 2) { sp[(2 * 8 + j) * 256 + threadIdx.x] = t1[j]; sp[(3 * 8 + j) * 256 + threadIdx.x] = t2[j]; }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nk * C; i += 256) {
    const int k = i / C, c = i - k * C, ccl = c >> 3, j = c & 7;
    float t = 0.f;
    for (int q = 0; q < NV; ++q) t += sp[(k * 8 + j) * 256 + ccl + CL * q];
    double* dst = (k < 2 ? acc : acc_r) + ((long)b * C + c) * 2 + (k & 1);
    atomicAdd(dst, (double)t);
  }
}
template <typename T>
__global__ void in_finalize_kernel(const double* acc, float* stats, const T* x, long n, long V, int C, double invV, float eps) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const long b = i / C, c = i - b * C;
    double m = acc[2 * i] * invV, var = acc[2 * i + 1] * invV - m * m;
    if (var < 0) var = 0;
    if (x) m += (double)to_f<T>(x[b * V * C + c]);  // shift used by the standalone reduction (none for epilogue-fused sums)
    stats[2 * i] = (float)m;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
static inline long in_vox_per_block(long V, int B = 4) {
  // <= 256 blocks per sample (their fp64 atomics hit the same 2*C addresses) -- up to 1024 blocks in total for small batches: one
  // workgroup per CU cannot hide its load latency (160^3, 1 sample: 286 -> ~150 us) -- but >= 128 voxels per block so small volumes
  // still spread over the chip
  const long per_sample = B >= 4 ? 256 : 1024 / (B < 1 ? 1 : B);
  long vpb = (V + per_sample - 1) / per_sample;
  vpb = (vpb + 63) / 64 * 64;
  return vpb < 128 ? 128 : vpb;
}
int k_in_stats(int dt, const void* x, float* stats, double* scratch, int B, long V, int C, float eps, hipStream_t st) {
  if (C % 8 || C / 8 > 256) return -2;
  hipError_t e = nmh_zero_async(scratch, sizeof(double) * 2 * B * C, st);
  if (e != hipSuccess) return (int)e;
  const long vpb = in_vox_per_block(V);
  dim3 grid((unsigned)((V + vpb - 1) / vpb), B);
  size_t lds = (4 * C + 32 * 256) * sizeof(float);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL((in_reduce_kernel<bf16_t, 0>), grid, dim3(256), lds, st, (const bf16_t*)x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, scratch, nullptr, V, C, 0.f, vpb);
  else hipLaunchKernelGGL((in_reduce_kernel<float, 0>), grid, dim3(256), lds, st, (const float*)x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, scratch, nullptr, V, C, 0.f, vpb);
  NMH_CHECK_LAUNCH();
  long n = (long)B * C;
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(in_finalize_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scratch, stats, (const bf16_t*)x, n, V, C, 1.0 / (double)V, eps);
  else hipLaunchKernelGGL(in_finalize_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scratch, stats, (const float*)x, n, V, C, 1.0 / (double)V, eps);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_in_finalize(int dt, const double* acc, float* stats, int B, long V, int C, float eps, hipStream_t st) {
  (void)dt;
  long n = (long)B * C;
  hipLaunchKernelGGL(in_finalize_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, acc, stats, (const float*)nullptr, n, V, C, 1.0 / (double)V, eps);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_in_bwd_reduce(int dt, const void* dout, const void* out, const void* x, const float* stats, const void* r, const float* stats_r, int rmode,
                    double* sums, double* sums_r, int B, long V, int C, float slope, hipStream_t st) {
  if (C % 8 || C / 8 > 256 || (!out && rmode != 0)) return -2;
  hipError_t e = nmh_zero_async(sums, sizeof(double) * 2 * B * C, st);
  if (e != hipSuccess) return (int)e;
  if (rmode == 2) { e = nmh_zero_async(sums_r, sizeof(double) * 2 * B * C, st); if (e != hipSuccess) return (int)e; }
  const long vpb = in_vox_per_block(V, B);
  dim3 grid((unsigned)((V + vpb - 1) / vpb), B);
  size_t lds = (4 * C + 32 * 256) * sizeof(float);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL((in_reduce_kernel<bf16_t, 1>), grid, dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)dout, (const bf16_t*)out, stats, (const bf16_t*)r, stats_r, rmode, sums, sums_r, V, C, slope, vpb);
  else hipLaunchKernelGGL((in_reduce_kernel<float, 1>), grid, dim3(256), lds, st, (const float*)x, (const float*)dout, (const float*)out, stats, (const float*)r, stats_r, rmode, sums, sums_r, V, C, slope, vpb);
  NMH_CHECK_LAUNCH();
  return 0;
}

#define IN_APPLY_VOX_PER_BLOCK 2048
#define IN_STREAM_MAX_C 96   // widest tensor the streaming form of the apply passes takes (LDS constant tables)
// voxels per block of the apply passes: 2048 on the big volumes, fewer on the coarse decoder levels so that >= ~1024 blocks exist
// streaming form of the apply passes (bf16 tensors larger than L2 + MALL, i.e. the 160^3 level -- 393 MB per grid): non-temporal loads and stores and SHORT
// blocks.  In the replayed step (same box, three alternating runs each): 8 grids 49.22 ms against 49.66 without (49.45 with short blocks but temporal
// accesses), 2 grids 16.71 against 16.97, 1 grid 11.06 against 11.18.  tools/probes/hbm_stream_probe.hip: a 2-read + 1-write pass over 3.1-GB tensors runs at 5.3 TB/s from 2048-4096 looping blocks, 6.0-6.25
// from blocks that touch 16-64 bytes per thread and tensor with non-temporal accesses.  NMH_IN_STREAM=0 disables, NMH_IN_STREAM_VPB sets the voxels per block.
static inline bool in_apply_streaming(long V, int B, int C, int dt) {
  static const int on = getenv("NMH_IN_STREAM") ? atoi(getenv("NMH_IN_STREAM")) : 1;
  static const double min_bytes = 1.0e6 * (getenv("NMH_IN_STREAM_MIN_MB") ? atof(getenv("NMH_IN_STREAM_MIN_MB")) : 300.0);
  return on && dt == NMH_DT_BF16 && C <= IN_STREAM_MAX_C && (double)V * B * C * 2 >= min_bytes;
}
// voxels per thread of a streaming block (mult x 256 / (C/8) voxels per block).  Measured at 8 x 160^3 x 48 (tools/bench_inbwd.py, tools/bench_tail.py):
// forward apply 1.23 ms looping -> 1.04 / 1.03 / 1.12 / 1.14 with 2 / 4 / 8 / 16; backward apply 1.78 -> 1.80 / 1.60 / 1.58 / 1.67; tail backward
// 2.14 -> 2.45 / 2.18 / 2.09 / 2.06 with 4 / 8 / 16 / 32 (its table is 8 arrays and two fp64 divisions per block)
static inline bool in_stream_nt() { static const int v = getenv("NMH_IN_STREAM_NT") ? atoi(getenv("NMH_IN_STREAM_NT")) : 1; return v != 0; }
static inline long in_apply_vpb(long V, int B, int C, bool streaming = false, int mult = 8) {
  const int NV = 256 / (C >> 3);
  if (streaming) {
    static const int sv = getenv("NMH_IN_STREAM_VPB") ? atoi(getenv("NMH_IN_STREAM_VPB")) : 0;
    return sv > 0 ? sv : (long)mult * NV;
  }
  long vpb = (V * B + 1023) / 1024;
  if (vpb < 4L * NV) vpb = 4L * NV;
  return vpb > IN_APPLY_VOX_PER_BLOCK ? IN_APPLY_VOX_PER_BLOCK : vpb;
}
template <typename T, bool ST = false, bool NT = false>   // ST: short streaming blocks (LDS constant table); NT: non-temporal accesses
__global__ __launch_bounds__(256) void in_apply_kernel(const T* x, const float* stats, const T* r, const float* stats_r, int rmode, T* out, long V, int C, float slope, long vpb) {
  // grid (voxel blocks, B); thread = (8-channel chunk cl, voxel lane vl): no integer division in the loop, statistics in registers
  const int CL = C >> 3, NV = 256 / CL;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL, b = blockIdx.y;
  float mu[8], rs[8], mur[8], rsr[8];
  if (ST) {   // short streaming blocks: the per-channel constants reach the threads through an LDS table (one global load per channel and block
    //           instead of 16-32 per thread: with 4-8 voxels per thread the per-thread prologue moved as many bytes through L2 as the pass itself)
    __shared__ float tab[4 * IN_STREAM_MAX_C];
    for (int i = threadIdx.x; i < C; i += 256) {
      const long sc = ((long)b * C + i) * 2;
      tab[i] = stats[sc]; tab[C + i] = stats[sc + 1];
      if (rmode == 2) { tab[2 * C + i] = stats_r[sc]; tab[3 * C + i] = stats_r[sc + 1]; }
    }
    __syncthreads();
    if (vl >= NV) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = tab[cl * 8 + j]; rs[j] = tab[C + cl * 8 + j];
      if (rmode == 2) { mur[j] = tab[2 * C + cl * 8 + j]; rsr[j] = tab[3 * C + cl * 8 + j]; }
    }
  } else {
    if (vl >= NV) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long sc = ((long)b * C + cl * 8 + j) * 2;
      mu[j] = stats[sc]; rs[j] = stats[sc + 1];
      if (rmode == 2) { mur[j] = stats_r[sc]; rsr[j] = stats_r[sc + 1]; }
    }
  }
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  constexpr int U = 2;
  for (long vb = v0 + vl; vb < v1; vb += (long)NV * U) {
    float xv[U][8], rv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
        const long o = ((long)b * V + v) * C + cl * 8;
        if (NT) { Vec8<T>::load_nt(x + o, xv[u]); if (rmode) Vec8<T>::load_nt(r + o, rv[u]); }
        else { Vec8<T>::load(x + o, xv[u]); if (rmode) Vec8<T>::load(r + o, rv[u]); }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = (xv[u][j] - mu[j]) * rs[j];
          if (rmode == 1) y += rv[u][j];
          else if (rmode == 2) y += (rv[u][j] - mur[j]) * rsr[j];
          xv[u][j] = y > 0.f ? y : slope * y;
        }
        if (NT) Vec8<T>::store_nt(out + ((long)b * V + v) * C + cl * 8, xv[u]);
        else Vec8<T>::store(out + ((long)b * V + v) * C + cl * 8, xv[u]);
      }
    }
  }
}
int k_in_apply(int dt, const void* x, const float* stats, const void* r, const float* stats_r, int rmode, void* out, int B, long V, int C, float slope, hipStream_t st) {
  if (C % 8 || C / 8 > 256) return -2;
  const bool nt = in_apply_streaming(V, B, C, dt);
  const long vpb = in_apply_vpb(V, B, C, nt, 4);
  dim3 grid((unsigned)((V + vpb - 1) / vpb), B);
  if (nt && in_stream_nt()) hipLaunchKernelGGL((in_apply_kernel<bf16_t, true, true>), grid, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)r, stats_r, rmode, (bf16_t*)out, V, C, slope, vpb);
  else if (nt) hipLaunchKernelGGL((in_apply_kernel<bf16_t, true, false>), grid, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)r, stats_r, rmode, (bf16_t*)out, V, C, slope, vpb);
  else if (dt == NMH_DT_BF16) hipLaunchKernelGGL(in_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)r, stats_r, rmode, (bf16_t*)out, V, C, slope, vpb);
  else hipLaunchKernelGGL(in_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, stats, (const float*)r, stats_r, rmode, (float*)out, V, C, slope, vpb);
  NMH_CHECK_LAUNCH();
  return 0;
}

// the arithmetic of the apply pass, pinned (no compiler-chosen contraction) so that every kernel that runs it -- the general one and the background launch
// below -- rounds the same way: g = dout * lrelu'(sign source), x-hat, dx = rstd * fma(-x-hat, S2/V, g - S1/V)
__device__ __forceinline__ void in_bwd_terms(float dv, float sgn, float xv, float mu, float rs, float slope, float& g, float& xh) {
#pragma clang fp contract(off)
  g = dv * (sgn > 0.f ? 1.0f : slope);
  xh = (xv - mu) * rs;
}
__device__ __forceinline__ float in_bwd_dx(float g, float xh, float rs, float m1, float m2) {
#pragma clang fp contract(off)
  return rs * __builtin_fmaf(-xh, m2, g - m1);
}

template <typename T, bool ST = false, bool NT = false>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* dout, const T* outp, const T* x, const float* stats, const double* sums, const T* r, const float* stats_r,
                                                            const double* sums_r, int rmode, T* dx, T* dr, int dr_acc, long V, int C, float slope, long vpb) {
  const int CL = C >> 3, NV = 256 / CL;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL, b = blockIdx.y;
  const float invV = 1.0f / (float)V;
  float mu[8], rs[8], m1[8], m2[8], mur[8], rsr[8], n1[8], n2[8];
  if (ST) {   // constants through an LDS table (see in_apply_kernel)
    __shared__ float tab[8 * IN_STREAM_MAX_C];
    for (int i = threadIdx.x; i < C; i += 256) {
      const long sc = ((long)b * C + i) * 2;
      tab[i] = stats[sc]; tab[C + i] = stats[sc + 1];
      tab[2 * C + i] = (float)sums[sc] * invV; tab[3 * C + i] = (float)sums[sc + 1] * invV;
      if (rmode == 2) { tab[4 * C + i] = stats_r[sc]; tab[5 * C + i] = stats_r[sc + 1]; tab[6 * C + i] = (float)sums_r[sc] * invV; tab[7 * C + i] = (float)sums_r[sc + 1] * invV; }
    }
    __syncthreads();
    if (vl >= NV) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cl * 8 + j;
      mu[j] = tab[c]; rs[j] = tab[C + c]; m1[j] = tab[2 * C + c]; m2[j] = tab[3 * C + c];
      if (rmode == 2) { mur[j] = tab[4 * C + c]; rsr[j] = tab[5 * C + c]; n1[j] = tab[6 * C + c]; n2[j] = tab[7 * C + c]; }
    }
  } else {
    if (vl >= NV) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long sc = ((long)b * C + cl * 8 + j) * 2;
      mu[j] = stats[sc]; rs[j] = stats[sc + 1];
      m1[j] = (float)sums[sc] * invV; m2[j] = (float)sums[sc + 1] * invV;
      if (rmode == 2) { mur[j] = stats_r[sc]; rsr[j] = stats_r[sc + 1]; n1[j] = (float)sums_r[sc] * invV; n2[j] = (float)sums_r[sc + 1] * invV; }
    }
  }
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  for (long v = v0 + vl; v < v1; v += NV) {
    const long o = ((long)b * V + v) * C + cl * 8;
    float dv[8], ov[8], xv[8], rv[8], od[8], orr[8];
    if (NT) {
      Vec8<T>::load_nt(dout + o, dv);
      if (outp) Vec8<T>::load_nt(outp + o, ov);
      Vec8<T>::load_nt(x + o, xv);
      if (rmode == 2) Vec8<T>::load_nt(r + o, rv);
      if (rmode == 1 && dr_acc) Vec8<T>::load_nt(dr + o, orr);
    } else {
      Vec8<T>::load(dout + o, dv);
      if (outp) Vec8<T>::load(outp + o, ov);
      Vec8<T>::load(x + o, xv);
      if (rmode == 2) Vec8<T>::load(r + o, rv);
      if (rmode == 1 && dr_acc) Vec8<T>::load(dr + o, orr);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g, xh;
      in_bwd_terms(dv[j], outp ? ov[j] : xv[j] - mu[j], xv[j], mu[j], rs[j], slope, g, xh);
      od[j] = in_bwd_dx(g, xh, rs[j], m1[j], m2[j]);
      if (rmode == 1) orr[j] = dr_acc ? orr[j] + g : g;
      else if (rmode == 2) {
        const float rh = (rv[j] - mur[j]) * rsr[j];
        orr[j] = rsr[j] * (g - n1[j] - rh * n2[j]);
      }
    }
    if (NT) { Vec8<T>::store_nt(dx + o, od); if (rmode) Vec8<T>::store_nt(dr + o, orr); }
    else { Vec8<T>::store(dx + o, od); if (rmode) Vec8<T>::store(dr + o, orr); }
  }
}
int k_in_bwd_apply(int dt, const void* dout, const void* out, const void* x, const float* stats, const double* sums, const void* r, const float* stats_r,
                   const double* sums_r, int rmode, void* dx, void* dr, int dr_accumulate, int B, long V, int C, float slope, hipStream_t st) {
  if (C % 8 || C / 8 > 256 || (!out && rmode != 0)) return -2;
  const bool nt = in_apply_streaming(V, B, C, dt);
  const long vpb = in_apply_vpb(V, B, C, nt);
  dim3 grid((unsigned)((V + vpb - 1) / vpb), B);
  if (nt && in_stream_nt())
    hipLaunchKernelGGL((in_bwd_apply_kernel<bf16_t, true, true>), grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x, stats, sums,
                       (const bf16_t*)r, stats_r, sums_r, rmode, (bf16_t*)dx, (bf16_t*)dr, dr_accumulate, V, C, slope, vpb);
  else if (nt)
    hipLaunchKernelGGL((in_bwd_apply_kernel<bf16_t, true, false>), grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x, stats, sums,
                       (const bf16_t*)r, stats_r, sums_r, rmode, (bf16_t*)dx, (bf16_t*)dr, dr_accumulate, V, C, slope, vpb);
  else if (dt == NMH_DT_BF16)
    hipLaunchKernelGGL(in_bwd_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x, stats, sums,
                       (const bf16_t*)r, stats_r, sums_r, rmode, (bf16_t*)dx, (bf16_t*)dr, dr_accumulate, V, C, slope, vpb);
  else
    hipLaunchKernelGGL(in_bwd_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)out, (const float*)x, stats, sums,
                       (const float*)r, stats_r, sums_r, rmode, (float*)dx, (float*)dr, dr_accumulate, V, C, slope, vpb);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- the apply pass as a BACKGROUND kernel (round 5; bf16, C = 48, rmode 0, sign from x: decoder-1's first InstanceNorm) -------------------------
// In decoder-1's backward this pass (9.4 GB, HBM-bound, no matrix work) sits between two persistent MFMA kernels that leave HBM two-thirds idle; it
// depends on conv2's input gradient only, conv2's weight gradient on neither.  One persistent 256-thread workgroup per CU with <= 96 VGPRs and no dynamic
// LDS is the footprint that fits on a CU BESIDE conv48_wgrad_kernel's workgroup (8 waves x 208 VGPRs, 109 KB): issued on a forked stream next to it, the
// pass streams under the weight gradient's MFMAs.  A workgroup owns one contiguous voxel range of one sample (constants loaded once); U iterations of
// (dout, x) chunks are in flight per thread.  Same arithmetic, in the same order, as in_bwd_apply_kernel: bit-identical output.
template <int U>
__global__ __launch_bounds__(256) void in_bwd_apply_bg_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                               const double* __restrict__ sums, bf16_t* __restrict__ dx, long V, float slope, int wps) {
  constexpr int C = 48, CL = 6, NV = 42;
  const int b = blockIdx.x / wps, w = blockIdx.x - b * wps;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL;
  if (vl >= NV) return;
  const float invV = 1.0f / (float)V;
  float mu[8], rs[8], m1[8], m2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long sc = ((long)b * C + cl * 8 + j) * 2;
    mu[j] = stats[sc]; rs[j] = stats[sc + 1];
    m1[j] = (float)sums[sc] * invV; m2[j] = (float)sums[sc + 1] * invV;
  }
  const long per = (V + wps - 1) / wps, v0 = (long)w * per;
  long v1 = v0 + per;
  if (v1 > V) v1 = V;
  const bf16_t* const db = dout + (long)b * V * C + cl * 8;
  const bf16_t* const xb = x + (long)b * V * C + cl * 8;
  bf16_t* const ob = dx + (long)b * V * C + cl * 8;
  for (long vb = v0 + vl; vb < v1; vb += (long)NV * U) {
    u32x4 dw[U], xw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
        dw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(db + v * C));
        xw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xb + v * C));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
        float od[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * i + h;
            const float dv = __uint_as_float(h ? dw[u][i] & 0xffff0000u : dw[u][i] << 16), xv = __uint_as_float(h ? xw[u][i] & 0xffff0000u : xw[u][i] << 16);
            float g, xh;
            in_bwd_terms(dv, xv - mu[j], xv, mu[j], rs[j], slope, g, xh);
            od[j] = in_bwd_dx(g, xh, rs[j], m1[j], m2[j]);
          }
        }
        Vec8<bf16_t>::store_nt(ob + v * C, od);
      }
    }
  }
}
// The same background launch for the CENTERED decoder1 (csrc/cconv.hip): the stored tensor is z = lrelu(t), t = y - mean, stats = (e = residual mean of t, rstd).
//   dx = rstd (g - m1 - xhat m2),  g = d lrelu'(t),  xhat = (t - e) rstd,  t = z (z > 0 ? 1 : 1 / slope)
//      = d (z > 0 ? A : A slope)  -  z (z > 0 ? Cc : Cc / slope)  -  Bc,     A = rstd, Cc = rstd^2 m2, Bc = rstd m1 - rstd^2 m2 e
// one compare, two selects between per-channel constants and two FMAs per element (the classic form needs eleven instructions: this launch lives on the issue
// slots the weight gradient beside it leaves free).  40 constants per lane: unroll 4 keeps it under the 96 registers that are free there.
template <int U>
__global__ __launch_bounds__(256) void in_bwd_apply_bg_centered_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ z, const float* __restrict__ stats,
                                                                        const double* __restrict__ sums, bf16_t* __restrict__ dx, long V, float slope, int wps) {
  constexpr int C = 48, CL = 6, NV = 42;
  const int b = blockIdx.x / wps, w = blockIdx.x - b * wps;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL;
  if (vl >= NV) return;
  const float invV = 1.0f / (float)V, inv_slope = 1.0f / slope;
  float A1[8], A2[8], C1[8], C2[8], Bc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long sc = ((long)b * C + cl * 8 + j) * 2;
    const float e = stats[sc], rs = stats[sc + 1], m1 = (float)sums[sc] * invV, m2 = (float)sums[sc + 1] * invV;
    A1[j] = rs; A2[j] = rs * slope;
    C1[j] = rs * rs * m2; C2[j] = C1[j] * inv_slope;
    Bc[j] = rs * m1 - C1[j] * e;
  }
  const long per = (V + wps - 1) / wps, v0 = (long)w * per;
  long v1 = v0 + per;
  if (v1 > V) v1 = V;
  const bf16_t* const db = dout + (long)b * V * C + cl * 8;
  const bf16_t* const zb = z + (long)b * V * C + cl * 8;
  bf16_t* const ob = dx + (long)b * V * C + cl * 8;
  for (long vb = v0 + vl; vb < v1; vb += (long)NV * U) {
    u32x4 dw[U], zw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
        dw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(db + v * C));
        zw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(zb + v * C));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = vb + (long)u * NV;
      if (v < v1) {
        float od[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int j = 2 * i + h;
            const float dv = __uint_as_float(h ? dw[u][i] & 0xffff0000u : dw[u][i] << 16), zv = __uint_as_float(h ? zw[u][i] & 0xffff0000u : zw[u][i] << 16);
            const bool pos = zv > 0.f;
            od[j] = __builtin_fmaf(-zv, pos ? C1[j] : C2[j], __builtin_fmaf(dv, pos ? A1[j] : A2[j], -Bc[j]));
          }
        }
        Vec8<bf16_t>::store_nt(ob + v * C, od);
      }
    }
  }
}
int k_in_bwd_apply_bg(int dt, const void* dout, const void* x, const float* stats, const double* sums, void* dx, int B, long V, int C, float slope, hipStream_t st, int centered) {
  if (dt != NMH_DT_BF16 || C != 48 || B < 1 || V < 1) return -2;
  if (centered) {
    if (!(slope > 0.f)) return -2;
    int wps = 256 / B;
    if (wps < 1) wps = 1;
    hipLaunchKernelGGL(in_bwd_apply_bg_centered_kernel<4>, dim3((unsigned)(B * wps)), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, stats, sums, (bf16_t*)dx, V, slope, wps);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  int wps = 256 / B;
  if (wps < 1) wps = 1;
  static const int unroll = [] { const char* e = getenv("NMH_INBWD_BG_U"); const int u = e ? atoi(e) : 5; return u < 3 ? 3 : (u > 5 ? 5 : u); }();   // test-only override, clamped to {3, 4, 5}:   // 74 / 82 / 90 VGPRs: 96 are free beside the weight gradient
  const dim3 grid((unsigned)(B * wps));
  if (unroll <= 3) hipLaunchKernelGGL(in_bwd_apply_bg_kernel<3>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, stats, sums, (bf16_t*)dx, V, slope, wps);
  else if (unroll == 4) hipLaunchKernelGGL(in_bwd_apply_bg_kernel<4>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, stats, sums, (bf16_t*)dx, V, slope, wps);
  else hipLaunchKernelGGL(in_bwd_apply_bg_kernel<5>, grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, stats, sums, (bf16_t*)dx, V, slope, wps);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- decoder tail backward: d0 = lrelu(IN(x) + r) -> 1x1 head -> loss ----------------------------------------------------------
// d(d0)[v][c] = sum_o dp[v][o] Wout[o][c] is a 4-term dot per element, so it is recomputed from the per-voxel d(pred) (16 B per voxel,
// written by the loss forward) instead of being written by one kernel and re-read by two.  Pass 0 (APPLY=0): IN-backward sums
// {sum g, sum g*xhat} with g = d(d0) * lrelu'(d0), plus the head weight gradient dW[o][c] = sum_v dp[v][o] d0[v][c];
// pass 1 (APPLY=1): dx = rstd (g - S1/V - xhat S2/V), dr = g.  Same thread mapping as the InstanceNorm kernels above.
template <typename T, int APPLY, bool ST = false, bool NT = false>
__global__ __launch_bounds__(256, ST ? 4 : 1) void tail_bwd_kernel(const T* __restrict__ d0, const T* __restrict__ rres, const T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ dp,
                                                       const double* __restrict__ lsums, const float* __restrict__ Wout, double* in_sums, T* __restrict__ dx,
                                                       T* __restrict__ dr, float slope, float* dWout, float* dbout, long V, int C, long vpb,
                                                       const unsigned char* __restrict__ smask) {
  extern __shared__ float sred[];  // [6][C]
  const int CL = C >> 3, NV = 256 / CL;
  const int cl = threadIdx.x % CL, vl = threadIdx.x / CL, b = blockIdx.y;
  if (!APPLY) {
    for (int i = threadIdx.x; i < 6 * C; i += 256) sred[i] = 0.f;
    __syncthreads();
  }
  float w[4][8], mu[8], rs[8], u1[8], u2[8], wacc[APPLY ? 1 : 4][8];
  float inv_occ = 0.f, inv_rm = 0.f;
  if (ST) {   // constants (incl. the scaled head weights: two fp64 divisions per BLOCK instead of per thread) through an LDS table (see in_apply_kernel)
    __shared__ float tab[8 * IN_STREAM_MAX_C];
    for (int i = threadIdx.x; i < C; i += 256) {
      const long sc = ((long)b * C + i) * 2;
      const float invV = 1.0f / (float)V, io = (float)(1.0 / lsums[1]), ir = (float)(1.0 / lsums[3]);
      tab[i] = stats[sc]; tab[C + i] = stats[sc + 1];
      tab[2 * C + i] = (float)in_sums[sc] * invV; tab[3 * C + i] = (float)in_sums[sc + 1] * invV;
#pragma unroll
      for (int o = 0; o < 4; ++o) tab[(4 + o) * C + i] = Wout[o * C + i] * (o < 3 ? io : ir);
    }
    __syncthreads();
    if (vl < NV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cl * 8 + j;
        mu[j] = tab[c]; rs[j] = tab[C + c]; u1[j] = tab[2 * C + c]; u2[j] = tab[3 * C + c];
#pragma unroll
        for (int o = 0; o < 4; ++o) w[o][j] = tab[(4 + o) * C + c];
      }
    }
  } else {
    inv_occ = (float)(1.0 / lsums[1]); inv_rm = (float)(1.0 / lsums[3]);
    if (vl < NV) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long sc = ((long)b * C + cl * 8 + j) * 2;
        mu[j] = stats[sc]; rs[j] = stats[sc + 1];
        u1[j] = u2[j] = 0.f;
        if (APPLY) { const float invV = 1.0f / (float)V; u1[j] = (float)in_sums[sc] * invV; u2[j] = (float)in_sums[sc + 1] * invV; }
#pragma unroll
        for (int o = 0; o < 4; ++o) { w[o][j] = Wout[o * C + cl * 8 + j] * (o < 3 ? inv_occ : inv_rm); if (!APPLY) wacc[o][j] = 0.f; }
      }
    }
  }
  if (vl < NV) {
    const long v0 = (long)blockIdx.x * vpb;
    long v1 = v0 + vpb;
    if (v1 > V) v1 = V;
    constexpr int U = 2;
    for (long vb = v0 + vl; vb < v1; vb += (long)NV * U) {
      float ov[ST ? 1 : U][8], xv[U][8];
      unsigned mb[U];   // ST (the streaming apply pass, always with the sign mask): the mask byte stays packed -- 8 registers less per voxel row in flight, 128
      //                   VGPRs = four waves per SIMD instead of three
      float4 dq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vb + (long)u * NV;
        if (v < v1) {
          const long o = ((long)b * V + v) * C + cl * 8;
          if (ST) {
            const uint2 mw = *reinterpret_cast<const uint2*>(smask + ((long)b * V + v) * 8);
            mb[u] = ((cl < 4 ? mw.x : mw.y) >> (8 * (cl & 3))) & 0xffu;
          } else if (APPLY && smask) {   // only the sign of d0 is needed (the sums came out of the forward pass): one byte instead of 16
            const uint2 mw = *reinterpret_cast<const uint2*>(smask + ((long)b * V + v) * 8);   // 8-byte rows (C = 48: 6 used), one load per voxel row
            const unsigned m8 = ((cl < 4 ? mw.x : mw.y) >> (8 * (cl & 3))) & 0xffu;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[ST ? 0 : u][j] = (m8 >> j) & 1u ? 1.0f : -1.0f;
          } else Vec8<T>::load((d0 ? d0 : rres) + o, ov[ST ? 0 : u]);
          if (NT) Vec8<T>::load_nt(x + o, xv[u]); else Vec8<T>::load(x + o, xv[u]);
          dq[u] = *reinterpret_cast<const float4*>(dp + ((long)b * V + v) * 4);
          if (!ST && !d0 && !(APPLY && smask)) {  // d0 was not stored by the forward: rebuild it bit-exactly (same fp32 expression, same rounding) from x and r
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float y = (xv[u][j] - mu[j]) * rs[j] + ov[ST ? 0 : u][j];
              y = y > 0.f ? y : slope * y;
              ov[ST ? 0 : u][j] = to_f<T>(from_f<T>(y));
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vb + (long)u * NV;
        if (v < v1) {
          float gq[8], od[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = dq[u].x * w[0][j] + dq[u].y * w[1][j] + dq[u].z * w[2][j] + dq[u].w * w[3][j];
            const float g = d * ((ST ? ((mb[u] >> j) & 1u) != 0u : ov[ST ? 0 : u][j] > 0.f) ? 1.0f : slope);
            const float xh = (xv[u][j] - mu[j]) * rs[j];
            if (!APPLY) {
              u1[j] += g; u2[j] += g * xh;
              wacc[0][j] += dq[u].x * ov[ST ? 0 : u][j]; wacc[1][j] += dq[u].y * ov[ST ? 0 : u][j]; wacc[2][j] += dq[u].z * ov[ST ? 0 : u][j]; wacc[3][j] += dq[u].w * ov[ST ? 0 : u][j];
            } else {
              gq[j] = g;
              od[j] = rs[j] * (g - u1[j] - xh * u2[j]);
            }
          }
          if (APPLY) {
            const long o = ((long)b * V + v) * C + cl * 8;
            if (NT) { Vec8<T>::store_nt(dx + o, od); Vec8<T>::store_nt(dr + o, gq); }
            else { Vec8<T>::store(dx + o, od); Vec8<T>::store(dr + o, gq); }
          }
        }
      }
    }
    if (!APPLY) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&sred[cl * 8 + j], u1[j]);
        atomicAdd(&sred[C + cl * 8 + j], u2[j]);
#pragma unroll
        for (int o = 0; o < 4; ++o) atomicAdd(&sred[(2 + o) * C + cl * 8 + j], wacc[o][j]);
      }
    }
  }
  if (APPLY) return;
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    atomicAdd(&in_sums[((long)b * C + i) * 2], (double)sred[i]);
    atomicAdd(&in_sums[((long)b * C + i) * 2 + 1], (double)sred[C + i]);
  }
  for (int i = threadIdx.x; i < 4 * C; i += 256) atomicAdd(&dWout[i], sred[2 * C + i] * (i / C < 3 ? inv_occ : inv_rm));
  if (blockIdx.x == 0 && b == 0 && threadIdx.x < 4) atomicAdd(&dbout[threadIdx.x], (float)(lsums[4 + threadIdx.x] * (threadIdx.x < 3 ? 1.0 / lsums[1] : 1.0 / lsums[3])));
}
// the forward's un-normalised reductions (k_tail_fwd, bwd_sums) -> in_sums[b][c] = {sum g, sum g*xhat}, dWout += , dbout += , with the loss normalisers
__global__ void tail_sums_finalize_kernel(const double* __restrict__ bs, const double* __restrict__ lsums, double* __restrict__ in_sums, float* __restrict__ dWout,
                                          float* __restrict__ dbout, int B, int C) {
  const double inv_occ = 1.0 / lsums[1], inv_rm = 1.0 / lsums[3];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * C) {
    in_sums[2 * i] = bs[4 * i] * inv_occ + bs[4 * i + 2] * inv_rm;
    in_sums[2 * i + 1] = bs[4 * i + 1] * inv_occ + bs[4 * i + 3] * inv_rm;
  } else if (i < B * C + 4 * C) {
    const int k = i - B * C;
    dWout[k] += (float)(bs[(long)B * C * 4 + k] * (k / C < 3 ? inv_occ : inv_rm));
  } else if (i < B * C + 4 * C + 4) {
    const int o = i - B * C - 4 * C;
    dbout[o] += (float)(lsums[4 + o] * (o < 3 ? inv_occ : inv_rm));
  }
}
int k_tail_bwd(int dt, const void* d0, const void* r, const void* xin, const float* in_stats, const float* dp, const double* loss_sums, const float* Wout, double* in_sums,
               void* dx, void* dr, float slope, float* dWout, float* dbout, int B, long V, int C, const double* bwd_sums, hipStream_t st, const unsigned char* sign_mask) {
  if (C % 8 || C / 8 > 256 || (!d0 && !r && !(sign_mask && bwd_sums))) return -2;
  if (sign_mask && (!bwd_sums || C != 48)) return -4;   // the sums pass needs d0 itself; the mask has 8-byte rows for 48 channels
  const long vpb = in_vox_per_block(V, B);
  const bool nt = bwd_sums && sign_mask && in_apply_streaming(V, B, C, dt);
  static const int tail_sv = getenv("NMH_TAIL_STREAM_VPB") ? atoi(getenv("NMH_TAIL_STREAM_VPB")) : 0;
  const long vpa = nt && tail_sv > 0 ? tail_sv : in_apply_vpb(V, B, C, nt, 16);   // (round 5, four waves per SIMD: 1.93 ms with 16, 1.98 / 1.99 / 2.04 with 32 / 64 / 128)
  dim3 g0((unsigned)((V + vpb - 1) / vpb), B), g1((unsigned)((V + vpa - 1) / vpa), B);
  if (bwd_sums) {   // the reductions were taken by the forward pass: one apply pass is all that is left
    const int n = B * C + 4 * C + 4;
    hipLaunchKernelGGL(tail_sums_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bwd_sums, loss_sums, in_sums, dWout, dbout, B, C);
    if (nt && in_stream_nt())
      hipLaunchKernelGGL((tail_bwd_kernel<bf16_t, 1, true, true>), g1, dim3(256), 0, st, (const bf16_t*)d0, (const bf16_t*)r, (const bf16_t*)xin, in_stats, dp, loss_sums, Wout, in_sums, (bf16_t*)dx,
                         (bf16_t*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
    else if (nt)
      hipLaunchKernelGGL((tail_bwd_kernel<bf16_t, 1, true, false>), g1, dim3(256), 0, st, (const bf16_t*)d0, (const bf16_t*)r, (const bf16_t*)xin, in_stats, dp, loss_sums, Wout, in_sums, (bf16_t*)dx,
                         (bf16_t*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
    else if (dt == NMH_DT_BF16)
      hipLaunchKernelGGL((tail_bwd_kernel<bf16_t, 1>), g1, dim3(256), 0, st, (const bf16_t*)d0, (const bf16_t*)r, (const bf16_t*)xin, in_stats, dp, loss_sums, Wout, in_sums, (bf16_t*)dx,
                         (bf16_t*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
    else
      hipLaunchKernelGGL((tail_bwd_kernel<float, 1>), g1, dim3(256), 0, st, (const float*)d0, (const float*)r, (const float*)xin, in_stats, dp, loss_sums, Wout, in_sums, (float*)dx,
                         (float*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  hipError_t e = nmh_zero_async(in_sums, sizeof(double) * 2 * B * C, st);
  if (e != hipSuccess) return (int)e;
  const size_t lds = 6 * C * sizeof(float);
  if (dt == NMH_DT_BF16) {
    hipLaunchKernelGGL((tail_bwd_kernel<bf16_t, 0>), g0, dim3(256), lds, st, (const bf16_t*)d0, (const bf16_t*)r, (const bf16_t*)xin, in_stats, dp, loss_sums, Wout, in_sums, (bf16_t*)nullptr,
                       (bf16_t*)nullptr, slope, dWout, dbout, V, C, vpb, (const unsigned char*)nullptr);
    hipLaunchKernelGGL((tail_bwd_kernel<bf16_t, 1>), g1, dim3(256), 0, st, (const bf16_t*)d0, (const bf16_t*)r, (const bf16_t*)xin, in_stats, dp, loss_sums, Wout, in_sums, (bf16_t*)dx,
                       (bf16_t*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
  } else {
    hipLaunchKernelGGL((tail_bwd_kernel<float, 0>), g0, dim3(256), lds, st, (const float*)d0, (const float*)r, (const float*)xin, in_stats, dp, loss_sums, Wout, in_sums, (float*)nullptr,
                       (float*)nullptr, slope, dWout, dbout, V, C, vpb, (const unsigned char*)nullptr);
    hipLaunchKernelGGL((tail_bwd_kernel<float, 1>), g1, dim3(256), 0, st, (const float*)d0, (const float*)r, (const float*)xin, in_stats, dp, loss_sums, Wout, in_sums, (float*)dx,
                       (float*)dr, slope, dWout, dbout, V, C, vpa, sign_mask);
  }
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- decoder tail forward: d0 = lrelu(IN(x) + r) -> 1x1 head -> loss terms, one pass ------------------------------------------
// The normalisation pass already holds every d0 chunk in registers, so the 1x1 head (C -> 4) is evaluated there: each lane forms the
// partial dot of its 8 channels, the C/8 lanes of a voxel are summed with ds_bpermute (a wave holds floor(64/(C/8)) whole voxels, the
// rest of its lanes idle), and the voxel's leader lane evaluates the loss terms / writes d(pred) -- the separate loss kernel (one more
// read of d0) disappears.  Same math as loss_kernel<T,0> (misc.hip); the head sees the stored (rounded) d0, as the backward does.
// BS (a.bwd_sums): the reductions of the tail BACKWARD are taken here as well -- the pass already holds d0 and x-hat in registers and
// learns d(pred) per voxel, so {sum g, sum g*xhat} of the last InstanceNorm and the head's weight gradient come out of the forward
// (kept separately for the RGB and the alpha part of the loss: their normalisers are only known once the pass is complete), and the
// backward is a single apply pass (k_tail_bwd with bwd_sums): one 3-tensor read pass over 160^3 x C less per step.
template <typename T, bool BS>
__global__ __launch_bounds__(256) void tail_fwd_kernel(const T* __restrict__ x, const float* __restrict__ stats, const T* __restrict__ r, T* __restrict__ out, LossArgs a,
                                                       long V, int C, float slope, long vpb) {
  __shared__ float sacc[8];
  extern __shared__ float slab[];   // BS: [64][256] block-reduction slab
  const int CL = C >> 3, VPW = 64 / CL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
  const int vl = lane / CL, cl = lane - vl * CL;
  const bool active = vl < VPW;
  const bool leader = active && cl == 0;
  int p2half = 1;
  while (p2half * 2 < CL) p2half *= 2;   // largest power of two below CL (CL = 6 -> 4)
  if (threadIdx.x < 8) sacc[threadIdx.x] = 0.f;
  __syncthreads();
  float mu[8], rs[8], w[4][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = (active ? cl : 0) * 8 + j;
    mu[j] = stats[((long)b * C + c) * 2]; rs[j] = stats[((long)b * C + c) * 2 + 1];
#pragma unroll
    for (int o = 0; o < 4; ++o) w[o][j] = a.Wout[o * C + c];
  }
  const float b0 = a.bout[0], b1 = a.bout[1], b2 = a.bout[2], b3 = a.bout[3];
  const int e0 = a.extents[b * 3], e1 = a.extents[b * 3 + 1], e2 = a.extents[b * 3 + 2];
  const unsigned Ru = (unsigned)a.R;
  const int g = a.R >> 2;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bs[BS ? 8 : 1][8];   // BS: rows 0..3 = {s1_rgb, s2_rgb, s1_a, s2_a}, rows 4..7 = head weight-gradient sums for the 4 outputs
  if (BS) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) bs[BS ? k : 0][j] = 0.f;
  }
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  // software pipeline: the loads of iteration i+1 (x, r chunks as raw 16-byte words; the leader's target / mask values) are issued before
  // iteration i is processed -- with the backward sums this kernel runs at 2 waves per SIMD and every iteration is one long dependent
  // chain (dot -> shuffle tree -> loss terms -> broadcast -> accumulate), so the memory latency has to be hidden inside the wave
  constexpr int RW = sizeof(T) == 2 ? 1 : 2;     // 16-byte words per 8-element chunk
  uint4 nx[RW], nr[RW];
  float nt0 = 0.f, nt1 = 0.f, nt2 = 0.f, nt3 = 0.f;
  unsigned char ntm = 0;
  auto issue = [&](long bs0) {
    const long vq = bs0 + vl;
#pragma unroll
    for (int q = 0; q < RW; ++q) { nx[q] = make_uint4(0, 0, 0, 0); nr[q] = make_uint4(0, 0, 0, 0); }
    nt0 = nt1 = nt2 = nt3 = 0.f; ntm = 0;
    if (active && vq < v1) {
      const long o = ((long)b * V + vq) * C + cl * 8;
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        nx[q] = reinterpret_cast<const uint4*>(x + o)[q];
        nr[q] = reinterpret_cast<const uint4*>(r + o)[q];
      }
      if (cl == 0) {
        const unsigned vox = (unsigned)vq, tq = vox / Ru, zq = tq / Ru;
        const int xq = (int)(vox - tq * Ru), yq_ = (int)(tq - zq * Ru), zq_ = (int)zq;
        const float* tg = a.target + (long)b * 4 * V + vq;
        nt0 = tg[0]; nt1 = tg[V]; nt2 = tg[2 * V]; nt3 = tg[3 * V];
        ntm = a.tokmask[((zq_ >> 2) * g + (yq_ >> 2)) * g + (xq >> 2)];
      }
    }
  };
  auto unpack = [&](const uint4 (&wv)[RW], float (&f)[8]) {
    if constexpr (sizeof(T) == 2) {
      const unsigned ww[4] = {wv[0].x, wv[0].y, wv[0].z, wv[0].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(ww[i] << 16); f[2 * i + 1] = __uint_as_float(ww[i] & 0xffff0000u); }
    } else {
      f[0] = __uint_as_float(wv[0].x); f[1] = __uint_as_float(wv[0].y); f[2] = __uint_as_float(wv[0].z); f[3] = __uint_as_float(wv[0].w);
      f[4] = __uint_as_float(wv[RW - 1].x); f[5] = __uint_as_float(wv[RW - 1].y); f[6] = __uint_as_float(wv[RW - 1].z); f[7] = __uint_as_float(wv[RW - 1].w);
    }
  };
  if (v0 + wave * VPW < v1) issue(v0 + wave * VPW);
  for (long base = v0 + wave * VPW; base < v1; base += 4 * VPW) {   // wave-uniform trip count: the shuffles below need every lane
    const long v = base + vl;
    const bool ok = active && v < v1;
    float xv[8], rv[8];
    unpack(nx, xv);
    unpack(nr, rv);
    const float t0 = nt0, t1 = nt1, t2 = nt2, t3 = nt3;
    const unsigned char tmk = ntm;
    int zz = 0, yy = 0, xx = 0;
    if (ok && cl == 0) {
      const unsigned vox = (unsigned)v, tq = vox / Ru, zq = tq / Ru;
      xx = (int)(vox - tq * Ru); yy = (int)(tq - zq * Ru); zz = (int)zq;
    }
    if (base + 4 * VPW < v1) issue(base + 4 * VPW);
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    float xh[BS ? 8 : 1], yq[BS ? 8 : 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xhat = (xv[j] - mu[j]) * rs[j];
      float y = xhat + rv[j];
      y = y > 0.f ? y : slope * y;
      xv[j] = y;
      const float yr = to_f<T>(from_f<T>(y));   // the value the backward (and the reference's next op) sees
      if (BS) { xh[BS ? j : 0] = xhat; yq[BS ? j : 0] = yr; }
      p[0] += yr * w[0][j]; p[1] += yr * w[1][j]; p[2] += yr * w[2][j]; p[3] += yr * w[3][j];
    }
    if (ok && out) Vec8<T>::store(out + ((long)b * V + v) * C + cl * 8, xv);
    // only the voxel's leader lane needs the four sums: a shift-down tree over the CL lanes of the voxel (log2 steps instead of CL
    // broadcasts; partial sums in lanes the leader never reads may include neighbours' values)
    float tot[4] = {p[0], p[1], p[2], p[3]};
    for (int sft = p2half; sft >= 1; sft >>= 1) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float q = __shfl_down(tot[o], sft, 64);
        if (cl + sft < CL) tot[o] += q;
      }
    }
    if (ok && cl == 0) {
      const float p0 = tot[0] + b0, p1 = tot[1] + b1, p2 = tot[2] + b2, p3 = tot[3] + b3;
      const bool occ = t3 > 0.01f;
      const bool rm = zz < e0 && yy < e1 && xx < e2 && tmk != 0;
      const float sg = 1.0f / (1.0f + __expf(-p3));
      if (occ) { acc[0] += (p0 - t0) * (p0 - t0) + (p1 - t1) * (p1 - t1) + (p2 - t2) * (p2 - t2); acc[1] += 1.f; }
      if (rm) { acc[2] += (sg - t3) * (sg - t3); acc[3] += 1.f; }
      if (a.pred) { float* pr = a.pred + (long)b * 4 * V + v; pr[0] = p0; pr[V] = p1; pr[2 * V] = p2; pr[3 * V] = p3; }
      if (a.dp) {
        float4 d;
        d.x = occ ? 2.f * (p0 - t0) : 0.f; d.y = occ ? 2.f * (p1 - t1) : 0.f; d.z = occ ? 2.f * (p2 - t2) : 0.f;
        d.w = rm ? 2.f * (sg - t3) * sg * (1.f - sg) : 0.f;
        *reinterpret_cast<float4*>(a.dp + ((long)b * V + v) * 4) = d;
        acc[4] += d.x; acc[5] += d.y; acc[6] += d.z; acc[7] += d.w;
        if (BS) { tot[0] = d.x; tot[1] = d.y; tot[2] = d.z; tot[3] = d.w; }   // handed to the voxel's other lanes below
      }
    }
    if (BS) {
      const int src = lane - cl;   // the voxel's leader lane
      const float d0x = __shfl(tot[0], src, 64), d0y = __shfl(tot[1], src, 64), d0z = __shfl(tot[2], src, 64), d0w = __shfl(tot[3], src, 64);
      if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float lr = yq[BS ? j : 0] > 0.f ? 1.0f : slope;
          const float gr = (d0x * w[0][j] + d0y * w[1][j] + d0z * w[2][j]) * lr, ga = d0w * w[3][j] * lr;
          const float xq = xh[BS ? j : 0], yv = yq[BS ? j : 0];
          bs[0][j] += gr; bs[BS ? 1 : 0][j] += gr * xq; bs[BS ? 2 : 0][j] += ga; bs[BS ? 3 : 0][j] += ga * xq;
          bs[BS ? 4 : 0][j] += d0x * yv; bs[BS ? 5 : 0][j] += d0y * yv; bs[BS ? 6 : 0][j] += d0z * yv; bs[BS ? 7 : 0][j] += d0w * yv;
        }
      }
    }
  }
  if (BS) {
    // block reduction through an LDS slab (no same-address atomics): thread t parks its 64 partials in column t; the sums of a channel
    // chunk cl live in the threads wave*64 + vl*CL + cl
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) slab[(k * 8 + j) * 256 + threadIdx.x] = active ? bs[BS ? k : 0][j] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * CL; i += 256) {
      const int kk = i / CL, c8 = i - kk * CL, k = kk >> 3, j = kk & 7;
      float t = 0.f;
      for (int wv = 0; wv < 4; ++wv)
        for (int q = 0; q < VPW; ++q) t += slab[kk * 256 + wv * 64 + q * CL + c8];
      const int c = c8 * 8 + j;
      double* dst = k < 4 ? a.bwd_sums + ((long)b * C + c) * 4 + k : a.bwd_sums + (long)a.B * C * 4 + (long)(k - 4) * C + c;
      atomicAdd(dst, (double)t);
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const float s = wave_sum(leader ? acc[o] : 0.f);
    if (lane == 0) atomicAdd(&sacc[o], s);
  }
  __syncthreads();
  if (threadIdx.x < (a.dp ? 8 : 4)) atomicAdd(&a.sums[threadIdx.x], (double)sacc[threadIdx.x]);
}
// ---- the same pass on the matrix cores (bf16, C = 48, training: d(pred) + backward sums, d0 not stored) ------------------------------------
// tail_fwd_kernel<bf16, true> is VALU-bound (354 packed-fp32 instructions per 10 voxels, 2.4 TB/s): the 48 -> 4 head is a dot product per lane
// plus a shuffle tree, the loss terms run on one lane in six, and the backward sums are 128 FMAs per lane and iteration.  Here a wave
// takes 32 voxels per step as two 16-voxel MFMA tiles:
//  * lane (vi, g) loads the 16-byte chunk g of voxel vi of either tile (channels 8g..8g+7) plus one chunk of channels 32..47 (lanes g < 2:
//    tile 0, g >= 2: tile 1) -- three dense wave-loads per operand and step -- and forms x-hat, d0 = lrelu(x-hat + r) rounded to bf16;
//  * the packed d0 chunks ARE the A fragments of the head GEMM P[voxel][o] = sum_c d0[c] W[o][c] (k-slot = channel; the shared third chunk
//    is multiplied by a B fragment that is zero on the other tile's k-slots); W as bf16 hi + lo parts (fp32-exact to 2^-17);
//  * the product leaves lane (o, g) with P[voxels 4g..4g+3][o]: lanes o < 4 evaluate the loss terms and d(pred) of four consecutive voxels of
//    ONE output each (16-byte target loads along x);
//  * their d(pred) values, as they sit, are the A fragment D[o][voxel] of the reduction GEMMs S_q[o][c] = sum_v D[o][v] F_q[v][c] over the
//    32 voxels, F = {[d0>0], [d0>0] x-hat, x-hat, d0} in bf16 ([d0>0] and d0 exact), written by the loading lanes to a wave-private LDS image
//    [voxel][48] (96-byte rows: conflict-free ds_read_b64_tr_b16) and read back transposed as B fragments: 12 MFMAs per 32 voxels replace
//    the per-lane FMAs.  The LeakyReLU slope enters as lr = slope + (1 - slope) [d0 > 0] when the block combines its sums, so no operand
//    carries the inexact bf16 value of the slope.
template <bool NT>
__global__ __launch_bounds__(256) void tail_fwd_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats, const bf16_t* __restrict__ r, LossArgs a,
                                                            long V, float slope, long vpb) {
  constexpr int C = 48, OPB = 32 * 96, WLDS = 4 * OPB;   // per wave: 4 operand images of 32 voxel rows x 96 bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float sacc[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, vi = lane & 15, g = lane >> 4, b = blockIdx.y;
  char* const wl = smem + wave * WLDS;
  if (tid < 8) sacc[tid] = 0.f;
  __syncthreads();
  // channel chunks of this lane: sets 0 / 1 = channels 8g.. of tile 0 / 1, set 2 = channels 32 + 8(g&1).. of tile g>>1
  const int c0 = 8 * g, c1 = 32 + 8 * (g & 1), t2 = g >> 1;
  typedef float tf2 __attribute__((ext_vector_type(2)));
  typedef short ts2 __attribute__((ext_vector_type(2)));
  tf2 mu0[4], rs0[4], mu1[4], rs1[4];   // channel pairs, mu* = MINUS the mean: the normalisation runs on packed fp32 (v_pk_add_f32 / v_pk_mul_f32)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mu0[j >> 1][j & 1] = -stats[((long)b * C + c0 + j) * 2]; rs0[j >> 1][j & 1] = stats[((long)b * C + c0 + j) * 2 + 1];
    mu1[j >> 1][j & 1] = -stats[((long)b * C + c1 + j) * 2]; rs1[j >> 1][j & 1] = stats[((long)b * C + c1 + j) * 2 + 1];
  }
  const tf2 slope2 = {slope, slope};
  // head weight fragments (B operand: lane (o = vi, g), k-slot j = channel): k-step 0 = channels 8g + j; the shared chunk: tile 0 reads
  // lanes g < 2 (channels 32 + 8g + j), tile 1 lanes g >= 2 (channels 32 + 8(g-2) + j); hi + lo bf16 parts of the fp32 weights
  Frag<bf16_t> w0h, w0l, wah, wal, w1h, w1l, wbh, wbl;   // w0*/wa*: tile 0 (output o in lane vi = o); w1*/wb*: tile 1 (output o in lane vi = 4 + o)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ow = vi & 3;
    const float wa = vi < 8 ? a.Wout[ow * C + c0 + j] : 0.f, wb = vi < 8 ? a.Wout[ow * C + c1 + j] : 0.f;
    const bf16_t ah = f2bf(wa), bh = f2bf(wb);
    const bf16_t al = f2bf(wa - bf2f(ah)), bl = f2bf(wb - bf2f(bh));
    const bool t0 = vi < 4, t1 = vi >= 4 && vi < 8;
    w0h.v[j] = t0 ? (short)ah : (short)0; w0l.v[j] = t0 ? (short)al : (short)0;
    w1h.v[j] = t1 ? (short)ah : (short)0; w1l.v[j] = t1 ? (short)al : (short)0;
    wah.v[j] = (t0 && g < 2) ? (short)bh : (short)0; wal.v[j] = (t0 && g < 2) ? (short)bl : (short)0;
    wbh.v[j] = (t1 && g >= 2) ? (short)bh : (short)0; wbl.v[j] = (t1 && g >= 2) ? (short)bl : (short)0;
  }
  const float bias_l = a.bout[vi & 3];                // the loss part runs on lanes vi < 8: output vi & 3 of tile vi >> 2
  const int e0 = a.extents[b * 3], e1 = a.extents[b * 3 + 1], e2 = a.extents[b * 3 + 2];
  const unsigned Ru = (unsigned)a.R;
  const int gd = a.R >> 2;
  f32x4 S[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int n = 0; n < 3; ++n) S[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lsq = 0.f, cnt = 0.f, dsum = 0.f;
  const long v0 = (long)blockIdx.x * vpb;
  long v1 = v0 + vpb;
  if (v1 > V) v1 = V;
  const bf16_t* const xb = x + (long)b * V * C;
  const bf16_t* const rb = r + (long)b * V * C;
  // Everything a step reads from global memory is requested one step ahead (round 5).  Before, the loss lanes loaded their targets and their token-mask
  // byte inside the step: vector-memory loads return in order, so that wait also drained the x / r prefetch of the next step issued a moment earlier --
  // every step stalled for a full memory latency (VALU active 0.44, 3.9 TB/s).
  struct Ops { uint4 x[3], r[3]; float4 tg, t3; unsigned tm; int xk; };
  Ops nxt;
  const unsigned xoff = (unsigned)(vi * C + c0) * 2u, xoff2 = (unsigned)((16 * t2 + vi) * C + c1) * 2u;   // byte offsets inside a step's 32 voxel rows
  const int tl_ = (vi >> 2) & 1, ol_ = vi & 3;   // the loss part runs on lanes vi < 8 (tile vi >> 2, output vi & 3); lanes vi >= 8 request the same words
  auto issue = [&](long base, Ops& o) {   // base + 32 <= V (V is a multiple of 64: R is a multiple of 4)
    const char* const xs_ = reinterpret_cast<const char*>(xb) + base * (C * 2);   // uniform part of the address: scalar registers
    const char* const rs_ = reinterpret_cast<const char*>(rb) + base * (C * 2);
    if (NT) {   // one-pass operands far larger than the caches: non-temporal
      auto ldn = [](const char* p) { const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(w[0], w[1], w[2], w[3]); };
      o.x[0] = ldn(xs_ + xoff); o.r[0] = ldn(rs_ + xoff); o.x[1] = ldn(xs_ + (xoff + 16u * C * 2u)); o.r[1] = ldn(rs_ + (xoff + 16u * C * 2u));
      o.x[2] = ldn(xs_ + xoff2); o.r[2] = ldn(rs_ + xoff2);
    } else {
      o.x[0] = *reinterpret_cast<const uint4*>(xs_ + xoff); o.r[0] = *reinterpret_cast<const uint4*>(rs_ + xoff);
      o.x[1] = *reinterpret_cast<const uint4*>(xs_ + (xoff + 16u * C * 2u)); o.r[1] = *reinterpret_cast<const uint4*>(rs_ + (xoff + 16u * C * 2u));
      o.x[2] = *reinterpret_cast<const uint4*>(xs_ + xoff2); o.r[2] = *reinterpret_cast<const uint4*>(rs_ + xoff2);
    }
    const long vq = base + 16 * tl_ + 4 * g;
    const unsigned vox = (unsigned)vq, tq = vox / Ru, zq = tq / Ru;
    const int x0 = (int)(vox - tq * Ru), yy = (int)(tq - zq * Ru), zz = (int)zq;
    o.tg = *reinterpret_cast<const float4*>(a.target + ((long)b * 4 + ol_) * V + vq);
    o.t3 = *reinterpret_cast<const float4*>(a.target + ((long)b * 4 + 3) * V + vq);
    o.tm = a.tokmask[((zz >> 2) * gd + (yy >> 2)) * gd + (x0 >> 2)];
    o.xk = (zz < e0 && yy < e1) ? x0 : (1 << 30);   // x0 + q < e2 fails for a row outside the sample's extent
  };
  // one chunk: x-hat, d0 (bf16), and the four packed operand rows for the reduction GEMMs
  // (round 5: 40 -> 17 vector instructions per channel pair.  The arithmetic is packed fp32 with the expressions and roundings of before ((x - mean) * rstd,
  //  + r, max(y, slope y)); inline asm because the compiler splits most of the packed operations again.  The three mask products no longer cost compares,
  //  selects, multiplies and conversions per channel: [d0 > 0] of a PAIR is read off the packed bf16 d0 by two packed 16-bit integer instructions -- bf16
  //  d0 > 0 <=> its bits as int16 > 0 <=> the saturating 0 - bits is negative -- and the 0xffff / 0 halves AND the bf16 constant 1.0, the packed x-hat and
  //  the pair's two bits of the chunk's mask byte.  LeakyReLU keeps the sign (0 < slope < 1) and the sign of a bf16 rounding is that of the fp32 value.)
  const unsigned sh15 = 0x000f000fu;
  auto chunk = [&](const uint4& xw, const uint4& rw, const tf2 (&nmu)[4], const tf2 (&rs)[4], uint4& ypk, uint4& mpk, uint4& mxpk, uint4& xpk, unsigned& bits) {
    const unsigned xs[4] = {xw.x, xw.y, xw.z, xw.w}, rr[4] = {rw.x, rw.y, rw.z, rw.w};
    unsigned yo[4], mo[4], mxo[4], xo[4];
    unsigned u = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const tf2 xv = {__uint_as_float(xs[i] << 16), __uint_as_float(xs[i] & 0xffff0000u)};
      const tf2 rv = {__uint_as_float(rr[i] << 16), __uint_as_float(rr[i] & 0xffff0000u)};
      tf2 xh, y, ys;
      asm("v_pk_add_f32 %0, %1, %2" : "=v"(xh) : "v"(xv), "v"(nmu[i]));
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(xh) : "v"(xh), "v"(rs[i]));
      asm("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(xh), "v"(rv));
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(ys) : "v"(y), "v"(slope2));
      float ya, yc;   // LeakyReLU, 0 < slope < 1 (asm: behind opaque operands the compiler would canonicalise both inputs of fmaxf first)
      asm("v_max_f32 %0, %1, %2" : "=v"(ya) : "v"(y.x), "v"(ys.x));
      asm("v_max_f32 %0, %1, %2" : "=v"(yc) : "v"(y.y), "v"(ys.y));
      const unsigned yp = pk_bf16(ya, yc), xp = pk_bf16(xh.x, xh.y);
      unsigned nm;
      asm("v_pk_sub_i16 %0, 0, %1 clamp\n\tv_pk_ashrrev_i16 %0, %2, %0" : "=&v"(nm) : "v"(yp), "v"(sh15));   // 0xffff / 0 per half
      yo[i] = yp; xo[i] = xp;
      mo[i] = nm & 0x3f803f80u;   // bf16 1.0 / 0.0
      mxo[i] = nm & xp;           // [d0 > 0] x-hat (the masked-out halves are +0)
      u |= nm & (0x00010001u << (2 * i));
    }
    bits = (u | (u >> 15)) & 0xffu;   // bit 2 i = low half of pair i, bit 2 i + 1 = high half
    ypk = make_uint4(yo[0], yo[1], yo[2], yo[3]); mpk = make_uint4(mo[0], mo[1], mo[2], mo[3]);
    mxpk = make_uint4(mxo[0], mxo[1], mxo[2], mxo[3]); xpk = make_uint4(xo[0], xo[1], xo[2], xo[3]);
  };
  auto put = [&](int row, int cb, const uint4& mpk, const uint4& mxpk, const uint4& xpk, const uint4& ypk) {
    char* p = wl + row * 96 + cb * 2;
    *reinterpret_cast<uint4*>(p) = mpk;
    *reinterpret_cast<uint4*>(p + OPB) = mxpk;
    *reinterpret_cast<uint4*>(p + 2 * OPB) = xpk;
    *reinterpret_cast<uint4*>(p + 3 * OPB) = ypk;
  };
  const long gstep = 4 * 32;
  auto step = [&](long base) {
    const Ops o = nxt;
    if (base + gstep < v1) issue(base + gstep, nxt);
    uint4 y0, y1, y2, mp, mxp, xp;
    unsigned sb0, sb1, sb2;
    chunk(o.x[0], o.r[0], mu0, rs0, y0, mp, mxp, xp, sb0); put(vi, c0, mp, mxp, xp, y0);
    chunk(o.x[1], o.r[1], mu0, rs0, y1, mp, mxp, xp, sb1); put(16 + vi, c0, mp, mxp, xp, y1);
    chunk(o.x[2], o.r[2], mu1, rs1, y2, mp, mxp, xp, sb2); put(16 * t2 + vi, c1, mp, mxp, xp, y2);
    const float4 tg = o.tg, t3 = o.t3;
    const int xr = o.tm != 0u ? o.xk : (1 << 30);
    if (a.sign_mask) {   // [d0 > 0] of the lane's three chunks: byte c0 / 8 of voxels vi and 16 + vi, byte c1 / 8 of voxel 16 t2 + vi -- collected per
      //                    voxel in a wave-private LDS row of 8 bytes (6 used) and stored by lanes 0..31 behind the wave barrier below: 256 contiguous bytes
      //                    per step (single-byte global stores cost the pass 0.17 ms)
      unsigned char* sm = reinterpret_cast<unsigned char*>(smem) + 4 * WLDS + wave * 256;
      sm[vi * 8 + g] = (unsigned char)sb0;
      sm[(16 + vi) * 8 + g] = (unsigned char)sb1;
      sm[(16 * t2 + vi) * 8 + 4 + (g & 1)] = (unsigned char)sb2;
    }
    // head: P[voxel][o] for both tiles
    Frag<bf16_t> f0, f1, f2;
    f0.v = __builtin_bit_cast(bf16x8, y0); f1.v = __builtin_bit_cast(bf16x8, y1); f2.v = __builtin_bit_cast(bf16x8, y2);
    f32x4 P[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    mma(P[0], f0, w0h); mma(P[0], f0, w0l); mma(P[0], f2, wah); mma(P[0], f2, wal);
    mma(P[1], f1, w1h); mma(P[1], f1, w1l); mma(P[1], f2, wbh); mma(P[1], f2, wbl);
    // loss terms and d(pred): lane (o, g) of P[t] holds output o of voxels 4g..4g+3 of tile t (o = vi < 4)
    // (tile 1's product uses weight fragments whose output o sits in column 4 + o, so ONE pass over lanes vi < 8 -- tile vi >> 2, output
    //  vi & 3 -- covers both tiles without moving data between lanes)
    float dq[4] = {0.f, 0.f, 0.f, 0.f};
    if (vi < 8) {
      const int tl = vi >> 2, ol = vi & 3;
      const long vq = base + 16 * tl + 4 * g;
      const float tv[4] = {tg.x, tg.y, tg.z, tg.w}, t3v[4] = {t3.x, t3.y, t3.z, t3.w};
      const bool alpha = ol == 3;
      float pv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // branch-free over the lane's output: RGB terms (o < 3) and the alpha term (o = 3) differ in selects only
        const float p = (tl ? P[1][q] : P[0][q]) + bias_l;
        pv[q] = p;
        const bool on = alpha ? (xr + q < e2) : (t3v[q] > 0.01f);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-p));
        const float val = alpha ? sg : p, df = val - tv[q];
        const float dfac = alpha ? 2.f * sg * (1.f - sg) : 2.f;
        const float l = on ? df * df : 0.f, d = on ? dfac * df : 0.f;
        lsq += l; dsum += d; dq[q] = d;
        cnt += (on && (ol == 0 || alpha)) ? 1.f : 0.f;
        a.dp[((long)b * V + vq + q) * 4 + ol] = d;
      }
      if (a.pred) *reinterpret_cast<float4*>(a.pred + ((long)b * 4 + ol) * V + vq) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
    // reduction GEMMs over the 32 voxels: A = D (k-slot 4t + q <-> voxel 16t + 4g + q = LDS row), B = the operand images
    // rows 0-3 of A: output o over tile 0's voxels (k-slots 0-3), rows 4-7: output o over tile 1's voxels (k-slots 4-7); the two row groups
    // of S are added in the block reduction
    Frag<bf16_t> df;
    {
      const unsigned lo01 = pk_bf16(dq[0], dq[1]), lo23 = pk_bf16(dq[2], dq[3]);
      const bool t1 = (vi & 4) != 0;
      df.v = __builtin_bit_cast(bf16x8, make_uint4(t1 ? 0u : lo01, t1 ? 0u : lo23, t1 ? lo01 : 0u, t1 ? lo23 : 0u));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (a.sign_mask && lane < 32)
      *reinterpret_cast<uint2*>(a.sign_mask + ((long)b * V + base + lane) * 8) = *reinterpret_cast<const uint2*>(smem + 4 * WLDS + wave * 256 + lane * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const char* pa = wl + q * OPB + (4 * g + (vi >> 2)) * 96 + (n * 4 + (vi & 3)) * 8;
        const bf16x4 lo = ds_read_tr16(pa), hi = ds_read_tr16(pa + 16 * 96);
        Frag<bf16_t> bf;
        bf.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        mma(S[q][n], df, bf);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  {
    long base = v0 + wave * 32;
    if (base < v1) issue(base, nxt);
    for (; base < v1; base += gstep) step(base);
  }
  // ---- block reduction: S[q][n][r] of lane (c = vi, g = tile) is that tile's sum for (operand q, output o = r, channel 16n + c) ---------
  __syncthreads();   // every wave is done with its operand images
  float* red = reinterpret_cast<float*>(smem);   // [wave * 2 + tile][q][o][48]
  if (g < 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[(((wave * 2 + g) * 4 + q) * 4 + rr) * C + 16 * n + vi] = S[q][n][rr];
  }
  // loss sums
  {
    float l = vi < 8 ? lsq : 0.f, c = vi < 8 ? cnt : 0.f, d = vi < 8 ? dsum : 0.f;   // lane (vi, g): output vi & 3
    l += __shfl_xor(l, 4, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    c += __shfl_xor(c, 4, 64); c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
    d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 16, 64); d += __shfl_xor(d, 32, 64);
    if (lane < 4) {
      if (lane < 3) atomicAdd(&sacc[0], l); else atomicAdd(&sacc[2], l);
      if (lane == 0) atomicAdd(&sacc[1], c);
      if (lane == 3) atomicAdd(&sacc[3], c);
      atomicAdd(&sacc[4 + lane], d);
    }
  }
  __syncthreads();
  for (int i = tid; i < 8 * C; i += 256) {
    const int k = i / C, c = i - k * C;
    float T[4][4];   // [operand][o]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) {
        float t = 0.f;
        for (int w = 0; w < 8; ++w) t += red[((w * 4 + q) * 4 + oo) * C + c];
        T[q][oo] = t;
      }
    float val;
    if (k < 4) {
      const bool xh = k & 1;                        // k = 0: sum g_rgb, 1: sum g_rgb x-hat, 2: sum g_a, 3: sum g_a x-hat
      val = 0.f;
      const int o_lo = k < 2 ? 0 : 3, o_hi = k < 2 ? 3 : 4;
      for (int oo = o_lo; oo < o_hi; ++oo) {
        const float plain = xh ? T[2][oo] : sacc[4 + oo], masked = xh ? T[1][oo] : T[0][oo];   // sum_v d [* x-hat],  sum_v d [d0>0] [* x-hat]
        val += a.Wout[oo * C + c] * (slope * plain + (1.0f - slope) * masked);
      }
    } else val = T[3][k - 4];
    double* dst = k < 4 ? a.bwd_sums + ((long)b * C + c) * 4 + k : a.bwd_sums + (long)a.B * C * 4 + (long)(k - 4) * C + c;
    atomicAdd(dst, (double)val);
  }
  if (tid < 8) atomicAdd(&a.sums[tid], (double)sacc[tid]);
}
// ---- the same pass with the residual formed from the COARSE tensor (round 6) -------------------------------------------------------------
// The residual of decoder1 is r = ConvT_{k=s=4}(xc) + bt (unetr_block.py:151-158): 3072 outputs of 96 inputs per coarse cell.  Written by
// upconv4_fwd (3.15 GB at 8 x 160^3) it was read back by this pass alone; here the pass forms it on the matrix cores from the 192-byte row of
// the cell.  A 16-voxel MFMA tile needs one weight set, i.e. one phase (dz, dy, dx) of 16 DIFFERENT cells:
//  * workgroup = (range of cells, (dz, dy), sample), 8 waves: waves 0-3 the phases dx = 0, 1 (two tiles of the same 16 consecutive cells per step:
//    32 voxels, neighbours in pairs), waves 4-7 dx = 2, 3 -- the 384-byte run of a cell's fine line is consumed by ONE workgroup (its three 128-byte
//    lines meet in one L2); the phase weights of the line, 4 x (96 x 48) bf16 = 36 KB in fragment order, sit in LDS;
//  * r^T[channel][cell] = W_phase^T xc^T: A = weights (row m of channel block cb <-> channel chan(cb, m)), B = the cell rows as loaded (lane
//    (cell, g): inputs 32 ks + 8 g ..), accumulators start at the bias.  With chan(0, m) = 8 (m >> 2) + (m & 3), chan(1, m) = chan(0, m) + 4,
//    chan(2, m) = 32 + m the lane (cell vi, g) leaves with r of channels 8g .. 8g+7 and 32 + 4g .. 32 + 4g + 3 of ITS voxel in fp32 -- the channels of
//    the 16-byte chunk g and the 8-byte chunk g of the voxel's row of y2, which the lane loads;
//  * head transposed: P^T[o][voxel] = Wout d0^T (A = head weights, rows 0-3: tile 0, rows 4-7: tile 1; B = the packed d0 chunks as they sit; the
//    third fragment carries tile 0 in k-slots 0-3 and tile 1 in 4-7 against weights that are zero on the other tile's slots): lane (vi, g < 2) holds
//    the four outputs of voxel vi of tile g -- own target loads, one 16-byte d(pred) store, one sigmoid per voxel;
//  * d(pred) goes through a 256-byte wave-private LDS image [o][tile][16] to become the A fragment of the reduction GEMMs, which are those of
//    tail_fwd_mfma_kernel (same operand images, row = 16 tile + cell).
// r is no longer rounded to bf16 before the sum; everything downstream (d0's rounding, the head, the sums, the sign mask) as above.
#ifndef TAILC_RS
#define TAILC_RS 112
#endif
template <int DBG>
__global__ __launch_bounds__(512) void tail_fwd_coarse_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats, const bf16_t* __restrict__ xcoarse,
                                                              const bf16_t* __restrict__ Wr, const float* __restrict__ bt, LossArgs a, long V, float slope, int cpb,
                                                              unsigned mg1, unsigned mg2) {
  constexpr int C = 48, RS = TAILC_RS, OPB = 32 * RS, WLDS = 4 * OPB, WBYTES = 36 * 1024, NW = 8;   // RS: row stride of the operand images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float sacc[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), vi = lane & 15, g = lane >> 4;
  const int h = wave >> 2, wq = wave & 3, cls = blockIdx.y, dz = cls >> 2, dy = cls & 3, b = blockIdx.z;
  char* const wl = smem + WBYTES + wave * WLDS;
  unsigned char* const sm = reinterpret_cast<unsigned char*>(smem) + WBYTES + NW * WLDS + wave * 256;
  char* const dT = smem + WBYTES + NW * WLDS + NW * 256 + wave * 256;
  char* const hw = smem + WBYTES + NW * WLDS + 2 * NW * 256 + lane * 16;   // head-weight fragments [6][64 lanes][16 B], written below
  if (tid < 8) sacc[tid] = 0.f;
  {   // the four phase-weight sets of this (dz, dy): [h][t][cb][ks][lane][8]
    const uint4* src = reinterpret_cast<const uint4*>(Wr) + (long)cls * (WBYTES / 16);
    for (int i = tid; i < WBYTES / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[i];
  }
  __syncthreads();
  const char* const wsm = smem + h * (18 * 1024) + lane * 16;
  typedef float tf2 __attribute__((ext_vector_type(2)));
  tf2 mu0[4], rs0[4], mu1[4], rs1[4];   // mu* = MINUS the mean; set 0: channels 8g + j, set 1: channels 32 + 4g + (j & 3)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ca = 8 * g + j, cb_ = 32 + 4 * g + (j & 3);
    mu0[j >> 1][j & 1] = -stats[((long)b * C + ca) * 2]; rs0[j >> 1][j & 1] = stats[((long)b * C + ca) * 2 + 1];
    mu1[j >> 1][j & 1] = -stats[((long)b * C + cb_) * 2]; rs1[j >> 1][j & 1] = stats[((long)b * C + cb_) * 2 + 1];
  }
  const tf2 slope2 = {slope, slope};
  // head weights as A fragments (lane (m = vi, g), k-slot j), bf16 hi + lo parts: kept in LDS (24 registers), six 16-byte reads per step
  if (wave == 0) {
    Frag<bf16_t> A0h, A0l, A1h, A1l, A2h, A2l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ow = vi & 3;
      const float wa = vi < 8 ? a.Wout[ow * C + 8 * g + j] : 0.f, wb = vi < 8 ? a.Wout[ow * C + 32 + 4 * g + (j & 3)] : 0.f;
      const bf16_t ah = f2bf(wa), bh = f2bf(wb);
      const bf16_t al = f2bf(wa - bf2f(ah)), bl = f2bf(wb - bf2f(bh));
      const bool t0 = vi < 4, t1 = vi >= 4 && vi < 8;
      A0h.v[j] = t0 ? (short)ah : (short)0; A0l.v[j] = t0 ? (short)al : (short)0;
      A1h.v[j] = t1 ? (short)ah : (short)0; A1l.v[j] = t1 ? (short)al : (short)0;
      const bool on2 = (t0 && j < 4) || (t1 && j >= 4);
      A2h.v[j] = on2 ? (short)bh : (short)0; A2l.v[j] = on2 ? (short)bl : (short)0;
    }
    *reinterpret_cast<bf16x8*>(hw) = A0h.v; *reinterpret_cast<bf16x8*>(hw + 1024) = A0l.v; *reinterpret_cast<bf16x8*>(hw + 2048) = A1h.v;
    *reinterpret_cast<bf16x8*>(hw + 3072) = A1l.v; *reinterpret_cast<bf16x8*>(hw + 4096) = A2h.v; *reinterpret_cast<bf16x8*>(hw + 5120) = A2l.v;
  }
  __syncthreads();
  f32x4 bb[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) { bb[0][i] = bt[8 * g + i]; bb[1][i] = bt[8 * g + 4 + i]; bb[2][i] = bt[32 + 4 * g + i]; }
  const float bo0 = a.bout[0], bo1 = a.bout[1], bo2 = a.bout[2], bo3 = a.bout[3];
  const int e0 = a.extents[b * 3], e1 = a.extents[b * 3 + 1], e2 = a.extents[b * 3 + 2];
  const unsigned Ru = (unsigned)a.R, gd = (unsigned)(a.R >> 2), gd2 = gd * gd, ncell = gd2 * gd;
  f32x4 S[4][3];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int n = 0; n < 3; ++n) S[q][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float l_rgb = 0.f, l_a = 0.f, n_rgb = 0.f, n_a = 0.f, ds0 = 0.f, ds1 = 0.f, ds2 = 0.f, ds3 = 0.f;
  const char* const xs_ = reinterpret_cast<const char*>(x + (long)b * V * C);
  const char* const cs_ = reinterpret_cast<const char*>(xcoarse + (long)b * ncell * 96);
  const int tl = g & 1;   // the tile of the lane's loss part (lanes g >= 2 idle there)
  // What a step reads from global memory, in three groups of two register sets each (step k uses set k & 1; no copies between sets, the loop is unrolled by two):
  //   B: the cell rows (L2) and the token mask, requested at the start of step k - 1;  X: the rows of y2 (HBM), requested in step k - 2 right behind its own
  //   chunk arithmetic;  T: the targets (HBM), requested in step k - 2 behind its loss part -- the HBM streams are a step and a half ahead
  //   (round 6: with everything one step ahead the pass moved 2.3 TB/s -- fewer bytes in flight per wave than the pass that also read r, at the same latency)
  struct OpsB { uint4 xc[3]; unsigned tm, vox; int ins; };
  struct OpsX { uint4 xa[2]; uint2 xb[2]; unsigned v0; int ins; };   // v0 / ins: voxel index and extent test of the cells requested last (passed on to that step's B set)
  struct OpsT { float tg[4]; };
  OpsB Bq[2]; OpsX Xq[2]; OpsT Tq[2];
  auto issueB = [&](unsigned c0, unsigned v0, int ins, OpsB& o) {   // (v0, ins): as computed when the step's y2 rows were requested
    const unsigned cell = c0 + (unsigned)vi;
    o.vox = v0; o.ins = ins;
    const char* const pc = cs_ + (unsigned long)cell * 192u + 16 * g;
    o.xc[0] = *reinterpret_cast<const uint4*>(pc); o.xc[1] = *reinterpret_cast<const uint4*>(pc + 64); o.xc[2] = *reinterpret_cast<const uint4*>(pc + 128);
    o.tm = a.tokmask[cell];
  };
  auto issueX = [&](unsigned c0, OpsX& o) {
    const unsigned cell = c0 + (unsigned)vi;
    const unsigned zc = __umulhi(cell, mg2), rem = cell - zc * gd2, yc = __umulhi(rem, mg1), xc_ = rem - yc * gd;
    const unsigned zf = 4u * zc + (unsigned)dz, yf = 4u * yc + (unsigned)dy, xf = 4u * xc_ + 2u * (unsigned)h;
    const unsigned v0 = (zf * Ru + yf) * Ru + xf;
    o.v0 = v0;
    o.ins = ((int)zf < e0 && (int)yf < e1 && (int)xf + tl < e2) ? 1 : 0;
    const char* const p = xs_ + (unsigned long)v0 * 96u;
    o.xa[0] = *reinterpret_cast<const uint4*>(p + 16 * g); o.xa[1] = *reinterpret_cast<const uint4*>(p + 96 + 16 * g);
    o.xb[0] = *reinterpret_cast<const uint2*>(p + 64 + 8 * g); o.xb[1] = *reinterpret_cast<const uint2*>(p + 160 + 8 * g);
  };
  auto issueT = [&](unsigned v0, OpsT& o) {
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) o.tg[oo] = a.target[((long)b * 4 + oo) * V + v0 + tl];   // (lanes g >= 2 repeat the addresses of g - 2: no branch around a load)
  };
  const unsigned sh15 = 0x000f000fu;
  auto chunk = [&](const uint4& xw, const f32x4& ra, const f32x4& rb, const tf2 (&nmu)[4], const tf2 (&rs)[4], uint4& ypk, uint4& mpk, uint4& mxpk, uint4& xpk, unsigned& bits) {
    const unsigned xs[4] = {xw.x, xw.y, xw.z, xw.w};
    const tf2 rr[4] = {tf2{ra[0], ra[1]}, tf2{ra[2], ra[3]}, tf2{rb[0], rb[1]}, tf2{rb[2], rb[3]}};
    unsigned yo[4], mo[4], mxo[4], xo[4];
    unsigned u = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const tf2 xv = {__uint_as_float(xs[i] << 16), __uint_as_float(xs[i] & 0xffff0000u)};
      tf2 xh, y, ys;
      asm("v_pk_add_f32 %0, %1, %2" : "=v"(xh) : "v"(xv), "v"(nmu[i]));
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(xh) : "v"(xh), "v"(rs[i]));
      y = xh + rr[i];   // NOT inline asm: rr comes straight out of an MFMA, and the compiler places the wait states of a matrix-core result only in front of
      //                   instructions it knows (an asm v_pk_add_f32 right behind the last MFMA of the residual read the accumulator too early: wrong values)
      asm("v_pk_mul_f32 %0, %1, %2" : "=v"(ys) : "v"(y), "v"(slope2));
      float ya, yc;
      asm("v_max_f32 %0, %1, %2" : "=v"(ya) : "v"(y.x), "v"(ys.x));
      asm("v_max_f32 %0, %1, %2" : "=v"(yc) : "v"(y.y), "v"(ys.y));
      const unsigned yp = pk_bf16(ya, yc), xp = pk_bf16(xh.x, xh.y);
      unsigned nm;
      asm("v_pk_sub_i16 %0, 0, %1 clamp\n\tv_pk_ashrrev_i16 %0, %2, %0" : "=&v"(nm) : "v"(yp), "v"(sh15));   // 0xffff / 0 per half
      yo[i] = yp; xo[i] = xp;
      mo[i] = nm & 0x3f803f80u;
      mxo[i] = nm & xp;
      u |= nm & (0x00010001u << (2 * i));
    }
    bits = (u | (u >> 15)) & 0xffu;
    ypk = make_uint4(yo[0], yo[1], yo[2], yo[3]); mpk = make_uint4(mo[0], mo[1], mo[2], mo[3]);
    mxpk = make_uint4(mxo[0], mxo[1], mxo[2], mxo[3]); xpk = make_uint4(xo[0], xo[1], xo[2], xo[3]);
  };
  auto put16 = [&](int row, const uint4& mpk, const uint4& mxpk, const uint4& xpk, const uint4& ypk) {
    char* p = wl + row * RS + 16 * g;
    *reinterpret_cast<uint4*>(p) = mpk;
    *reinterpret_cast<uint4*>(p + OPB) = mxpk;
    *reinterpret_cast<uint4*>(p + 2 * OPB) = xpk;
    *reinterpret_cast<uint4*>(p + 3 * OPB) = ypk;
  };
  auto put8 = [&](const uint4& mpk, const uint4& mxpk, const uint4& xpk, const uint4& ypk) {   // tile 0: row vi <- .xy, tile 1: row 16 + vi <- .zw
    char* p = wl + vi * RS + 64 + 8 * g;
    *reinterpret_cast<uint2*>(p) = make_uint2(mpk.x, mpk.y); *reinterpret_cast<uint2*>(p + 16 * RS) = make_uint2(mpk.z, mpk.w);
    *reinterpret_cast<uint2*>(p + OPB) = make_uint2(mxpk.x, mxpk.y); *reinterpret_cast<uint2*>(p + OPB + 16 * RS) = make_uint2(mxpk.z, mxpk.w);
    *reinterpret_cast<uint2*>(p + 2 * OPB) = make_uint2(xpk.x, xpk.y); *reinterpret_cast<uint2*>(p + 2 * OPB + 16 * RS) = make_uint2(xpk.z, xpk.w);
    *reinterpret_cast<uint2*>(p + 3 * OPB) = make_uint2(ypk.x, ypk.y); *reinterpret_cast<uint2*>(p + 3 * OPB + 16 * RS) = make_uint2(ypk.z, ypk.w);
  };
  const unsigned cbeg = blockIdx.x * (unsigned)cpb, cend = cbeg + (unsigned)cpb < ncell ? cbeg + (unsigned)cpb : ncell;
  auto step = [&](unsigned c0, auto parity) {
    constexpr int PS = decltype(parity)::value;
    const OpsB& o = Bq[PS];
    OpsX& ox = Xq[PS];
    OpsT& ot = Tq[PS];
    // (requests past the wave's last step re-read the current one's addresses: no branch around a load -- with one the compiler can no longer count the loads in
    //  flight and waits for all but the newest in front of the first use of this step's operands)
    issueB(c0 + 64u < cend ? c0 + 64u : c0, Xq[PS ^ 1].v0, Xq[PS ^ 1].ins, Bq[PS ^ 1]);   // (the other X set still holds the next step's cells)
    // the residual of both tiles: r[t][cb] = lane (cell vi, g): channels chan(cb, 4g + i)
    Frag<bf16_t> xcf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) xcf[ks].v = __builtin_bit_cast(bf16x8, o.xc[ks]);
    f32x4 r[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        r[t][cb] = bb[cb];
        if (!(DBG & 1))
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) {
            Frag<bf16_t> wf;
            wf.v = *reinterpret_cast<const bf16x8*>(wsm + ((t * 3 + cb) * 3 + ks) * 1024);
            mma(r[t][cb], wf, xcf[ks]);
          }
        else r[t][cb][0] += __uint_as_float(o.xc[cb].x & 0x3f800000u);
      }
    uint4 y0, y1, y2, mp, mxp, xp;
    unsigned sb0, sb1, sb2;
    chunk(ox.xa[0], r[0][0], r[0][1], mu0, rs0, y0, mp, mxp, xp, sb0); put16(vi, mp, mxp, xp, y0);
    chunk(ox.xa[1], r[1][0], r[1][1], mu0, rs0, y1, mp, mxp, xp, sb1); put16(16 + vi, mp, mxp, xp, y1);
    chunk(make_uint4(ox.xb[0].x, ox.xb[0].y, ox.xb[1].x, ox.xb[1].y), r[0][2], r[1][2], mu1, rs1, y2, mp, mxp, xp, sb2); put8(mp, mxp, xp, y2);
    issueX(c0 + 128u < cend ? c0 + 128u : c0, ox);   // (the set is free again: y2 of the step after next)
    {   // [d0 > 0]: byte g of both voxels; the lane's nibbles of byte 4 + (g >> 1) meet those of lane g ^ 1
      sm[vi * 8 + g] = (unsigned char)sb0;
      sm[(16 + vi) * 8 + g] = (unsigned char)sb1;
      const unsigned pn = (unsigned)__shfl_xor((int)sb2, 16, 64);
      if (!(g & 1)) {
        sm[vi * 8 + 4 + (g >> 1)] = (unsigned char)((sb2 & 15u) | ((pn & 15u) << 4));
        sm[(16 + vi) * 8 + 4 + (g >> 1)] = (unsigned char)((sb2 >> 4) | ((pn >> 4) << 4));
      }
    }
    // head: P^T[o][voxel], rows 0-3 tile 0, rows 4-7 tile 1
    Frag<bf16_t> f0, f1, f2;
    f0.v = __builtin_bit_cast(bf16x8, y0); f1.v = __builtin_bit_cast(bf16x8, y1); f2.v = __builtin_bit_cast(bf16x8, y2);
    f32x4 P = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      Frag<bf16_t> hf[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) hf[i].v = *reinterpret_cast<const bf16x8*>(hw + i * 1024);
      mma(P, hf[0], f0); mma(P, hf[1], f0); mma(P, hf[2], f1); mma(P, hf[3], f1); mma(P, hf[4], f2); mma(P, hf[5], f2);
    }
    if (g < 2) {   // lane (vi, g): the four outputs of voxel o.vox + g
      const long vq = (long)o.vox + g;
      const float p0 = P[0] + bo0, p1 = P[1] + bo1, p2 = P[2] + bo2, p3 = P[3] + bo3;
      const bool occ = ot.tg[3] > 0.01f, rm = o.tm != 0u && o.ins != 0;
      const float d0f = p0 - ot.tg[0], d1f = p1 - ot.tg[1], d2f = p2 - ot.tg[2];
      const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-p3)), d3f = sg - ot.tg[3];
      const float q0 = occ ? 2.f * d0f : 0.f, q1 = occ ? 2.f * d1f : 0.f, q2 = occ ? 2.f * d2f : 0.f, q3 = rm ? 2.f * sg * (1.f - sg) * d3f : 0.f;
      l_rgb += occ ? d0f * d0f + d1f * d1f + d2f * d2f : 0.f; l_a += rm ? d3f * d3f : 0.f;
      n_rgb += occ ? 1.f : 0.f; n_a += rm ? 1.f : 0.f;
      ds0 += q0; ds1 += q1; ds2 += q2; ds3 += q3;
      if (!(DBG & 4) || q0 == 12345.f) *reinterpret_cast<float4*>(a.dp + ((long)b * V + vq) * 4) = make_float4(q0, q1, q2, q3);
      if (a.pred) { a.pred[((long)b * 4 + 0) * V + vq] = p0; a.pred[((long)b * 4 + 1) * V + vq] = p1; a.pred[((long)b * 4 + 2) * V + vq] = p2; a.pred[((long)b * 4 + 3) * V + vq] = p3; }
      bf16_t* const dt_ = reinterpret_cast<bf16_t*>(dT) + g * 16 + vi;   // [o][tile][16]
      const unsigned q01 = pk_bf16(q0, q1), q23 = pk_bf16(q2, q3);
      dt_[0] = (bf16_t)(q01 & 0xffffu); dt_[32] = (bf16_t)(q01 >> 16); dt_[64] = (bf16_t)(q23 & 0xffffu); dt_[96] = (bf16_t)(q23 >> 16);
    }
    issueT(ox.v0, ot);   // (ox.v0: already that of the step after next)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 32 && (!(DBG & 4) || sb0 == 0x12345u))   // lane = 16 tile + cell
      *reinterpret_cast<uint2*>(a.sign_mask + ((long)b * V + o.vox + g) * 8) = *reinterpret_cast<const uint2*>(sm + lane * 8);
    Frag<bf16_t> df;
    {
      uint2 w = make_uint2(0u, 0u);
      if (vi < 8) w = *reinterpret_cast<const uint2*>(dT + (((vi & 3) * 2 + (vi >> 2)) * 16 + 4 * g) * 2);
      const bool t1 = (vi & 4) != 0;
      df.v = __builtin_bit_cast(bf16x8, make_uint4(t1 ? 0u : w.x, t1 ? 0u : w.y, t1 ? w.x : 0u, t1 ? w.y : 0u));
    }
    if (!(DBG & 2))
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const char* pa = wl + q * OPB + (4 * g + (vi >> 2)) * RS + (n * 4 + (vi & 3)) * 8;
        const bf16x4 lo = ds_read_tr16(pa), hi = ds_read_tr16(pa + 16 * RS);
        Frag<bf16_t> bf;
        bf.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        mma(S[q][n], df, bf);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  {
    unsigned c0 = cbeg + 16u * (unsigned)wq;
    if (c0 < cend) {
      issueX(c0, Xq[0]); issueB(c0, Xq[0].v0, Xq[0].ins, Bq[0]); issueT(Xq[0].v0, Tq[0]);
      issueX(c0 + 64u < cend ? c0 + 64u : c0, Xq[1]); issueT(Xq[1].v0, Tq[1]);
      for (; c0 + 64u < cend; c0 += 128u) {   // two steps per trip: straight-line code between the requests and their uses
        step(c0, std::integral_constant<int, 0>{});
        step(c0 + 64u, std::integral_constant<int, 1>{});
      }
      if (c0 < cend) step(c0, std::integral_constant<int, 0>{});
    }
  }
  // ---- block reduction (as tail_fwd_mfma_kernel, 8 waves) ----------------------------------------------------------------------------------
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);   // [wave * 2 + tile][q][o][48]
  if (g < 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) red[(((wave * 2 + g) * 4 + q) * 4 + rr) * C + 16 * n + vi] = S[q][n][rr];
  }
  {
    const float v8[8] = {l_rgb, n_rgb, l_a, n_a, ds0, ds1, ds2, ds3};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s = wave_sum(v8[i]);
      if (lane == 0) atomicAdd(&sacc[i], s);
    }
  }
  __syncthreads();
  for (int i = tid; i < 8 * C; i += 512) {
    const int k = i / C, c = i - k * C;
    float T[4][4];   // [operand][o]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int oo = 0; oo < 4; ++oo) {
        float t = 0.f;
        for (int w = 0; w < 2 * NW; ++w) t += red[((w * 4 + q) * 4 + oo) * C + c];
        T[q][oo] = t;
      }
    float val;
    if (k < 4) {
      const bool xh = k & 1;
      val = 0.f;
      const int o_lo = k < 2 ? 0 : 3, o_hi = k < 2 ? 3 : 4;
      for (int oo = o_lo; oo < o_hi; ++oo) {
        const float plain = xh ? T[2][oo] : sacc[4 + oo], masked = xh ? T[1][oo] : T[0][oo];
        val += a.Wout[oo * C + c] * (slope * plain + (1.0f - slope) * masked);
      }
    } else val = T[3][k - 4];
    double* dst = k < 4 ? a.bwd_sums + ((long)b * C + c) * 4 + k : a.bwd_sums + (long)a.B * C * 4 + (long)(k - 4) * C + c;
    atomicAdd(dst, (double)val);
  }
  if (tid < 8) atomicAdd(&a.sums[tid], (double)sacc[tid]);
}
// Wr bf16 [16 (dz, dy)][2 h][2 t][3 cb][3 ks][64 lanes][8] from WtT fp32 [64 phases][96][48] (the workspace of cconv_pack): A fragments of the phase dx = 2h + t
__global__ __launch_bounds__(512) void tail_r_pack_kernel(const float* __restrict__ WtT, bf16_t* __restrict__ Wr) {
  const int blk = blockIdx.x, tid = threadIdx.x;   // one workgroup per fragment
  const int cls = blk / 36, f = blk - cls * 36, ht = f / 9, cb = (f - ht * 9) / 3, ks = f - ht * 9 - cb * 3;
  const int lane = tid >> 3, j = tid & 7, li = lane & 15, kg = lane >> 4;
  const int ph = cls * 4 + ht, ci = 32 * ks + 8 * kg + j;
  const int co = cb == 2 ? 32 + li : 8 * (li >> 2) + (li & 3) + 4 * cb;
  Wr[((long)blk * 64 + lane) * 8 + j] = f2bf(WtT[((long)ph * 96 + ci) * 48 + co]);
}
long k_tail_r_pack_numel() { return 16L * 36 * 512; }
int k_tail_r_pack(const float* ws, void* Wr, hipStream_t st) {
  hipLaunchKernelGGL(tail_r_pack_kernel, dim3(16 * 36), dim3(512), 0, st, ws, (bf16_t*)Wr);
  NMH_CHECK_LAUNCH();
  return 0;
}
// training pass only (d(pred), the backward sums and the sign mask are all written): bf16, 48 channels, R a multiple of 4 with (R/4)^3 a multiple of 16, R <= 256
int k_tail_fwd_coarse(const LossArgs& a, const void* x, const float* stats, const void* xcoarse, const void* Wr, const float* bt, float slope, hipStream_t st) {
  const long V = (long)a.R * a.R * a.R;
  const int gd = a.R / 4;
  if (a.dt != NMH_DT_BF16 || a.Cd != 48 || a.R % 4 || a.R > 256 || ((long)gd * gd * gd) % 16 || !a.dp || !a.bwd_sums || !a.sign_mask || !(slope > 0.f && slope < 1.f)) return -4;
  hipError_t e = nmh_zero_async(a.sums, 8 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  e = nmh_zero_async(a.bwd_sums, sizeof(double) * ((size_t)a.B * 48 * 4 + 4 * 48), st);
  if (e != hipSuccess) return (int)e;
  constexpr int LDS = 36 * 1024 + 8 * 4 * 32 * TAILC_RS + 8 * 256 + 8 * 256 + 6 * 1024;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    e = hipFuncSetAttribute((const void*)tail_fwd_coarse_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_fwd_coarse_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_fwd_coarse_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_fwd_coarse_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_fwd_coarse_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  static const int dbg = getenv("NMH_TAILC_DBG") ? atoi(getenv("NMH_TAILC_DBG")) : 0;   // timing experiments only (wrong results): 1 no residual MFMAs, 2 no reduction GEMMs, 4 no stores
  static const int cpb_env = getenv("NMH_TAILC_CPB") ? atoi(getenv("NMH_TAILC_CPB")) : 0;
  const int ncell = gd * gd * gd;
  int cpb = cpb_env > 0 ? (cpb_env + 63) / 64 * 64 : 4096;   // (8 x 160^3 inside the step: 42.38 ms with 2048, 42.20 with 4096, 42.87 with 1024)
  if (cpb > (ncell + 63) / 64 * 64) cpb = (ncell + 63) / 64 * 64;
  const unsigned mg1 = (unsigned)((0x100000000ULL + (unsigned)gd - 1) / (unsigned)gd), mg2 = (unsigned)((0x100000000ULL + (unsigned)(gd * gd) - 1) / (unsigned)(gd * gd));
  dim3 grid((unsigned)((ncell + cpb - 1) / cpb), 16, a.B);
#define TC_LAUNCH(D) hipLaunchKernelGGL(tail_fwd_coarse_kernel<D>, grid, dim3(512), LDS, st, (const bf16_t*)x, stats, (const bf16_t*)xcoarse, (const bf16_t*)Wr, bt, a, V, slope, cpb, mg1, mg2)
  if (dbg == 1) TC_LAUNCH(1); else if (dbg == 2) TC_LAUNCH(2); else if (dbg == 4) TC_LAUNCH(4); else if (dbg == 7) TC_LAUNCH(7); else TC_LAUNCH(0);
#undef TC_LAUNCH
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_tail_fwd(const LossArgs& a, const void* x, const float* stats, const void* r, void* out, float slope, hipStream_t st) {
  const int C = a.Cd;
  if (C % 8 || C > 512) return -2;
  const long V = (long)a.R * a.R * a.R;
  hipError_t e = nmh_zero_async(a.sums, (a.dp ? 8 : 4) * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  long vpb = (V * a.B + 2047) / 2048;
  if (vpb < 160) vpb = 160;
  if (vpb > (a.bwd_sums ? 4096 : 2048)) vpb = a.bwd_sums ? 4096 : 2048;   // (matrix-core pass at 8 x 160^3: 1.89 / 1.58 / 1.54 / 1.55 ms with 1024 / 2048 / 4096 / 8192)
  static const int tf_vpb = getenv("NMH_TAIL_FWD_VPB") ? atoi(getenv("NMH_TAIL_FWD_VPB")) : 0;
  if (tf_vpb > 0 && a.bwd_sums) vpb = tf_vpb;
  dim3 grid((unsigned)((V + vpb - 1) / vpb), a.B);
  if (a.bwd_sums) {
    if (!a.dp) return -4;   // the fused backward sums are built from d(pred)
    e = nmh_zero_async(a.bwd_sums, sizeof(double) * ((size_t)a.B * C * 4 + 4 * C), st);
    if (e != hipSuccess) return (int)e;
    const int use_mfma = getenv("NMH_TAIL_MFMA") ? atoi(getenv("NMH_TAIL_MFMA")) : 1;
    if (a.sign_mask && !(use_mfma && a.dt == NMH_DT_BF16 && C == 48 && !out && a.R % 4 == 0 && V < (1L << 31) && slope > 0.f && slope < 1.f)) return -4;
    if (use_mfma && a.dt == NMH_DT_BF16 && C == 48 && !out && a.R % 4 == 0 && V < (1L << 31) && slope > 0.f && slope < 1.f) {
      const long vpm = (vpb + 127) / 128 * 128;   // whole 32-voxel steps per wave
      dim3 gm((unsigned)((V + vpm - 1) / vpm), a.B);
      static const int tnt = getenv("NMH_TAIL_FWD_NT") ? atoi(getenv("NMH_TAIL_FWD_NT")) : 0;   // measured inside the step: 49.78 ms with, 49.51 without (three alternating runs each)
      if (tnt && in_apply_streaming(V, a.B, C, a.dt)) hipLaunchKernelGGL(tail_fwd_mfma_kernel<true>, gm, dim3(256), 4 * 4 * 32 * 96 + 4 * 256, st, (const bf16_t*)x, stats, (const bf16_t*)r, a, V, slope, vpm);
      else hipLaunchKernelGGL(tail_fwd_mfma_kernel<false>, gm, dim3(256), 4 * 4 * 32 * 96 + 4 * 256, st, (const bf16_t*)x, stats, (const bf16_t*)r, a, V, slope, vpm);
      NMH_CHECK_LAUNCH();
      return 0;
    }
    const size_t lds = 64 * 256 * sizeof(float);
    static NmhPerDeviceOnce attr_set;
    if (attr_set.need()) {
      e = hipFuncSetAttribute((const void*)tail_fwd_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tail_fwd_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_set.set();
    }
    if (a.dt == NMH_DT_BF16) hipLaunchKernelGGL((tail_fwd_kernel<bf16_t, true>), grid, dim3(256), lds, st, (const bf16_t*)x, stats, (const bf16_t*)r, (bf16_t*)out, a, V, C, slope, vpb);
    else hipLaunchKernelGGL((tail_fwd_kernel<float, true>), grid, dim3(256), lds, st, (const float*)x, stats, (const float*)r, (float*)out, a, V, C, slope, vpb);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  if (a.dt == NMH_DT_BF16) hipLaunchKernelGGL((tail_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)x, stats, (const bf16_t*)r, (bf16_t*)out, a, V, C, slope, vpb);
  else hipLaunchKernelGGL((tail_fwd_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, stats, (const float*)r, (float*)out, a, V, C, slope, vpb);
  NMH_CHECK_LAUNCH();
  return 0;
}
