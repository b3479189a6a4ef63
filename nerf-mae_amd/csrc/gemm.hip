// MFMA GEMM engine for the NeRF-MAE hot path (gfx950).
//
//  gemm_nt : C[M,N] = epi( A[M,K] . B[N,K]^T )       A rows either direct or the implicit-GEMM view of a
//            3x3x3 convolution over a channels-last (B,D,H,W,Cin) volume (K = 27*Cin, zero padding).
//            Used for every Linear forward/dgrad (SURVEY O1,O4,O5,O6,O8), the 1x1 convs and the
//            3x3x3 decoder convs forward + dgrad (O10; dgrad = same kernel, flipped/transposed pack).
//  gemm_tn : dW[N,K] += sum_m A[m,N] . B[m,K]         weight gradients (contraction-major operands staged
//            as loaded, fragments via ds_read_b64_tr_b16), split over m with fp32 atomics; B rows direct
//            or the conv-tap gather; output index remapped to PyTorch parameter layouts.
//
// Both are templated on the storage type: bf16 (v_mfma_f32_16x16x32_bf16) and f32 (exact
// v_mfma_f32_16x16x4_f32) share all indexing, so the f32 build is the 1e-3-parity mode of the same code.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>
static inline bool lda_ok(long lda, long ldb) { return lda % 8 == 0 && ldb % 8 == 0; }

// ------------------------------------------------------------------------------------------------
// fast unsigned division by a runtime constant (n < 2^31), host-built
// ------------------------------------------------------------------------------------------------
FDiv make_fdiv(unsigned d) {
  FDiv f;
  f.d = d;
  if (d <= 1) { f.M = 0; f.sh = -1; return f; }
  int cl = 0;
  while ((1u << cl) < d) ++cl;
  unsigned long long p2 = 1ull << (31 + cl);
  f.M = (unsigned)((p2 + d - 1) / d);
  f.sh = cl - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FDiv& f) { return f.sh < 0 ? n : (__umulhi(n, f.M) >> f.sh); }

// ------------------------------------------------------------------------------------------------
// A-row providers for gemm_nt.  Row m of the GEMM = (sample b = blockIdx.z, local row).
// ------------------------------------------------------------------------------------------------
template <typename T> struct ADirect {
  const T* A; long lda; long rows_per_z;  // rows_per_z = M when gridDim.z == 1
  struct Row { const T* p; };
  struct Kst { int k; bool ok; };
  __device__ __forceinline__ void init_row(Row& r, int m, int M, int zb) const {
    r.p = (m < M) ? A + ((long)zb * rows_per_z + m) * lda : nullptr;
  }
  __device__ __forceinline__ Kst init_k(int k, int K) const { return Kst{k, k < K}; }
  __device__ __forceinline__ uint4 load(const Row& r, const Kst& ks) const {
    if (r.p == nullptr || !ks.ok) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(r.p + ks.k);
  }
};

// pixel-shuffled view of a fine-grid tensor: A[m][tap*Cout + co] = X[fine(m, tap)][co] (ConvTranspose3d k = stride, input gradient)
template <typename T> struct AUp {
  const T* X; long ldc; int v, k, Cout; FDiv dv, dk, dc;
  struct Row { long fine0; bool ok; };
  struct Kst { long off; bool ok; };
  __device__ __forceinline__ void init_row(Row& r, int m, int M, int /*zb*/) const {
    r.ok = m < M;
    unsigned q = fdiv((unsigned)m, dv), x = m - q * v;
    unsigned q2 = fdiv(q, dv), y = q - q2 * v;
    unsigned b = fdiv(q2, dv), z = q2 - b * v;
    const long V = (long)v * k;
    r.fine0 = (((long)b * V + z * k) * V + y * k) * V + x * k;
  }
  __device__ __forceinline__ Kst init_k(int kk, int K) const {
    Kst s;
    s.ok = kk < K;
    unsigned tap = fdiv((unsigned)kk, dc);
    int co = kk - (int)tap * Cout;
    unsigned tq = fdiv(tap, dk), tx = tap - tq * k, tz = fdiv(tq, dk), ty = tq - tz * k;
    const long V = (long)v * k;
    s.off = (((long)tz * V + ty) * V + tx) * ldc + co;
    return s;
  }
  __device__ __forceinline__ uint4 load(const Row& r, const Kst& s) const {
    if (!(r.ok && s.ok)) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(X + r.fine0 * ldc + s.off);
  }
};

// implicit-GEMM view of conv3d k=3 pad=1 over channels-last X[(b*D+z)*H+y)*W+x][Cin]; k = tap*Cin + ci,
// tap = (dz+1)*9 + (dy+1)*3 + (dx+1)
template <typename T> struct AConv3 {
  const T* X; int Cin, D, H, W; FDiv dW, dH, dC;
  struct Row { long vox; int z, y, x; bool ok; };
  struct Kst { int dz, dy, dx, ci; long off; bool ok; };
  __device__ __forceinline__ void init_row(Row& r, int m, int M, int zb) const {
    r.ok = m < M;
    unsigned q = fdiv((unsigned)m, dW);
    r.x = m - q * W;
    unsigned q2 = fdiv(q, dH);
    r.y = q - q2 * H;
    r.z = q2;
    r.vox = (long)zb * ((long)D * H * W) + m;
  }
  __device__ __forceinline__ Kst init_k(int k, int K) const {
    Kst s;
    s.ok = k < K;
    int tap = (int)fdiv((unsigned)k, dC);
    s.ci = k - tap * Cin;
    int t9 = tap / 9, r9 = tap - t9 * 9, t3 = r9 / 3;
    s.dz = t9 - 1; s.dy = t3 - 1; s.dx = r9 - t3 * 3 - 1;
    s.off = ((long)s.dz * H + s.dy) * W + s.dx;
    return s;
  }
  __device__ __forceinline__ uint4 load(const Row& r, const Kst& s) const {
    int z = r.z + s.dz, y = r.y + s.dy, x = r.x + s.dx;
    bool ok = r.ok && s.ok && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    if (!ok) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(X + (r.vox + s.off) * Cin + s.ci);
  }
};

// The same operand for bf16 with Cin a multiple of 64 (every decoder level): a 64-deep k-tile then lies inside ONE tap, so the tap, its voxel offset and the
// channel offset of the tile are wave-uniform (scalar unit: they go into the buffer descriptor's base), and what is left per lane is fixed for the whole
// M tile -- the byte offset of (row voxel, the lane's 8-channel piece) and a 27-bit mask of the taps that stay inside the volume for that row.  A load costs
// a bit test, a select and the buffer load (an out-of-range offset reads zero: the conv's padding), where AConv3 spends ~20 vector instructions per
// 16-byte load on the tap decode, three range tests and 64-bit address arithmetic -- as many issue cycles per k-tile as its 24 MFMAs (the 10^3 / 20^3 decoder
// convs ran at 0.19-0.25 of the MFMA peak).  Sample tensors must stay below 2 GiB (checked by the launcher).
template <typename T> struct AConv3F {
  const T* X; int Cin, D, H, W; FDiv dW, dH, dC;
  struct Row { unsigned voff, mask; };
  struct Kst { const T* base; int tap; };
  __device__ __forceinline__ void init_row(Row& r, int m, int M, int zb) const {
    const bool ok = m < M;
    const unsigned q = fdiv((unsigned)m, dW);
    const int x = m - q * W;
    const unsigned q2 = fdiv(q, dH);
    const int y = q - q2 * H, z = (int)q2;
    r.voff = (unsigned)((((long)zb * M + m) * Cin + (threadIdx.x & 7) * 8) * (long)sizeof(T));
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dz = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
      const bool v = ok && (unsigned)(z + dz) < (unsigned)D && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
      mk |= v ? (1u << t) : 0u;
    }
    r.mask = mk;
  }
  __device__ __forceinline__ Kst init_k(int k, int K) const {
    const int kb = __builtin_amdgcn_readfirstlane(k & ~63);   // the k-tile's first column: the same in every lane (k = kb + 8 (tid & 7))
    Kst s;
    int tap = (int)fdiv((unsigned)kb, dC);
    const int ci = kb - tap * Cin;
    const int t9 = tap / 9, r9 = tap - t9 * 9, t3 = r9 / 3;
    const long off = ((long)(t9 - 1) * H + (t3 - 1)) * W + (r9 - t3 * 3 - 1);
    s.base = X + off * Cin + ci;
    s.tap = kb < K ? tap : 27;   // (bit 27 of a row mask is never set)
    return s;
  }
  __device__ __forceinline__ uint4 load(const Row& r, const Kst& s) const {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)s.base, 0, 0x7fffffff, 0x00020000);
    const unsigned vo = ((r.mask >> s.tap) & 1u) ? r.voff : 0x80000000u;
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
  }
};

// ------------------------------------------------------------------------------------------------
// epilogue (runtime-flagged, wave-uniform branches); operates on 8 consecutive columns
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void epilogue8(const EpiParams& ep, long grow, int gcol, float (&v)[8]) {
  if (ep.up_k) {   // ConvTranspose3d pixel shuffle: (coarse voxel, tap, co) -> fine voxel row, channel co
    const unsigned m = (unsigned)grow, k = (unsigned)ep.up_k, vv = (unsigned)ep.up_v;
    const unsigned q = fdiv(m, ep.up_dv), x = m - q * vv, q2 = fdiv(q, ep.up_dv), y = q - q2 * vv, b = fdiv(q2, ep.up_dv), z = q2 - b * vv;
    const unsigned tap = fdiv((unsigned)gcol, ep.up_dc);
    const int co = gcol - (int)tap * ep.up_cout;
    const unsigned tq = fdiv(tap, ep.up_dk), tx = tap - tq * k, tz = fdiv(tq, ep.up_dk), ty = tq - tz * k;
    const long V = (long)vv * k;
    const long fine = (((long)b * V + z * k + tz) * V + y * k + ty) * V + x * k + tx;
    if (ep.bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += ep.bias[co + j];
    }
    Vec8<T>::store((T*)ep.C + fine * ep.ldc + co, v);
    return;
  }
  if (ep.bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += ep.bias[gcol + j];
  }
  if (ep.win_on) {   // window-ordered GEMM row -> token row; pad rows have no token
    grow = win_to_tok(ep.wm, grow);
    if (grow < 0) return;
  }
  const long o = grow * ep.ldc + gcol;
  if (ep.act == 1) {
    Vec8<T>::store((T*)ep.C2 + o, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = sizeof(T) == 2 ? gelu_fast_f(v[j]) : gelu_f(v[j]);
  } else if (ep.act == 2) {
    float a[8];
    Vec8<T>::load((const T*)ep.C2 + o, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= sizeof(T) == 2 ? gelu_grad_fast_f(a[j]) : gelu_grad_f(a[j]);
  }
  if (ep.rowscale) {
    float s = ep.rowscale[grow / ep.rows_per_scale];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= s;
  }
  if (ep.resid) {
    float a[8];
    Vec8<T>::load((const T*)ep.resid + o, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
  }
  if (ep.accumulate) {
    float a[8];
    Vec8<T>::load((const T*)ep.C + o, a);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += a[j];
  }
  Vec8<T>::store((T*)ep.C + o, v);
}

// epilogue shared by the NT kernels: per wave, one 16-row m-tile at a time through a private LDS slab (16-byte row stores)
// (mw0, n0: first row / column of this wave's (16 MT) x (16 NT) sub-tile; `wave` picks the wave's private slab)
template <typename T, int MT, int NT>
__device__ __forceinline__ void nt_epilogue_w(f32x4 (&acc)[MT][NT], char* smem, const EpiParams& ep, int mw0, int n0, int M, int N, int wave, int lane, long zrow) {
  constexpr int BN = 16 * NT, SLD = BN + 4;
  const int g = lane >> 4, li = lane & 15;
  float* stg = reinterpret_cast<float*>(smem) + wave * 16 * SLD;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
#pragma unroll
    for (int b = 0; b < NT; ++b)   // the accumulators hold the TRANSPOSED tile (see the k-loops): four consecutive columns of one row per
      //                              fragment -> one 16-byte LDS store each (row stride BN + 4 floats: conflict-free for ds_write_b128)
      *reinterpret_cast<float4*>(stg + li * SLD + b * 16 + 4 * g) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int it = lane; it < 16 * (BN / 8); it += 64) {
      const int row = it / (BN / 8), cc = it - row * (BN / 8);
      const int grow = mw0 + a * 16 + row, gcol = n0 + cc * 8;
      if (grow < M && gcol < N) {
        float v[8];
        float4 x0 = *reinterpret_cast<const float4*>(stg + row * SLD + cc * 8);
        float4 x1 = *reinterpret_cast<const float4*>(stg + row * SLD + cc * 8 + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
        epilogue8<T>(ep, zrow + grow, gcol, v);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
template <typename T, int MT, int NT>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[MT][NT], char* smem, const EpiParams& ep, int m0, int n0, int M, int N, int wave, int lane, int zb = -1) {
  nt_epilogue_w<T, MT, NT>(acc, smem, ep, m0 + wave * 16 * MT, n0, M, N, wave, lane, (long)(zb < 0 ? (int)blockIdx.z : zb) * M);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt kernel: 256 threads = 4 waves stacked along M, each wave (16*MT) x (16*NT); WG tile (64*MT) x (16*NT)
// LDS: two stages of swizzled 128-byte rows (A: 64*MT rows, B: 16*NT rows); register prefetch of tile t+1
// overlaps the MFMAs of tile t (one barrier per K tile); epilogue restaged through LDS for 16-B row stores.
// ------------------------------------------------------------------------------------------------
template <typename T, int MT, int NT, class AL>
__global__ __launch_bounds__(256) void gemm_nt_kernel(AL al, const T* __restrict__ Bw, long ldb, int M, int N, int K, EpiParams ep) {
  constexpr int BM = 64 * MT, BN = 16 * NT, KT = Row128<T>::KT, KS = Row128<T>::KSTEPS, CE = 16 / (int)sizeof(T);
  constexpr int AR = BM / 32, BR = (BN + 31) / 32, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int lc = tid & 7, lr = tid >> 3;
  const int nbz = ep.ksplit > 1 ? ep.nbatch : (int)gridDim.z;      // grid.z = batch (x contraction splits)
  const int zb = (int)blockIdx.z % nbz, ksp = (int)blockIdx.z / nbz;

  typename AL::Row arow[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) al.init_row(arow[i], m0 + lr + 32 * i, M, zb);
  const T* brow[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    int n = n0 + lr + 32 * i;
    brow[i] = (lr + 32 * i < BN && n < N) ? Bw + (long)n * ldb : nullptr;
  }
  uint4 areg[AR], breg[BR];
  auto gload = [&](int kt) {
    const int k = kt * KT + lc * CE;
    typename AL::Kst ks = al.init_k(k, K);
#pragma unroll
    for (int i = 0; i < AR; ++i) areg[i] = al.load(arow[i], ks);
#pragma unroll
    for (int i = 0; i < BR; ++i) breg[i] = (brow[i] && k < K) ? *reinterpret_cast<const uint4*>(brow[i] + k) : make_uint4(0, 0, 0, 0);
  };
  auto sstore = [&](int st) {
    char* As = smem + st * STAGE;
    char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<uint4*>(As + swz_off(lr + 32 * i, lc)) = areg[i];
#pragma unroll
    for (int i = 0; i < BR; ++i)
      if (lr + 32 * i < BN) *reinterpret_cast<uint4*>(Bs + swz_off(lr + 32 * i, lc)) = breg[i];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = (K + KT - 1) / KT, kt0 = 0;
  if (ep.ksplit > 1) {   // this workgroup's share of the contraction
    const int per = (nk + ep.ksplit - 1) / ep.ksplit;
    kt0 = ksp * per;
    nk = kt0 + per < nk ? kt0 + per : nk;
  }
  if (kt0 < nk) { gload(kt0); sstore(kt0 & 1); }
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* As = smem + cur * STAGE;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      Frag<T> bf[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = lds_frag(Bs, b * 16 + li, s, g, (T*)nullptr);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        Frag<T> af = lds_frag(As, wave * 16 * MT + a * 16 + li, s, g, (T*)nullptr);
#pragma unroll
        for (int b = 0; b < NT; ++b) mma(acc[a][b], bf[b], af);   // transposed product: lane (m = li, g) holds columns 16 b + 4 g + r of row m
      }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  if (ep.ksplit > 1) {   // raw fp32 partial tile; summed (and `accumulate` applied) by nt_ksplit_reduce_kernel
    float* part = ep.kpart + ((long)(ksp * nbz + zb) * M) * N;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wave * 16 * MT + a * 16 + li, col = n0 + b * 16 + 4 * g + r;
          if (row < M && col < N) part[(long)row * N + col] = acc[a][b][r];
        }
    return;
  }
  nt_epilogue<T, MT, NT>(acc, smem, ep, m0, n0, M, N, wave, lane, zb);
}

// ------------------------------------------------------------------------------------------------
// gemm_nt for contractions of at most two k-tiles (bf16, K <= 128) on many rows: the 40^3-token Linears (K = 96), the decoder1 transpose conv.
// These launches are latency-bound at the workgroups per CU their LDS allows (PMC: a wave lives ~5 us -- load round trip, 0.3 us of MFMAs,
// then the store drain -- and issues for ~1 us).  With the transposed product the X rows are the MFMA B operand, i.e. lane (m, g) needs
// X[m][32 s + 8 g ..]: a plain 16-byte global load -- so the row tile never touches LDS, only the weight tile does (both k-tiles at once, one
// barrier), and the workgroup needs 25-34 KB instead of 40-48 KB: four to six workgroups per CU instead of three.
// ------------------------------------------------------------------------------------------------
// KS = k-steps of 32 held in registers: 4 (K <= 128) or 6 (K <= 192: the 20^3-token Linears of stage 1 -- three k-tiles of weights in LDS at once).
template <int NT, int KS = 4>
__global__ __launch_bounds__(256) void gemm_nt_k128_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bw, long ldb, int M, int N, int K, EpiParams ep) {
  constexpr int BN = 16 * NT, BR = (BN + 31) / 32;
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * BN, lc = tid & 7, lr = tid >> 3;
  const int row = m0 + wave * 16 + li;
  Frag<T> af[KS];
  {
    const bf16_t* ar = A + (long)(row < M ? row : M - 1) * lda + 8 * g;   // clamped rows are computed but never stored
#pragma unroll
    for (int s = 0; s < KS; ++s) af[s].v = 32 * s + 8 * g < K ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ar + 32 * s)) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
#pragma unroll
  for (int kt = 0; kt < KS / 2; ++kt) {
    const int k = kt * 64 + lc * 8;
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int rl = lr + 32 * i, n = n0 + rl;
      if (rl < BN) {
        const uint4 v = (n < N && k < K) ? *reinterpret_cast<const uint4*>(Bw + (long)n * ldb + k) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(smem + kt * BN * 128 + swz_off(rl, lc)) = v;
      }
    }
  }
  __syncthreads();
  f32x4 acc[1][NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) acc[0][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (32 * s < K) {
      const char* Bs = smem + (s >> 1) * BN * 128;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        Frag<T> bf = lds_frag(Bs, b * 16 + li, s & 1, g, (T*)nullptr);
        mma(acc[0][b], bf, af[s]);
      }
    }
  }
  __syncthreads();   // the epilogue slabs alias the weight tile
  nt_epilogue<T, 1, NT>(acc, smem, ep, m0, n0, M, N, wave, lane, 0);
}
template <int NT, int KS = 4>
static int launch_nt_k128(const bf16_t* A, long lda, const void* Bw, long ldb, int M, int N, int K, const EpiParams& ep, hipStream_t st) {
  constexpr int BN = 16 * NT, lds_main = (KS / 2) * BN * 128, lds_epi = 4 * 16 * (BN + 4) * 4, lds = lds_main > lds_epi ? lds_main : lds_epi;
  dim3 grid((M + 63) / 64, (N + BN - 1) / BN, 1);
  hipLaunchKernelGGL((gemm_nt_k128_kernel<NT, KS>), grid, dim3(256), lds, st, A, lda, (const bf16_t*)Bw, ldb, M, N, K, ep);
  NMH_CHECK_LAUNCH();
  return 0;
}

// sums the contraction splits: out[row][col] (T, leading dimension ldc) = sum_s part[s][row][col] (+ out if accumulate); 8 columns per thread
template <typename T> __global__ void nt_ksplit_reduce_kernel(const float* __restrict__ part, int ksplit, long rows, int N, T* __restrict__ out, long ldc, int accumulate) {
  const long n8 = rows * (N / 8);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (N / 8);
    const int c8 = (int)(i - row * (N / 8)) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ksplit; ++s) {
      const float4 p0 = *reinterpret_cast<const float4*>(part + ((long)s * rows + row) * N + c8), p1 = *reinterpret_cast<const float4*>(part + ((long)s * rows + row) * N + c8 + 4);
      v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    }
    if (accumulate) {
      float o[8];
      Vec8<T>::load(out + row * ldc + c8, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    Vec8<T>::store(out + row * ldc + c8, v);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt, LDS-DMA pipelined variant (bf16, plain row-major A): same tiles / swizzle / fragments / epilogue as gemm_nt_kernel, but
// the operand tiles go global -> LDS by `global_load_lds_dwordx4` into a ring of ST stages with ST-1 K-tiles in flight (counted
// vmcnt + one raw s_barrier per K-tile).  The register-prefetch kernel has ONE tile in flight per workgroup, so a short-M GEMM
// (encoder stage 2/3: 252..1008 workgroups) pays a full global-load latency per K-tile; here it pays it once.
// The DMA image is lane-linear (lane l -> 16 bytes at dst + 16 l = row l/8, chunk position l%8), so the XOR swizzle is applied on
// the source side: the lane at chunk position c' fetches logical chunk c' ^ ((row>>1)&7).
// ------------------------------------------------------------------------------------------------
__device__ uint4 g_zero16_nt[1];   // zero source for K-tail lanes

template <int MT, int NT, int ST>
__global__ __launch_bounds__(256) void gemm_nt_dma_kernel(const bf16_t* __restrict__ A, long lda, long rows_per_z, const bf16_t* __restrict__ Bw, long ldb, int M, int N, int K,
                                                          EpiParams ep) {
  using T = bf16_t;
  constexpr int BM = 64 * MT, BN = 16 * NT, BNP = (BN + 31) / 32 * 32, KT = 64;
  constexpr int AR = BM / 32, BR = BNP / 32, PCS = AR + BR, STAGE = (BM + BNP) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int lrow = lane >> 3, row8 = 8 * wave + lrow;
  const int kc = ((lane & 7) ^ ((row8 >> 1) & 7)) * 8;   // logical k offset (elements) this lane fetches in every piece
  const T* asrc[AR];
  const T* bsrc[BR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + 32 * i + row8;
    if (m > M - 1) m = M - 1;                              // clamped rows are computed but never stored
    asrc[i] = A + ((long)blockIdx.z * rows_per_z + m) * lda + kc;
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    int n = n0 + 32 * i + row8;
    if (n > N - 1) n = N - 1;
    bsrc[i] = Bw + (long)n * ldb + kc;
  }
  const int nk = (K + KT - 1) / KT;
  // the zero page lives in a VGPR pair so that the K-tail select is a v_cndmask, not a divergent branch around two load forms
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_nt;
  asm volatile("" : "+v"(zpage));
  auto issue = [&](int kt) {
    char* slot = smem + (kt % ST) * STAGE + wave * 1024;
    const bool ok = kt * KT + kc < K;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const void* src = (const void*)(ok ? (unsigned long long)(asrc[i] + kt * KT) : zpage);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const void* src = (const void*)(ok ? (unsigned long long)(bsrc[i] + kt * KT) : zpage);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + BM * 128 + i * 4096), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most the pieces of the ST-2 younger stages are outstanding (vmcnt retires in order)
    if (nk - 1 - kt >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone's pieces of stage kt are visible; everyone is done reading stage kt-1
    if (kt + ST - 1 < nk) issue(kt + ST - 1);   // refills the slot of stage kt-1
    const char* As = smem + (kt % ST) * STAGE;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> bf[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = lds_frag(Bs, b * 16 + li, s, g, (T*)nullptr);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        Frag<T> af = lds_frag(As, wave * 16 * MT + a * 16 + li, s, g, (T*)nullptr);
#pragma unroll
        for (int b = 0; b < NT; ++b) mma(acc[a][b], bf[b], af);   // transposed product: lane (m = li, g) holds columns 16 b + 4 g + r of row m
      }
    }
  }
  __syncthreads();   // the epilogue reuses the ring as staging space
  nt_epilogue<T, MT, NT>(acc, smem, ep, m0, n0, M, N, wave, lane);
}

template <int MT, int NT, int ST>
static int launch_nt_dma(const ADirect<bf16_t>& al, const void* Bw, long ldb, int M, int N, int K, int batch, const EpiParams& ep, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 16 * NT, BNP = (BN + 31) / 32 * 32;
  constexpr int lds_main = ST * (BM + BNP) * 128, lds_epi = 4 * 16 * (BN + 4) * 4;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, batch);
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_dma_kernel<MT, NT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_nt_dma_kernel<MT, NT, ST>), grid, dim3(256), lds, st, al.A, al.lda, al.rows_per_z, (const bf16_t*)Bw, ldb, M, N, K, ep);
  NMH_CHECK_LAUNCH();
  return 0;
}
// ------------------------------------------------------------------------------------------------
// gemm_nt, LDS-DMA ring, waves as a 2 x 2 grid (round 5).  The kernels above stack their four waves along M, so every wave reads the
// workgroup's whole B tile: a 64 x 96 tile costs 1.17 KB of LDS fragment reads per MFMA -- more than the 1 KB per MFMA the LDS can
// deliver at the matrix peak -- and fetches 160 operand rows per 6144 outputs.  Here a wave owns a (16 MT) x (16 NT) quadrant of a
// (32 MT) x (32 NT) workgroup tile (128 x 128: 0.5 KB of fragment reads per MFMA, 256 operand rows per 16384 outputs); tiles, swizzle,
// DMA image, counted-vmcnt ring and epilogue slabs are those of gemm_nt_dma_kernel.  1-D grid: XCD x takes a contiguous range of work
// items with the N tiles of one row panel adjacent, so an A panel is pulled into one L2 once.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int ST>
__global__ __launch_bounds__(256) void gemm_nt_w22_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bw, long ldb, int M, int N, int K, int ntn,
                                                          EpiParams ep) {
  using T = bf16_t;
  constexpr int BM = 32 * MT, BN = 32 * NT, KT = 64, PCS = MT + NT, STAGE = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  int bid = (int)blockIdx.x;
  {
    const int G = (int)gridDim.x, x = bid & 7, q = G >> 3, rem = G & 7;
    bid = x * q + (x < rem ? x : rem) + (bid >> 3);
  }
  const int mt = bid / ntn, nt = bid - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = lane >> 3, row8 = 8 * wave + lrow;
  const int kc = ((lane & 7) ^ ((row8 >> 1) & 7)) * 8;   // logical k offset (elements) this lane fetches in every piece
  const T* asrc[MT];
  const T* bsrc[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int m = m0 + 32 * i + row8;
    if (m > M - 1) m = M - 1;                              // clamped rows are computed but never stored
    asrc[i] = A + (long)m * lda + kc;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    int n = n0 + 32 * i + row8;
    if (n > N - 1) n = N - 1;
    bsrc[i] = Bw + (long)n * ldb + kc;
  }
  const int nk = (K + KT - 1) / KT;
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_nt;
  asm volatile("" : "+v"(zpage));
  auto issue = [&](int kt) {
    char* slot = smem + (kt % ST) * STAGE + wave * 1024;
    const bool ok = kt * KT + kc < K;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const void* src = (const void*)(ok ? (unsigned long long)(asrc[i] + kt * KT) : zpage);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const void* src = (const void*)(ok ? (unsigned long long)(bsrc[i] + kt * KT) : zpage);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + BM * 128 + i * 4096), 16, 0, 0);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    if (nk - 1 - kt >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // everyone's pieces of stage kt are visible; everyone is done reading stage kt-1
    if (kt + ST - 1 < nk) issue(kt + ST - 1);   // refills the slot of stage kt-1
    const char* As = smem + (kt % ST) * STAGE + wm * (16 * MT * 128);
    const char* Bs = smem + (kt % ST) * STAGE + BM * 128 + wn * (16 * NT * 128);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag<T> bf[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = lds_frag(Bs, b * 16 + li, s, g, (T*)nullptr);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        Frag<T> af = lds_frag(As, a * 16 + li, s, g, (T*)nullptr);
#pragma unroll
        for (int b = 0; b < NT; ++b) mma(acc[a][b], bf[b], af);   // transposed product: lane (m = li, g) holds columns 16 b + 4 g + r of row m
      }
    }
  }
  __syncthreads();   // the epilogue reuses the ring as staging space
  nt_epilogue_w<T, MT, NT>(acc, smem, ep, m0 + wm * 16 * MT, n0 + wn * 16 * NT, M, N, wave, lane, 0L);
}

template <int MT, int NT, int ST>
static int launch_nt_w22(const bf16_t* A, long lda, const void* Bw, long ldb, int M, int N, int K, const EpiParams& ep, hipStream_t st) {
  constexpr int BM = 32 * MT, BN = 32 * NT;
  constexpr int lds_main = ST * (BM + BN) * 128, lds_epi = 4 * 16 * (16 * NT + 4) * 4;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  const int mtiles = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_w22_kernel<MT, NT, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_nt_w22_kernel<MT, NT, ST>), dim3(mtiles * ntn), dim3(256), lds, st, A, lda, (const bf16_t*)Bw, ldb, M, N, K, ntn, ep);
  NMH_CHECK_LAUNCH();
  return 0;
}
// cfg = MT * 100 + NT * 10 + ST
static int launch_nt_w22_cfg(int cfg, const bf16_t* A, long lda, const void* Bw, long ldb, int M, int N, int K, const EpiParams& ep, hipStream_t st) {
  switch (cfg) {
#define W22(mt, nt, s) case mt * 100 + nt * 10 + s: return launch_nt_w22<mt, nt, s>(A, lda, Bw, ldb, M, N, K, ep, st)
    W22(4, 4, 2); W22(4, 4, 3); W22(4, 3, 2); W22(4, 3, 3); W22(2, 4, 2); W22(2, 4, 3); W22(2, 4, 4); W22(2, 3, 3); W22(2, 3, 4); W22(2, 2, 4); W22(4, 2, 3); W22(4, 2, 4);
#undef W22
  }
  return -2;
}

// The pipelined kernel pays off where few workgroups exist to hide a global-load latency per k-tile (measured, M <= 8192:
// fc2 4000x384x1536 21.9 -> 15.2 us, 500x768x3072 34.6 -> 24.5 us; with K = 384 the 4-deep ring holds most of the contraction at once:
// whole step at 1 grid/GPU 13.75 -> 13.35 ms with the threshold lowered from K >= 1024 to K >= 384); with >= 1000 workgroups the
// register-prefetch kernel's smaller LDS footprint (more co-resident workgroups) wins.  NMH_GEMM_DMA=0 disables, =3/4 forces it.
template <typename T, int MT, int NT, class AL>
static bool try_dma(const AL& al, const void* Bw, long ldb, int M, int N, int K, int batch, const EpiParams& ep, hipStream_t st, int* rc) {
  if constexpr (std::is_same<AL, ADirect<bf16_t>>::value && MT == 1) {
    static const int dma_st = [] { const char* e = getenv("NMH_GEMM_DMA"); return e ? atoi(e) : -1; }();
    if (dma_st == 0 || !lda_ok(al.lda, ldb)) return false;
    static const int min_k = [] { const char* e = getenv("NMH_GEMM_DMA_MINK"); return e ? atoi(e) : 384; }();
    static const long max_m = [] { const char* e = getenv("NMH_GEMM_DMA_MAXM"); return e ? atol(e) : 8192L; }();
    // (K < 1024: only while the launch has <= 640 workgroups -- 4000x1536x384 = 1008 workgroups measured 17.2 us with the register-prefetch kernel against 20.9, 8000x1536x384 = 2000 workgroups 37 us with the register-prefetch
    //  kernel, 50 us with this one)
    const long wgs = ((long)M + 64 * MT - 1) / (64 * MT) * ((N + 16 * NT - 1) / (16 * NT)) * batch;
    const bool auto_on = dma_st < 0 && (long)M * batch <= max_m && (K >= 1024 || (K >= min_k && wgs <= 640));
    if (dma_st == 3) { *rc = launch_nt_dma<MT, NT, 3>(al, Bw, ldb, M, N, K, batch, ep, st); return true; }
    if (dma_st == 4 || auto_on) { *rc = launch_nt_dma<MT, NT, 4>(al, Bw, ldb, M, N, K, batch, ep, st); return true; }
  }
  return false;
}

template <typename T, int MT, int NT, class AL>
static int launch_nt(const AL& al, const void* Bw, long ldb, int M, int N, int K, int batch, const EpiParams& ep, hipStream_t st) {
  constexpr int BM = 64 * MT, BN = 16 * NT;
  constexpr int lds_main = 2 * (BM + BN) * 128, lds_epi = 4 * 16 * (BN + 4) * 4;
  constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
  {
    int rc = 0;
    if (try_dma<T, MT, NT, AL>(al, Bw, ldb, M, N, K, batch, ep, st, &rc)) return rc;
  }
  dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN, batch * (ep.ksplit > 1 ? ep.ksplit : 1));
  static NmhPerDeviceOnce attr_set;  // > 64 KiB of dynamic LDS must be opted into once per kernel
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<T, MT, NT, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_nt_kernel<T, MT, NT, AL>), grid, dim3(256), lds, st, al, (const T*)Bw, ldb, M, N, K, ep);
  NMH_CHECK_LAUNCH();
  return 0;
}

// which launches take the 2 x 2-wave kernel, and with which tile (0: none).  NMH_GEMM_W22=<cfg> forces one tile everywhere it applies (tools/bench_nt_w22.py),
// NMH_GEMM_W22=0 disables.
static int w22_config(int M, int N, int K, const EpiParams& ep) {
  const char* e = getenv("NMH_GEMM_W22");   // (read per dispatch: the tile sweep changes it inside one process)
  const int forced = e ? atoi(e) : -1;
  if (forced >= 0) return (K >= 64 && N >= 64) ? forced : 0;
  // measured (tools/bench_nt_w22.py, graph replay, 8 grids; default dispatch -> this kernel): stage-2 Linears on 8000 rows 384 x 1536: 21.4 -> 17.3 us,
  // 384 x 1152: 17.1 -> 14.2, 384 x 384: 10.0 -> 9.2 (64 x 96 tiles, ring of 3: 60 KB), 1536 x 384 + GELU' epilogue: 31.1 -> 24.9 (64 x 128, ring of 2: 48 KB);
  // stage-1 Linears on 64000 rows 192 x 768: 49.7 -> 40.4, 192 x 576: 37.8 -> 34.0 (128 x 128, ring of 2).  4000 rows: 15.0 -> 13.9; 1000 rows lose (8.1 -> 12).
  if (ep.win_on) return 0;
  if (M >= 2048 && M <= 16384 && K >= 384) {
    if (N % 96 == 0 && N <= 768) return 233;
    if (N % 128 == 0) return 242;
  }
  if (M >= 32768 && N == 192 && K >= 576) return 442;
  return 0;
}

template <typename T, class AL>
static int dispatch_nt(const AL& al, const void* Bw, long ldb, int M, int N, int K, int batch, const EpiParams& ep, hipStream_t st) {
  if (N % 8 != 0 || K % 8 != 0) return -2;
  const int t16 = (N + 15) / 16;
  if constexpr (std::is_same<AL, ADirect<bf16_t>>::value) {
    if (batch == 1 && ep.ksplit <= 1 && !ep.up_k && lda_ok(al.lda, ldb)) {
      const int cfg = w22_config(M, N, K, ep);
      if (cfg > 0) return launch_nt_w22_cfg(cfg, al.A, al.lda, Bw, ldb, M, N, K, ep, st);
    }
  }
  if (const char* ov = getenv("NMH_GEMM_CFG")) {  // tuning override "MT,NT"
    int mt = ov[0] - '0', nt = atoi(ov + 2);
    if (mt == 1 && nt == 2) return launch_nt<T, 1, 2, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 1 && nt == 3) return launch_nt<T, 1, 3, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 1 && nt == 4) return launch_nt<T, 1, 4, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 1 && nt == 6) return launch_nt<T, 1, 6, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 1 && nt == 8) return launch_nt<T, 1, 8, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 2 && nt == 6) return launch_nt<T, 2, 6, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 2 && nt == 8) return launch_nt<T, 2, 8, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 4 && nt == 6) return launch_nt<T, 4, 6, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (mt == 4 && nt == 8) return launch_nt<T, 4, 8, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
  }
  // 64-row tiles (32-45 KB of LDS -> 3-5 workgroups per CU) for (a) small problems (stage 2/3 tokens, 10^3..20^3 decoder volumes), so
  // that more than a few dozen CUs get work, and (b) short contractions (every Linear of the encoder, the transpose convs): those
  // are HBM-bound with a latency-bound prologue/epilogue per workgroup, and co-resident workgroups are what hides it
  // (measured on 256000x288x96: 145 us with 256x128 tiles, 67 us with 64x96)
  if (((long)M * batch <= 8192 || K <= 1024) && t16 >= 4) {
    // wide outputs of the 40^3-token Linears (fc1 / fc2-dgrad: N = 384, K = 96, >= 256 k rows) are write-bound: 128-column tiles store
    // 256-byte-aligned row segments (96-column tiles split every second 128-byte line between two workgroups): 393 -> 312 us at 512 k rows
    if constexpr (std::is_same<AL, ADirect<bf16_t>>::value) {
      static const int k128_on = getenv("NMH_GEMM_K128") ? atoi(getenv("NMH_GEMM_K128")) : 1;
      static const long k128_min = getenv("NMH_GEMM_K128_MINM") ? atol(getenv("NMH_GEMM_K128_MINM")) : 16384;
      if (k128_on && batch == 1 && ep.ksplit <= 1 && K <= 128 && M >= k128_min && lda_ok(al.lda, ldb)) {
        static const long nt8_min = getenv("NMH_GEMM_K128_NT8_MINM") ? atol(getenv("NMH_GEMM_K128_NT8_MINM")) : 16384;   // (1 grid: 64000 rows -- 12.48 -> 12.43 ms with 128-column tiles there too)
        if (t16 % 8 == 0 && (long)M >= nt8_min) return launch_nt_k128<8>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
        if (t16 % 6 == 0) return launch_nt_k128<6>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
        if (t16 % 4 == 0) return launch_nt_k128<4>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
      }
      // (measured at 8 grids, 64000 rows x 768 / 576 / 192 x 192: step 51.75 ms with it, 51.59 without, 52.0 with 128-column tiles: off by default)
      static const int k192_on = getenv("NMH_GEMM_K192") ? atoi(getenv("NMH_GEMM_K192")) : 0;
      if (k192_on && k128_on && batch == 1 && ep.ksplit <= 1 && K > 128 && K <= 192 && M >= k128_min && lda_ok(al.lda, ldb)) {
        if (k192_on == 8 && t16 % 8 == 0) return launch_nt_k128<8, 6>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
        if (t16 % 6 == 0) return launch_nt_k128<6, 6>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
        if (t16 % 4 == 0) return launch_nt_k128<4, 6>(al.A, al.lda, Bw, ldb, M, N, K, ep, st);
      }
    }
    // launches of a few dozen workgroups with a long contraction (stage 3 / 4 Linears at 1-2 grids per GPU): one workgroup per CU is
    // bound by its own LDS fragment reads (a wave reads the whole B tile: 56 KB per 64-deep k-tile with 96 columns, 24 KB with 32), so the
    // narrowest tile that still gives at most 1.5 workgroups per CU wins (measured, graph replay: 1000x384x1536 13.3 -> 8.0 us,
    // 125x768x3072 22.7 -> 12.9, 512x768x1536 13.3 -> 8.0, 216x768x768 8.1 -> 5.0; 4000x384x1536 stays at 96 columns: 15.1 vs 17.9 / 23.7)
    if constexpr (std::is_same<AL, ADirect<bf16_t>>::value) {
      static const int narrow_on = getenv("NMH_GEMM_NARROW") ? atoi(getenv("NMH_GEMM_NARROW")) : 1;
      if (narrow_on && K >= 384 && ep.ksplit <= 1 && lda_ok(al.lda, ldb)) {   // (aligned rows: these launches take the LDS-DMA kernel)
        const long rt = ((long)M + 63) / 64 * batch;
        auto fits = [&](int nt) { return t16 % nt == 0 && rt * (t16 / nt) <= 384; };
        if (rt * ((t16 + 5) / 6) <= 192) {   // (252 workgroups of 96 columns -- 4000x384x1536 -- are already one per CU)
          if (fits(2)) return launch_nt<T, 1, 2, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
          if (fits(3)) return launch_nt<T, 1, 3, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
          if (fits(4)) return launch_nt<T, 1, 4, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
        }
      }
    }
    if (t16 % 8 == 0 && K <= 128 && (long)M * batch >= 100000) return launch_nt<T, 1, 8, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (t16 % 6 == 0) return launch_nt<T, 1, 6, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (t16 % 4 == 0) return launch_nt<T, 1, 4, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    if (t16 % 3 == 0) return launch_nt<T, 1, 3, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
    return launch_nt<T, 1, 4, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
  }
  if (t16 <= 3) return launch_nt<T, 4, 3, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
  if (t16 == 4) return launch_nt<T, 4, 4, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
  // long contractions on big M (implicit-GEMM convs at 20^3..40^3, transpose-conv dgrad): 128-row tiles (2 workgroups per CU)
  // measured 10-30 % faster than 256-row tiles (1 per CU) on every decoder shape
  if (t16 % 6 == 0) return launch_nt<T, 2, 6, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
  return launch_nt<T, 2, 8, AL>(al, Bw, ldb, M, N, K, batch, ep, st);
}

int k_gemm_nt(int dt, const void* A, long lda, const void* Bw, long ldb, int M, int N, int K, const EpiParams& ep, hipStream_t st) {
  if (dt == NMH_DT_BF16) {
    ADirect<bf16_t> al{(const bf16_t*)A, lda, (long)M};
    return dispatch_nt<bf16_t>(al, Bw, ldb, M, N, K, 1, ep, st);
  }
  ADirect<float> al{(const float*)A, lda, (long)M};
  return dispatch_nt<float>(al, Bw, ldb, M, N, K, 1, ep, st);
}

// output tiles of the tile shape dispatch_nt picks for a conv (see there): used to decide the contraction split
static long conv_nt_tiles(long M, int N, int batch, int K) {
  const int t16 = (N + 15) / 16;
  int bm, bn;
  if ((M * batch <= 8192 || K <= 1024) && t16 >= 4) { bm = 64; bn = t16 % 6 == 0 ? 96 : (t16 % 4 == 0 ? 64 : (t16 % 3 == 0 ? 48 : 64)); }
  else if (t16 <= 3) { bm = 256; bn = 48; }
  else if (t16 == 4) { bm = 256; bn = 64; }
  else { bm = 128; bn = t16 % 6 == 0 ? 96 : 128; }
  return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
}
int k_conv3_nt(int dt, const void* X, const void* Wp, int B, int D, int H, int W, int Cin, int Cout, const EpiParams& ep0, hipStream_t st, float* ws, long ws_floats) {
  const int M = D * H * W, K = 27 * Cin;
  EpiParams ep = ep0;
  ep.ksplit = 1; ep.nbatch = B; ep.kpart = nullptr;
  // Small volumes (the 10^3 / 20^3 decoder levels): a 768 -> 384 conv at 10^3 is 64 output tiles per grid with 324 k-tiles each -- a quarter of
  // the chip, every workgroup latency-bound (218 us for 16 GFLOP).  Split the contraction so that ~1024 workgroups exist (up to 600 tiles).
  static const int ks_max = getenv("NMH_CONV_KSPLIT") ? atoi(getenv("NMH_CONV_KSPLIT")) : 8;
  const bool plain = !ep.bias && !ep.act && !ep.resid && !ep.rowscale && !ep.up_k && !ep.win_on && Cout % 8 == 0;
  // (bf16 only: the fp32 parity mode keeps one accumulation chain per output element -- the chaotic golden trace g13 is sensitive to the
  //  summation order of its 6.7e4-norm gradients)
  if (plain && ws && ks_max > 1 && dt == NMH_DT_BF16) {
    const long tiles = conv_nt_tiles(M, Cout, B, K);
    const int nk = (K + (dt == NMH_DT_BF16 ? 64 : 32) - 1) / (dt == NMH_DT_BF16 ? 64 : 32);
    static const long ks_tiles = getenv("NMH_CONV_KS_TILES") ? atol(getenv("NMH_CONV_KS_TILES")) : 600, ks_target = getenv("NMH_CONV_KS_TARGET") ? atol(getenv("NMH_CONV_KS_TARGET")) : 1024;
    int s = (int)((ks_target + tiles - 1) / tiles);
    if (s > ks_max) s = ks_max;
    if (s > nk / 16) s = nk / 16;
    while (s > 1 && (long)s * B * M * Cout > ws_floats) --s;
    if (tiles <= ks_tiles && s > 1) { ep.ksplit = s; ep.kpart = ws; }
  }
  int rc;
  static const int fast_on = getenv("NMH_CONV_FAST_A") ? atoi(getenv("NMH_CONV_FAST_A")) : 1;
  if (dt == NMH_DT_BF16 && fast_on && Cin % 64 == 0 && (long)B * M * Cin * 2 < (1L << 31)) {
    AConv3F<bf16_t> al{(const bf16_t*)X, Cin, D, H, W, make_fdiv(W), make_fdiv(H), make_fdiv(Cin)};
    // 128 x 192 tiles where the output width allows (every decoder level: 192 / 384 / 768): the kernel is bound by L2 -> LDS traffic (1 / BM + 1 / BN per FLOP: the
    // 128 x 96 tile moved 4.7 GB for the 255-GFLOP conv at 20^3), not by load latency (an LDS-DMA ring of 3-4 stages measured 10 % SLOWER at the same tile) and
    // not by the address arithmetic alone (AConv3F at 128 x 96: -7 %); 256 x 192 spills.  NMH_CONV_TILE = 26: the tile of rounds 2-5.  Sum of the six 10^3 / 20^3
    // shapes at 8 grids: 1622 -> 1261 us (tools/bench_aconv3_sweep.py)
    static const int tile_cfg = getenv("NMH_CONV_TILE") ? atoi(getenv("NMH_CONV_TILE")) : 212;
    if (tile_cfg == 212 && Cout % 192 == 0) rc = launch_nt<bf16_t, 2, 12, AConv3F<bf16_t>>(al, Wp, K, M, Cout, K, B, ep, st);
    else rc = dispatch_nt<bf16_t>(al, Wp, K, M, Cout, K, B, ep, st);
  } else if (dt == NMH_DT_BF16) {
    AConv3<bf16_t> al{(const bf16_t*)X, Cin, D, H, W, make_fdiv(W), make_fdiv(H), make_fdiv(Cin)};
    rc = dispatch_nt<bf16_t>(al, Wp, K, M, Cout, K, B, ep, st);
  } else {
    AConv3<float> al{(const float*)X, Cin, D, H, W, make_fdiv(W), make_fdiv(H), make_fdiv(Cin)};
    rc = dispatch_nt<float>(al, Wp, K, M, Cout, K, B, ep, st);
  }
  if (rc || ep.ksplit <= 1) return rc;
  const long rows = (long)B * M, n8 = rows * (Cout / 8);
  unsigned nb = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
  if (dt == NMH_DT_BF16) hipLaunchKernelGGL(nt_ksplit_reduce_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, ws, ep.ksplit, rows, Cout, (bf16_t*)ep.C, ep.ldc, ep.accumulate);
  else hipLaunchKernelGGL(nt_ksplit_reduce_kernel<float>, dim3(nb), dim3(256), 0, st, ws, ep.ksplit, rows, Cout, (float*)ep.C, ep.ldc, ep.accumulate);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: dW[n][k] += sum_m A[m][n] * rs(m) * B[m][k]
// WG = 4 waves (2x2), wave tile (16*NTW) x (16*KTW); chunk = 64 contraction rows per barrier.
// grid = (n tiles, k tiles, m splits).  Output via OMap (row,col)->element offset, fp32 atomicAdd.
// ------------------------------------------------------------------------------------------------
struct BDirectTN {
  long ldb;
  template <typename T> __device__ __forceinline__ uint4 load(const T* B, long mglob, int k, int /*K*/, const TnGeom&) const {
    return *reinterpret_cast<const uint4*>(B + mglob * ldb + k);
  }
  template <typename T> __device__ __forceinline__ const T* addr(const T* B, long mglob, int k, const TnGeom&) const { return B + mglob * ldb + k; }
};
struct BConv3TN {
  // B[m][k] = X[voxel(m)+tap(k)][ci(k)] with zero padding; m is a global voxel index over (b,z,y,x)
  template <typename T> __device__ __forceinline__ uint4 load(const T* X, long mglob, int k, int /*K*/, const TnGeom& gm) const {
    unsigned tap = fdiv((unsigned)k, gm.dC);
    int ci = k - (int)tap * gm.Cin;
    int t9 = tap / 9, r9 = tap - t9 * 9, t3 = r9 / 3;
    int dz = t9 - 1, dy = t3 - 1, dx = r9 - t3 * 3 - 1;
    unsigned b = fdiv((unsigned)mglob, gm.dV);
    unsigned ml = (unsigned)mglob - b * gm.V;
    unsigned q = fdiv(ml, gm.dW);
    int x = ml - q * gm.W + dx;
    unsigned q2 = fdiv(q, gm.dH);
    int y = q - q2 * gm.H + dy, z = (int)q2 + dz;
    if ((unsigned)z >= (unsigned)gm.D || (unsigned)y >= (unsigned)gm.H || (unsigned)x >= (unsigned)gm.W) return make_uint4(0, 0, 0, 0);
    return *reinterpret_cast<const uint4*>(X + (mglob + ((long)dz * gm.H + dy) * gm.W + dx) * gm.Cin + ci);
  }
  // same element as load(), as an address (nullptr = zero padding)
  template <typename T> __device__ __forceinline__ const T* addr(const T* X, long mglob, int k, const TnGeom& gm) const {
    unsigned tap = fdiv((unsigned)k, gm.dC);
    int ci = k - (int)tap * gm.Cin;
    int t9 = tap / 9, r9 = tap - t9 * 9, t3 = r9 / 3;
    int dz = t9 - 1, dy = t3 - 1, dx = r9 - t3 * 3 - 1;
    unsigned b = fdiv((unsigned)mglob, gm.dV);
    unsigned ml = (unsigned)mglob - b * gm.V;
    unsigned q = fdiv(ml, gm.dW);
    int x = ml - q * gm.W + dx;
    unsigned q2 = fdiv(q, gm.dH);
    int y = q - q2 * gm.H + dy, z = (int)q2 + dz;
    if ((unsigned)z >= (unsigned)gm.D || (unsigned)y >= (unsigned)gm.H || (unsigned)x >= (unsigned)gm.W) return nullptr;
    return X + (mglob + ((long)dz * gm.H + dy) * gm.W + dx) * gm.Cin + ci;
  }
};

__device__ __forceinline__ long omap_index(const TnGeom& gm, int n, int k) {
  switch (gm.omode) {
    case 1: {  // conv3 weight [Cout][Cin][27]: n = co, k = tap*Cin + ci
      unsigned tap = fdiv((unsigned)k, gm.dC);
      int ci = k - (int)tap * gm.Cin;
      return ((long)n * gm.Cin + ci) * 27 + tap;
    }
    case 2: {  // convT weight [Cin][Cout][k3]: n = tap*Cout + co (tap = n / Cout), k = ci ; gm.Cin := Cout, gm.V := k3
      unsigned tap = fdiv((unsigned)n, gm.dC);
      int co = n - (int)tap * gm.Cin;
      return ((long)k * gm.Cin + co) * gm.V + tap;
    }
    default: return (long)n * gm.ldo + k;
  }
}

template <int X> struct OddRS { static constexpr int v = ((X + 31) / 32 % 2 == 1) ? (X + 31) / 32 * 32 : ((X + 31) / 32 + 1) * 32; };

template <typename T, int NTW, int KTW, class BL>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const T* __restrict__ A, long lda, const T* __restrict__ Bm, BL bl, float* __restrict__ Out,
                                                       long Mtot, int N, int K, int m_per_split, const float* __restrict__ rowscale, int rows_per_scale, TnGeom gm) {
  constexpr int BNW = 32 * NTW, BKW = 32 * KTW, CH = 64, CE = 16 / (int)sizeof(T);
  constexpr int RSA = OddRS<BNW * (int)sizeof(T)>::v, RSB = OddRS<BKW * (int)sizeof(T)>::v;
  constexpr int CPA = BNW / CE, CPB = BKW / CE;          // 16-B chunks per row
  constexpr int IA = (CH * CPA + 255) / 256, IB = (CH * CPB + 255) / 256;
  __shared__ __attribute__((aligned(16))) char sA[CH * RSA];
  __shared__ __attribute__((aligned(16))) char sB[CH * RSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave >> 1, wk = wave & 1, g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * BNW, k0 = blockIdx.y * BKW;
  const long mbeg = (long)blockIdx.z * m_per_split;
  long mend = mbeg + m_per_split;
  if (mend > Mtot) mend = Mtot;

  f32x4 acc[NTW][KTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient: the first k-tile's wk==0 waves also multiply their A fragments with an all-ones B fragment, which
  // leaves the column sums of A (identical in all 16 columns) in NTW extra accumulators -- one more MFMA per A fragment
  // instead of a separate pass over A
  const bool want_bias = gm.dbias != nullptr && blockIdx.y == 0 && wk == 0;
  f32x4 bacc[NTW];
  Frag<T> ones;
#pragma unroll
  for (int a = 0; a < NTW; ++a) bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (sizeof(T) == 2) ones.v[j] = (short)0x3F80; else ones.v[j] = 1.0f;
  }

  uint4 ra[IA], rb[IB];
  auto gload = [&](long mc) {
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      int it = tid + 256 * i, r = it / CPA, c = it - r * CPA;
      long m = mc + r;
      int n = n0 + c * CE;
      ra[i] = make_uint4(0, 0, 0, 0);
      if (it < CH * CPA && m < mend && n < N) {
        if (gm.up_k) {   // pixel-shuffled view of the fine-grid gradient (ConvTranspose3d backward)
          const unsigned mu = (unsigned)m, k = (unsigned)gm.up_k, vv = (unsigned)gm.up_v;
          const unsigned q = fdiv(mu, gm.up_dv), x = mu - q * vv, q2 = fdiv(q, gm.up_dv), y = q - q2 * vv, b = fdiv(q2, gm.up_dv), z = q2 - b * vv;
          const unsigned tap = fdiv((unsigned)n, gm.dC);
          const int co = n - (int)tap * gm.Cin;
          const unsigned tq = fdiv(tap, gm.up_dk), tx = tap - tq * k, tz = fdiv(tq, gm.up_dk), ty = tq - tz * k;
          const long V = (long)vv * k;
          const long fine = (((long)b * V + z * k + tz) * V + y * k + ty) * V + x * k + tx;
          ra[i] = *reinterpret_cast<const uint4*>(A + fine * gm.up_ldc + co);
        } else
        ra[i] = *reinterpret_cast<const uint4*>(A + m * lda + n);
        if (rowscale) {
          float s = rowscale[m / rows_per_scale];
          float v[8];
          if constexpr (sizeof(T) == 2) {
            unsigned w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[0] = __uint_as_float(w[q] << 16) * s; v[1] = __uint_as_float(w[q] & 0xffff0000u) * s;
              w[q] = pk_bf16(v[0], v[1]);
            }
            ra[i] = make_uint4(w[0], w[1], w[2], w[3]);
          } else {
            ra[i].x = __float_as_uint(__uint_as_float(ra[i].x) * s); ra[i].y = __float_as_uint(__uint_as_float(ra[i].y) * s);
            ra[i].z = __float_as_uint(__uint_as_float(ra[i].z) * s); ra[i].w = __float_as_uint(__uint_as_float(ra[i].w) * s);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      int it = tid + 256 * i, r = it / CPB, c = it - r * CPB;
      long m = mc + r;
      int k = k0 + c * CE;
      rb[i] = (it < CH * CPB && m < mend && k < K) ? bl.template load<T>(Bm, m, k, K, gm) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      int it = tid + 256 * i, r = it / CPA, c = it - r * CPA;
      if (it < CH * CPA) *reinterpret_cast<uint4*>(sA + r * RSA + c * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      int it = tid + 256 * i, r = it / CPB, c = it - r * CPB;
      if (it < CH * CPB) *reinterpret_cast<uint4*>(sB + r * RSB + c * 16) = rb[i];
    }
  };

  if (mbeg < mend) gload(mbeg);
  for (long mc = mbeg; mc < mend; mc += CH) {
    __syncthreads();  // previous chunk fully consumed
    sstore();
    __syncthreads();
    if (mc + CH < mend) gload(mc + CH);
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
      Frag<T> bf[KTW];
#pragma unroll
      for (int b = 0; b < KTW; ++b) bf[b] = lds_frag_t(sB, RSB, s * 32, (wk * KTW + b) * 16, lane, (T*)nullptr);
#pragma unroll
      for (int a = 0; a < NTW; ++a) {
        Frag<T> af = lds_frag_t(sA, RSA, s * 32, (wn * NTW + a) * 16, lane, (T*)nullptr);
#pragma unroll
        for (int b = 0; b < KTW; ++b) mma(acc[a][b], af, bf[b]);
        if (want_bias) mma(bacc[a], af, ones);
      }
    }
  }
  // acc[a][b][r]: row n = n0 + (wn*NTW+a)*16 + 4g + r, col k = k0 + (wk*KTW+b)*16 + li
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = n0 + (wn * NTW + a) * 16 + 4 * g + r, k = k0 + (wk * KTW + b) * 16 + li;
        if (n < N && k < K) {
          if (gm.part) { gm.part[((long)blockIdx.z * N + n) * K + k] = acc[a][b][r]; continue; }  // summed by tn_reduce_kernel
          float* o = Out + omap_index(gm, n, k);
          if (gridDim.z == 1) *o += acc[a][b][r];  // sole owner of this output element: plain read-modify-write
          else atomicAdd(o, acc[a][b][r]);
        }
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * NTW + a) * 16 + 4 * g + r;
        if (n < N) {
          if (gm.part) { gm.part[(long)gridDim.z * N * K + (long)blockIdx.z * N + n] = bacc[a][r]; continue; }
          // (the shuffled view repeats each bias channel once per tap: always atomics there)
          const int nb = gm.up_k ? n - (int)fdiv((unsigned)n, gm.dC) * gm.Cin : n;
          if (gridDim.z == 1 && !gm.up_k) gm.dbias[nb] += bacc[a][r];
          else atomicAdd(gm.dbias + nb, bacc[a][r]);
        }
      }
  }
}


// ------------------------------------------------------------------------------------------------
// gemm_tn, LDS-DMA pipelined variant (bf16): same 96x96 workgroup tile, wave layout, slot map and epilogue as gemm_tn_kernel, but
// the 64-row operand chunks go global -> LDS by `global_load_lds_dwordx4` into a ring of ST stages (ST-1 chunks in flight, counted
// vmcnt + one raw s_barrier per chunk) instead of one register-staged chunk: the contraction loop of a weight gradient is long
// (M = tokens or voxels) and its chunks are small (24 KB), so the single-stage kernel is bound by one global-load latency per chunk.
// LDS tiles are dense (192-byte rows; the DMA image is lane-linear) with the 8-byte column chunks of rows 4..7 (mod 8) XORed by 4,
// which keeps the transpose reads conflict-free; the swizzle is applied on the source address of each DMA lane.
// Stochastic-depth row scales are applied to the accumulators: the launcher aligns the M splits to the samples.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Frag<bf16_t> lds_frag_t_sw(const char* tile, int m0, int col0, int lane) {
  constexpr int RS = 192;
  const int g = lane >> 4, p = lane & 15;
  const int row = m0 + 4 * g + (p >> 2);
  const int ch = ((col0 >> 2) + (p & 3)) ^ (((row >> 2) & 1) << 2);
  const char* a = tile + row * RS + ch * 8;
  bf16x4 lo = ds_read_tr16(a);
  bf16x4 hi = ds_read_tr16(a + 16 * RS);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}

template <class BL, int ST>
__global__ __launch_bounds__(256) void gemm_tn_dma_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ Bm, BL bl, float* __restrict__ Out, long Mtot, int N,
                                                          int K, int m_per_split, const float* __restrict__ rowscale, int rows_per_scale, TnGeom gm) {
  using T = bf16_t;
  constexpr int NTW = 3, KTW = 3, BNW = 96, BKW = 96, CH = 64, RS = 192, TILE = CH * RS, STAGE = 2 * TILE, PCS = STAGE / 1024 / 4;  // 6 pieces per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave >> 1, wk = wave & 1, g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.x * BNW, k0 = blockIdx.y * BKW;
  const long mbeg = (long)blockIdx.z * m_per_split;
  long mend = mbeg + m_per_split;
  if (mend > Mtot) mend = Mtot;

  // this lane's position in each of its 6 pieces: piece i (< 3: A tile, >= 3: B tile) covers tile bytes [1024 (wave + 4 (i%3)) + 16 lane, +16)
  int prow[3], pcol[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int off = 1024 * (wave + 4 * i) + 16 * lane;
    const int row = off / RS, u = (off - row * RS) >> 4;
    prow[i] = row;
    pcol[i] = (u ^ (((row >> 2) & 1) << 1)) * 8;   // logical column (elements) stored at this LDS position
  }
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_nt;
  asm volatile("" : "+v"(zpage));
  auto issue = [&](int c) {   // chunk c of this split -> ring slot c % ST
    const long mc = mbeg + (long)c * CH;
    char* slot = smem + (c % ST) * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long m = mc + prow[i];
      const int n = n0 + pcol[i];
      unsigned long long src = zpage;
      if (m < mend && n < N) {
        if (gm.up_k) {
          const unsigned mu = (unsigned)m, k = (unsigned)gm.up_k, vv = (unsigned)gm.up_v;
          const unsigned q = fdiv(mu, gm.up_dv), x = mu - q * vv, q2 = fdiv(q, gm.up_dv), y = q - q2 * vv, b = fdiv(q2, gm.up_dv), z = q2 - b * vv;
          const unsigned tap = fdiv((unsigned)n, gm.dC);
          const int co = n - (int)tap * gm.Cin;
          const unsigned tq = fdiv(tap, gm.up_dk), tx = tap - tq * k, tz = fdiv(tq, gm.up_dk), ty = tq - tz * k;
          const long V = (long)vv * k;
          const long fine = (((long)b * V + z * k + tz) * V + y * k + ty) * V + x * k + tx;
          src = (unsigned long long)(A + fine * gm.up_ldc + co);
        } else {
          src = (unsigned long long)(A + m * lda + n);
        }
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const long m = mc + prow[i];
      const int k = k0 + pcol[i];
      unsigned long long src = zpage;
      if (m < mend && k < K) {
        const T* p = bl.template addr<T>(Bm, m, k, gm);
        if (p) src = (unsigned long long)p;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + TILE + i * 4096), 16, 0, 0);
    }
  };

  f32x4 acc[NTW][KTW];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_bias = gm.dbias != nullptr && blockIdx.y == 0 && wk == 0;
  f32x4 bacc[NTW];
  Frag<T> ones;
#pragma unroll
  for (int a = 0; a < NTW; ++a) bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.v[j] = (short)0x3F80;

  const int nc = mbeg < mend ? (int)((mend - mbeg + CH - 1) / CH) : 0;
  unsigned aofs[NTW], bofs[KTW];   // per-lane fragment offsets inside a tile (lds_frag_t_sw's address)
  {
    const int fg = lane >> 4, fp = lane & 15, frow = 4 * fg + (fp >> 2);
#pragma unroll
    for (int a = 0; a < NTW; ++a) aofs[a] = (unsigned)(frow * RS + (((((wn * NTW + a) * 16) >> 2) + (fp & 3)) ^ (((frow >> 2) & 1) << 2)) * 8);
#pragma unroll
    for (int b = 0; b < KTW; ++b) bofs[b] = (unsigned)(frow * RS + (((((wk * KTW + b) * 16) >> 2) + (fp & 3)) ^ (((frow >> 2) & 1) << 2)) * 8);
  }
  const unsigned smem_u = lds_addr_u(smem);
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nc) issue(s);
  for (int c = 0; c < nc; ++c) {
    if (nc - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + ST - 1 < nc) issue(c + ST - 1);
    // raw transpose reads (common.hpp): the compiler-visible builtin is ordered behind EVERY pending LDS-DMA (`s_waitcnt vmcnt(0)`), the chunks just requested
    // included, which made this ring a synchronous load
    const unsigned st_u = smem_u + (unsigned)((c % ST) * STAGE);
    TrFrag fa[CH / 32][NTW], fb[CH / 32][KTW];
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
      for (int b = 0; b < KTW; ++b) tr_read_raw<16 * RS>(fb[s][b], st_u + TILE + bofs[b] + s * 32 * RS);
#pragma unroll
      for (int a = 0; a < NTW; ++a) tr_read_raw<16 * RS>(fa[s][a], st_u + aofs[a] + s * 32 * RS);
    }
    tr_wait();
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
      for (int b = 0; b < KTW; ++b) tr_pin(fb[s][b]);
#pragma unroll
      for (int a = 0; a < NTW; ++a) tr_pin(fa[s][a]);
    }
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
      for (int a = 0; a < NTW; ++a) {
        const Frag<T> af = tr_frag(fa[s][a]);
#pragma unroll
        for (int b = 0; b < KTW; ++b) mma(acc[a][b], af, tr_frag(fb[s][b]));
        if (want_bias) mma(bacc[a], af, ones);
      }
    }
  }
  const float sc = (rowscale && mbeg < Mtot) ? rowscale[mbeg / rows_per_scale] : 1.0f;   // split lies inside one sample (launcher)
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < KTW; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n = n0 + (wn * NTW + a) * 16 + 4 * g + r, k = k0 + (wk * KTW + b) * 16 + li;
        if (n < N && k < K) {
          if (gm.part) { gm.part[((long)blockIdx.z * N + n) * K + k] = acc[a][b][r] * sc; continue; }
          float* o = Out + omap_index(gm, n, k);
          if (gridDim.z == 1) *o += acc[a][b][r] * sc;
          else atomicAdd(o, acc[a][b][r] * sc);
        }
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * NTW + a) * 16 + 4 * g + r;
        if (n < N) {
          if (gm.part) { gm.part[(long)gridDim.z * N * K + (long)blockIdx.z * N + n] = bacc[a][r] * sc; continue; }
          const int nb = gm.up_k ? n - (int)fdiv((unsigned)n, gm.dC) * gm.Cin : n;
          if (gridDim.z == 1 && !gm.up_k) gm.dbias[nb] += bacc[a][r] * sc;
          else atomicAdd(gm.dbias + nb, bacc[a][r] * sc);
        }
      }
  }
}


// sums the split partials [gz][N][K] (+ [gz][N] bias sums) into Out / dbias: one thread per output element, coalesced over k
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, int gz, int N, int K, float* __restrict__ Out, TnGeom gm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x, NK = (long)N * K;
  if (i < NK) {
    float s = 0.f;
    for (int z = 0; z < gz; ++z) s += part[(long)z * NK + i];
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    Out[omap_index(gm, n, k)] += s;
  } else if (gm.dbias && i < NK + N) {
    const int n = (int)(i - NK);
    float s = 0.f;
    for (int z = 0; z < gz; ++z) s += part[(long)gz * NK + (long)z * N + n];
    if (gm.up_k) atomicAdd(gm.dbias + (n - (int)fdiv((unsigned)n, gm.dC) * gm.Cin), s);
    else gm.dbias[n] += s;
  }
}
// many splits of a small output (the 256000-row stage-0 gradients: 125-250 splits) would write more partial bytes than the operands
// hold; those keep the atomics
static int tn_ws_max_splits() { static const int v = getenv("NMH_TN_WS_SPLITS") ? atoi(getenv("NMH_TN_WS_SPLITS")) : 16; return v; }
// plain row-major output, K % 4 == 0: four output elements per thread (16-byte partial loads and one 16-byte read-modify-write)
__global__ __launch_bounds__(256) void tn_reduce4_kernel(const float* __restrict__ part, int gz, int N, int K, float* __restrict__ Out, TnGeom gm) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x, NK = (long)N * K, NK4 = NK >> 2;
  if (i4 < NK4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < gz; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long)z * NK + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long i = i4 * 4;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    float4* o = reinterpret_cast<float4*>(Out + (long)n * gm.ldo + k);
    float4 c = *o;
    c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w;
    *o = c;
  } else if (gm.dbias && i4 < NK4 + N) {
    const int n = (int)(i4 - NK4);
    float s = 0.f;
    for (int z = 0; z < gz; ++z) s += part[(long)gz * NK + (long)z * N + n];
    gm.dbias[n] += s;
  }
}
static int launch_tn_reduce(const TnGeom& gm, int gz, int N, int K, float* Out, hipStream_t st) {
  if (gm.omode == 0 && !gm.up_k && K % 4 == 0 && gm.ldo % 4 == 0 && (((uintptr_t)Out | (uintptr_t)gm.part) & 15) == 0) {
    const long tot4 = (long)N * K / 4 + (gm.dbias ? N : 0);
    hipLaunchKernelGGL(tn_reduce4_kernel, dim3((unsigned)((tot4 + 255) / 256)), dim3(256), 0, st, gm.part, gz, N, K, Out, gm);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  const long tot = (long)N * K + (gm.dbias ? N : 0);
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, gm.part, gz, N, K, Out, gm);
  NMH_CHECK_LAUNCH();
  return 0;
}

template <class BL, int ST>
static int launch_tn_dma(const void* A, long lda, const void* Bm, const BL& bl, float* Out, long Mtot, int N, int K, long mps, const float* rs, int rps, const TnGeom& gm0, hipStream_t st) {
  constexpr int lds = ST * 2 * 64 * 192;
  int gx = (N + 95) / 96, gy = (K + 95) / 96, gz = (int)((Mtot + mps - 1) / mps);
  TnGeom gm = gm0;
  gm.part = (gm.ws && gz > 1 && gz <= tn_ws_max_splits() && (long)gz * N * (K + 1) <= gm.ws_floats) ? gm.ws : nullptr;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<BL, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_tn_dma_kernel<BL, ST>), dim3(gx, gy, gz), dim3(256), lds, st, (const bf16_t*)A, lda, (const bf16_t*)Bm, bl, Out, Mtot, N, K, (int)mps, rs, rps, gm);
  NMH_CHECK_LAUNCH();
  if (gm.part) return launch_tn_reduce(gm, gz, N, K, Out, st);
  return 0;
}

template <typename T, int NTW, int KTW, class BL>
static int launch_tn(const void* A, long lda, const void* Bm, const BL& bl, float* Out, long Mtot, int N, int K, const float* rs, int rps, const TnGeom& gm, hipStream_t st) {
  constexpr int BNW = 32 * NTW, BKW = 32 * KTW;
  int gx = (N + BNW - 1) / BNW, gy = (K + BKW - 1) / BKW;
  // the load->LDS->MFMA chain of one workgroup is not software-pipelined: latency is hidden by co-resident workgroups, so a
  // long contraction (conv wgrad over 4M voxels) wants ~4 workgroups per CU; short ones are bounded by the atomics instead
  long want = (Mtot >= (1L << 20) ? 1024 : 512) / ((long)gx * gy);  // aim for ~512 workgroups; many output tiles => no split (and no atomics)
  if (want < 1) want = 1;
  long mps = (Mtot + want - 1) / want;
  mps = (mps + 63) / 64 * 64;
  long mps_min = 1024;                 // every split ends in BNW*BKW atomics: give it >= 16 chunks of MFMA work first
  if (const char* e = getenv("NMH_TN_MPS")) mps_min = atol(e);
  if (const char* e = getenv("NMH_TN_WANT")) { want = atol(e) / ((long)gx * gy); if (want < 1) want = 1; mps = ((Mtot + want - 1) / want + 63) / 64 * 64; }
  if (mps < mps_min) mps = mps_min;
  if constexpr (std::is_same<T, bf16_t>::value && NTW == 3 && KTW == 3) {
    // measured: 8-12 % on the implicit-GEMM conv weight gradients (their gather loader is the longer dependency chain), within
    // noise on the plain ones (which are bound by L2 -> LDS bandwidth of the 96x96 tiles, not by latency): default on for convs only
    static const int dma_st = [] { const char* e = getenv("NMH_TN_DMA"); return e ? atoi(e) : (std::is_same<BL, BConv3TN>::value ? 3 : 0); }();
    if (dma_st && lda % 8 == 0) {
      long m2 = mps;
      if (rs) {   // row scales are applied per split: every split must lie inside one sample
        long j = (rps + m2 - 1) / m2;
        while (j > 1 && rps % j) --j;
        m2 = rps / j;
      }
      if (dma_st == 2) return launch_tn_dma<BL, 2>(A, lda, Bm, bl, Out, Mtot, N, K, m2, rs, rps, gm, st);
      if (dma_st == 4) return launch_tn_dma<BL, 4>(A, lda, Bm, bl, Out, Mtot, N, K, m2, rs, rps, gm, st);
      return launch_tn_dma<BL, 3>(A, lda, Bm, bl, Out, Mtot, N, K, m2, rs, rps, gm, st);
    }
  }
  int gz = (int)((Mtot + mps - 1) / mps);
  TnGeom gp = gm;
  gp.part = (gp.ws && gz > 1 && gz <= tn_ws_max_splits() && (long)gz * N * (K + 1) <= gp.ws_floats) ? gp.ws : nullptr;
  hipLaunchKernelGGL((gemm_tn_kernel<T, NTW, KTW, BL>), dim3(gx, gy, gz), dim3(256), 0, st, (const T*)A, lda, (const T*)Bm, bl, Out, Mtot, N, K, (int)mps, rs, rps, gp);
  NMH_CHECK_LAUNCH();
  if (gp.part) return launch_tn_reduce(gp, gz, N, K, Out, st);
  return 0;
}

template <typename T, class BL>
static int dispatch_tn(const void* A, long lda, const void* Bm, const BL& bl, float* Out, long Mtot, int N, int K, const float* rs, int rps, const TnGeom& gm, hipStream_t st) {
  if (K % 8 != 0 || lda % 8 != 0) return -2;  // N may be ragged: A rows are padded to lda, output guarded by n < N
  if (N <= 48) return launch_tn<T, 2, 3, BL>(A, lda, Bm, bl, Out, Mtot, N, K, rs, rps, gm, st);   // 64 x 96 tile (N padded)
  return launch_tn<T, 3, 3, BL>(A, lda, Bm, bl, Out, Mtot, N, K, rs, rps, gm, st);                  // 96 x 96 tile
}

int k_gemm_tn(int dt, const void* A, long lda, const void* Bm, long ldb, float* Out, long M, int N, int K, const float* rowscale, int rows_per_scale, const TnGeom& gm, hipStream_t st) {
  BDirectTN bl{ldb};
  if (dt == NMH_DT_BF16) return dispatch_tn<bf16_t>(A, lda, Bm, bl, Out, M, N, K, rowscale, rows_per_scale, gm, st);
  return dispatch_tn<float>(A, lda, Bm, bl, Out, M, N, K, rowscale, rows_per_scale, gm, st);
}

int k_conv3_tn(int dt, const void* dY, const void* X, float* dW, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st) {
  TnGeom gm{};
  gm.omode = 1; gm.Cin = Cin; gm.D = D; gm.H = H; gm.W = W; gm.V = (unsigned)(D * H * W);
  gm.dC = make_fdiv(Cin); gm.dW = make_fdiv(W); gm.dH = make_fdiv(H); gm.dV = make_fdiv(gm.V);
  BConv3TN bl;
  long M = (long)B * D * H * W;
  if (dt == NMH_DT_BF16) return dispatch_tn<bf16_t>(dY, Cout, X, bl, dW, M, Cout, 27 * Cin, nullptr, 1, gm, st);
  return dispatch_tn<float>(dY, Cout, X, bl, dW, M, Cout, 27 * Cin, nullptr, 1, gm, st);
}

// ---- ConvTranspose3d (kernel = stride) ------------------------------------------------------------------------------------------
static void up_epi(EpiParams& ep, int k, int v, int Cout) {
  ep.up_k = k; ep.up_v = v; ep.up_cout = Cout;
  ep.up_dv = make_fdiv(v); ep.up_dk = make_fdiv(k); ep.up_dc = make_fdiv(Cout);
}
int k_upconv_fwd(int dt, const void* x, const void* Wt, const float* bias, void* cat, long ldc, int B, int v, int k, int Cin, int Cout, hipStream_t st) {
  if (Cout % 8 || Cin % 8) return -2;
  const int M = B * v * v * v, N = k * k * k * Cout;
  EpiParams ep{cat, ldc, bias, 0, nullptr, nullptr, nullptr, 1, 0};
  up_epi(ep, k, v, Cout);
  return k_gemm_nt(dt, x, Cin, Wt, Cin, M, N, Cin, ep, st);
}
int k_upconv_dgrad(int dt, const void* dcat, long ldc, const void* Wd, void* dx, int B, int v, int k, int Cin, int Cout, hipStream_t st) {
  if (Cout % 8 || Cin % 8) return -2;
  const int M = B * v * v * v, K = k * k * k * Cout;
  EpiParams ep{dx, Cin, nullptr, 0, nullptr, nullptr, nullptr, 1, 0};
  if (dt == NMH_DT_BF16) {
    AUp<bf16_t> al{(const bf16_t*)dcat, ldc, v, k, Cout, make_fdiv(v), make_fdiv(k), make_fdiv(Cout)};
    return dispatch_nt<bf16_t>(al, Wd, K, M, Cin, K, 1, ep, st);
  }
  AUp<float> al{(const float*)dcat, ldc, v, k, Cout, make_fdiv(v), make_fdiv(k), make_fdiv(Cout)};
  return dispatch_nt<float>(al, Wd, K, M, Cin, K, 1, ep, st);
}
int k_upconv_wgrad(int dt, const void* dcat, long ldc, const void* x, float* dW, float* dbias, int B, int v, int k, int Cin, int Cout, hipStream_t st) {
  if (Cout % 8 || Cin % 8) return -2;
  const long M = (long)B * v * v * v;
  const int k3 = k * k * k;
  TnGeom gm{};
  gm.omode = 2; gm.Cin = Cout; gm.V = (unsigned)k3; gm.dC = make_fdiv(Cout); gm.dbias = dbias;
  gm.up_k = k; gm.up_v = v; gm.up_ldc = ldc; gm.up_dv = make_fdiv(v); gm.up_dk = make_fdiv(k);
  return k_gemm_tn(dt, dcat, (long)k3 * Cout, x, Cin, dW, M, k3 * Cout, Cin, nullptr, 1, gm, st);
}
