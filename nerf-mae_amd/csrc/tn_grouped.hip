// Grouped weight-gradient GEMM (gfx950, bf16): many independent  dW_p[N_p,K_p] += sum_m A_p[m,N_p]^T . B_p[m,K_p]  problems in ONE launch.
//
// The encoder's backward produces four weight gradients per Swin block (qkv, proj, fc1, fc2; swin_mae3d.py:366-369 backward) that do
// not feed the input-gradient chain.  Launched one by one they are 96 small kernels per stage-2 pass, each too small to fill the chip
// (a 384x384 gradient is 16 output tiles), so every one of them is split over the contraction, pays a second launch that sums the
// partials, and still runs at a fraction of the L2 rate because less than one workgroup per CU cannot hide a global-load latency per
// 64-row chunk.  Here the parallelism comes from the PROBLEMS instead of from split-K: the host queues the (dY, X, dW, dbias)
// descriptors while the input-gradient chain runs and issues them per stage; a launch carries up to 16 problems in its kernel
// arguments (no descriptor upload: the launch is graph-capturable as is), one 96x96 output tile per workgroup.  With >= ~400 tiles in
// a launch nothing is split: a workgroup owns its output tile, walks the whole contraction through a 3-stage LDS-DMA ring
// (`global_load_lds_dwordx4`, counted vmcnt, one s_barrier per 64-row chunk) and adds the result into dW with plain stores.  Small
// groups (stage 0/1: 12-48 tiles per block) are split at sample-aligned row ranges and summed by one grouped reduce launch.
//
// Stochastic-depth row scales (one factor per sample on the rows of A; the fc2 gradient) are applied to the ACCUMULATORS at sample
// boundaries -- the DMA path has no register pass over A -- so contraction ranges never straddle a sample: every sample's rows are
// walked as their own chunk sequence (the last chunk of a sample is zero-filled past its end).
// The bias gradient (column sums of A) comes out of the same pass: one extra MFMA per A fragment against an all-ones B fragment.
#include "common.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <vector>

namespace tng {
constexpr int MAXP = 16;
constexpr int BN = 96, BK = 96, CH = 64, RS = 192, TILE = CH * RS, STAGE = 2 * TILE, ST = 3, PCS = 6;
constexpr int LDS_BYTES = ST * STAGE;   // 73,728 B -> two workgroups per CU

struct Prob {
  const bf16_t* A; const bf16_t* B; float* Out; float* dbias; const float* rowscale; float* part;
  long lda, ldb, ldo;
  int N, K;
  int rps, nsamp;     // rows per sample, samples: M = nsamp * rps
  int sub, mps;       // sub-splits per sample and rows per sub-split (multiple of 64); zs == 1: unused
  int tk, ntile;      // k tiles, n tiles * k tiles
  int zs;             // contraction splits: 1 (the workgroup walks every sample) or nsamp * sub
  int wbegin;         // first workgroup of this problem
  int rbegin;         // first reduce workgroup (zs > 1)
  long son, sok;      // output strides: element (n, k) lives at Out[n * son + k * sok] (plain row-major: son = ldo, sok = 1)
  int up_k, up_v;     // > 0: A row m (a coarse voxel of a (nsamp, v, v, v) grid) is row ((b*V + z*k)*V + y*k)*V + x*k, V = v*k, of the fine tensor
  int bias_atomic;    // several problems of the launch share dbias: atomic adds
};
struct Args { Prob p[MAXP]; int nprob; };
}  // namespace tng

__device__ uint4 g_zero16_tng[1];

__device__ __forceinline__ Frag<bf16_t> tng_frag(const char* tile, int m0, int col0, int lane) {
  // dense 192-byte rows; the 8-byte column chunks of rows 4..7 (mod 8) are XORed by 4 (applied on the DMA source side)
  const int g = lane >> 4, p = lane & 15;
  const int row = m0 + 4 * g + (p >> 2);
  const int ch = ((col0 >> 2) + (p & 3)) ^ (((row >> 2) & 1) << 2);
  const char* a = tile + row * tng::RS + ch * 8;
  bf16x4 lo = ds_read_tr16(a);
  bf16x4 hi = ds_read_tr16(a + 16 * tng::RS);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}

template <bool RSC>   // RSC: the launch contains a problem with row scales (second accumulator set)
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(tng::Args ga) {
  using namespace tng;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave >> 1, wk = wave & 1, g = lane >> 4, li = lane & 15;
  // problem of this workgroup (uniform scan over <= 16 prefix sums)
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < ga.nprob; ++i)
    if ((int)blockIdx.x >= ga.p[i].wbegin) pi = i;
  const Prob& P = ga.p[pi];
  const int lw = (int)blockIdx.x - P.wbegin;
  const int z = lw / P.ntile, t = lw - z * P.ntile;
  const int nt = t / P.tk, kt = t - nt * P.tk;
  const int n0 = nt * BN, k0 = kt * BK;
  const int N = P.N, K = P.K;
  const bf16_t* __restrict__ A = P.A;
  const bf16_t* __restrict__ Bm = P.B;
  const long lda = P.lda, ldb = P.ldb;

  // segments: rows [seg_lo(s), seg_hi(s)) for s in [s0, s1); flat: one segment per sample; split: a single sub-range of one sample
  int s0, s1, sublo = 0, subhi = P.rps;
  if (P.zs == 1) { s0 = 0; s1 = P.nsamp; }
  else {
    const int b = z / P.sub, j = z - b * P.sub;
    s0 = b; s1 = b + 1;
    sublo = j * P.mps;
    subhi = sublo + P.mps < P.rps ? sublo + P.mps : P.rps;
    if (sublo > subhi) sublo = subhi;
  }
  const int seglen = subhi - sublo;
  const int cps = (seglen + CH - 1) / CH;          // chunks per segment
  const int nc = cps * (s1 - s0);

  int prow[3], pcol[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int off = 1024 * (wave + 4 * i) + 16 * lane;
    const int row = off / RS, u = (off - row * RS) >> 4;
    prow[i] = row;
    pcol[i] = (u ^ (((row >> 2) & 1) << 1)) * 8;
  }
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_tng;
  asm volatile("" : "+v"(zpage));
  // issue stream state: (segment, chunk within segment) of the next chunk to request
  int iseg = s0, ilc = 0, islot = 0;
  auto issue = [&]() {
    const long segbase = (long)iseg * P.rps + sublo;
    const int r0 = ilc * CH;
    char* slot = smem + islot * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = r0 + prow[i];
      const int n = n0 + pcol[i];
      unsigned long long src = zpage;
      if (r < seglen && n < N) {
        long arow = segbase + r;
        if (P.up_k) {   // pixel-shuffled view (ConvTranspose3d k = stride backward): coarse voxel -> the fine row of this problem's tap
          const unsigned vv = (unsigned)P.up_v, kk = (unsigned)P.up_k, m = (unsigned)arow;
          const unsigned q = m / vv, x = m - q * vv, q2 = q / vv, y = q - q2 * vv, bb = q2 / vv, zq = q2 - bb * vv;
          const long Vf = (long)vv * kk;
          arow = (((long)bb * Vf + zq * kk) * Vf + y * kk) * Vf + x * kk;
        }
        src = (unsigned long long)(A + arow * lda + n);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + i * 4096), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r = r0 + prow[i];
      const int k = k0 + pcol[i];
      unsigned long long src = zpage;
      if (r < seglen && k < K) src = (unsigned long long)(Bm + (segbase + r) * ldb + k);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + TILE + i * 4096), 16, 0, 0);
    }
    if (++ilc == cps) { ilc = 0; ++iseg; }
    if (++islot == ST) islot = 0;
  };

  f32x4 acc[3][3], tot[RSC ? 3 : 1][RSC ? 3 : 1];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (RSC) tot[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const bool want_bias = P.dbias != nullptr && kt == 0 && wk == 0;
  f32x4 bacc[3], btot[RSC ? 3 : 1];
  Frag<bf16_t> ones;
#pragma unroll
  for (int a = 0; a < 3; ++a) { bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f}; if (RSC) btot[a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.v[j] = (short)0x3F80;
  const bool scaled = RSC && P.rowscale != nullptr;

#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nc) issue();
  int cseg = s0, clc = 0, cslot = 0;
  for (int c = 0; c < nc; ++c) {
    if (nc - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + ST - 1 < nc) issue();
    const char* sA = smem + cslot * STAGE;
    const char* sB = sA + TILE;
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
      Frag<bf16_t> bf[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) bf[b] = tng_frag(sB, s * 32, (wk * 3 + b) * 16, lane);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        Frag<bf16_t> af = tng_frag(sA, s * 32, (wn * 3 + a) * 16, lane);
#pragma unroll
        for (int b = 0; b < 3; ++b) mma(acc[a][b], af, bf[b]);
        if (want_bias) mma(bacc[a], af, ones);
      }
    }
    if (++cslot == ST) cslot = 0;
    if (++clc == cps) {
      if (scaled) {   // end of a sample: fold its accumulators into the totals with the sample's stochastic-depth factor
        const float sc = P.rowscale[cseg];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { tot[RSC ? a : 0][RSC ? b : 0][r] += sc * acc[a][b][r]; acc[a][b][r] = 0.f; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { btot[RSC ? a : 0][r] += sc * bacc[a][r]; bacc[a][r] = 0.f; }
        }
      }
      clc = 0; ++cseg;
    }
  }
  // acc[a][b][r]: row n = n0 + (wn*3+a)*16 + 4g + r, col k = k0 + (wk*3+b)*16 + li
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * 3 + a) * 16 + 4 * g + r, k = k0 + (wk * 3 + b) * 16 + li;
        if (n < N && k < K) {
          const float v = scaled ? tot[RSC ? a : 0][RSC ? b : 0][r] : acc[a][b][r];
          if (P.zs > 1) P.part[((long)z * N + n) * K + k] = v;
          else P.Out[(long)n * P.son + (long)k * P.sok] += v;   // sole owner of this output element
        }
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * 3 + a) * 16 + 4 * g + r;
        if (n < N) {
          const float v = scaled ? btot[RSC ? a : 0][r] : bacc[a][r];
          if (P.zs > 1) P.part[(long)P.zs * N * K + (long)z * N + n] = v;
          else if (P.bias_atomic) atomicAdd(P.dbias + n, v);
          else P.dbias[n] += v;
        }
      }
  }
}

// sums the split partials of every split problem of the group into dW / dbias: 256 threads x 4 consecutive elements per workgroup
__global__ __launch_bounds__(256) void gemm_tn_grouped_reduce_kernel(tng::Args ga) {
  using namespace tng;
  int pi = -1;
#pragma unroll 1
  for (int i = 0; i < ga.nprob; ++i)
    if (ga.p[i].zs > 1 && (int)blockIdx.x >= ga.p[i].rbegin) pi = i;
  if (pi < 0) return;
  const Prob& P = ga.p[pi];
  const long NK = (long)P.N * P.K, NK4 = NK >> 2;   // K % 8 == 0
  const long i4 = (long)((int)blockIdx.x - P.rbegin) * 256 + threadIdx.x;
  if (i4 < NK4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < P.zs; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(P.part + (long)z * NK + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long i = i4 * 4;
    const int n = (int)(i / P.K), k = (int)(i - (long)n * P.K);
    float* o = P.Out + (long)n * P.son + (long)k * P.sok;
    o[0] += s.x; o[P.sok] += s.y; o[2 * P.sok] += s.z; o[3 * P.sok] += s.w;
  } else if (P.dbias && i4 < NK4 + P.N) {
    const int n = (int)(i4 - NK4);
    float s = 0.f;
    for (int z = 0; z < P.zs; ++z) s += P.part[(long)P.zs * NK + (long)z * P.N + n];
    if (P.bias_atomic) atomicAdd(P.dbias + n, s);
    else P.dbias[n] += s;
  }
}

// host side: chunk the problem list into launches of <= 16 problems, decide the contraction splits per launch
int k_gemm_tn_grouped(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st) {
  using namespace tng;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  static const int target_wgs = getenv("NMH_TNG_TARGET") ? atoi(getenv("NMH_TNG_TARGET")) : 640;
  static const int flat_tiles = getenv("NMH_TNG_FLAT") ? atoi(getenv("NMH_TNG_FLAT")) : 384;
  for (int base = 0; base < nprob; base += MAXP) {
    const int np = std::min(MAXP, nprob - base);
    Args ga{};
    ga.nprob = np;
    long tiles = 0;
    bool any_rs = false;
    for (int i = 0; i < np; ++i) {
      const TnProblemHost& h = probs[base + i];
      if (h.K % 8 || h.lda % 8 || h.ldb % 8 || h.rows_per_sample <= 0 || h.M % h.rows_per_sample || !h.A || !h.B || !h.dW) return -4;
      if (h.up_k > 0 && (h.up_v <= 0 || h.rows_per_sample != (long)h.up_v * h.up_v * h.up_v)) return -4;
      Prob& p = ga.p[i];
      p.A = (const bf16_t*)h.A; p.B = (const bf16_t*)h.B; p.Out = h.dW; p.dbias = h.dbias; p.rowscale = h.rowscale; p.part = nullptr;
      p.lda = h.lda; p.ldb = h.ldb; p.ldo = h.ldo; p.N = h.N; p.K = h.K;
      p.son = h.stride_k > 0 ? h.ldo : h.ldo; p.sok = h.stride_k > 0 ? h.stride_k : 1;
      p.up_k = h.up_k; p.up_v = h.up_v; p.bias_atomic = h.bias_atomic;
      p.rps = h.rows_per_sample; p.nsamp = (int)(h.M / h.rows_per_sample);
      p.tk = (h.K + BK - 1) / BK;
      p.ntile = ((h.N + BN - 1) / BN) * p.tk;
      tiles += p.ntile;
      any_rs |= h.rowscale != nullptr;
    }
    // enough output tiles to fill the chip: nobody splits.  Otherwise every problem of the launch is split at sample-aligned row
    // ranges so that the launch has ~target_wgs workgroups (bounded by 16 chunks of work per split and by the workspace)
    int S = tiles >= flat_tiles ? 1 : (int)((target_wgs + tiles - 1) / tiles);
    long wsoff = 0;
    int w = 0, rb = 0;
    bool any_split = false;
    for (int i = 0; i < np; ++i) {
      Prob& p = ga.p[i];
      p.zs = 1; p.sub = 1; p.mps = p.rps;
      if (S > 1) {
        int sub = (S + p.nsamp - 1) / p.nsamp;
        const int maxsub = std::max(1, p.rps / 1024);
        sub = std::max(1, std::min(sub, maxsub));
        int zs = p.nsamp * sub;
        const long need = (long)zs * p.N * (p.K + 1);
        if (zs > 1 && ws && wsoff + need <= ws_floats) {
          p.zs = zs; p.sub = sub;
          p.mps = ((p.rps + sub - 1) / sub + 63) / 64 * 64;
          p.part = ws + wsoff;
          wsoff += (need + 3) / 4 * 4;
          any_split = true;
        }
      }
      p.wbegin = w;
      w += p.ntile * p.zs;
      p.rbegin = rb;
      if (p.zs > 1) rb += (int)(((long)p.N * p.K / 4 + (p.dbias ? p.N : 0) + 255) / 256);
    }
    if (any_rs) hipLaunchKernelGGL(gemm_tn_grouped_kernel<true>, dim3(w), dim3(256), LDS_BYTES, st, ga);
    else hipLaunchKernelGGL(gemm_tn_grouped_kernel<false>, dim3(w), dim3(256), LDS_BYTES, st, ga);
    NMH_CHECK_LAUNCH();
    if (any_split) {
      hipLaunchKernelGGL(gemm_tn_grouped_reduce_kernel, dim3(rb), dim3(256), 0, st, ga);
      NMH_CHECK_LAUNCH();
    }
  }
  return 0;
}
