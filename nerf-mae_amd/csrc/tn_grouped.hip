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
constexpr int MAXP = 40;
constexpr int BN = 96, BK = 96, CH = 64, RS = 192, TILE = CH * RS, STAGE = 2 * TILE, ST = 2, PCS = 6;
// Ring of ST chunk stages.  Round 3: two instead of three (49,152 instead of 73,728 B per workgroup) -- the launches run on the side stream under the
// input-gradient chain, whose GEMM workgroups need 25-45 KB of the same CUs' LDS: step 50.42 -> 50.17 ms at 8 grids (same box, three runs each)
constexpr int LDS_BYTES = ST * STAGE;

// one problem of a launch, compressed to 96 bytes so that 40 of them fit the 4 KB kernel-argument segment (a launch that carries a whole
// stage's problems fills the chip for several rounds of workgroups: 16-problem launches ended in a half-empty second round each)
struct Prob {
  const bf16_t* A; const bf16_t* B; float* Out; float* dbias; const float* rowscale;
  int lda, ldb;       // row strides in elements (< 2^31)
  int N, K;
  int rps, nsamp;     // rows per sample, samples: M = nsamp * rps
  int sub;            // sub-splits per sample (rows per sub-split: mps(), a multiple of 64); zs == 1: unused
  int zs;             // contraction splits: 1 (the workgroup walks every sample) or nsamp * sub
  int rbegin;         // first reduce workgroup (zs > 1)
  int son, sok;       // output strides: element (n, k) lives at Out[n * son + k * sok] (plain row-major: son = ldo, sok = 1)
  int upflags;        // bits 0-7 up_k, bit 8 bias_atomic, bits 16-31 up_v.  up_k > 0: A row m (a coarse voxel of a (nsamp, v, v, v) grid) is
                      // row ((b*V + z*k)*V + y*k)*V + x*k, V = v*k, of the fine tensor; bias_atomic: several problems of the launch share dbias
  int partoff;        // zs > 1: offset (floats) of this problem's partials in the launch workspace
  int ncol2;          // 0, or n_inner | stride_n2 << 16: column n = (n / n_inner, n % n_inner) -> Out[(n % n_inner) * son + (n / n_inner) * stride_n2 + k * sok]
  __device__ __host__ int tk(int bt) const { return (K + bt - 1) / bt; }
  __device__ __host__ int ntile(int bt) const { return ((N + bt - 1) / bt) * tk(bt); }   // bt: 96, or 192 in the 8-wave variant
  __device__ __host__ int mps() const { return ((rps + sub - 1) / sub + 63) / 64 * 64; }
  __device__ int up_k() const { return upflags & 0xff; }
  __device__ int up_v() const { return (int)((unsigned)upflags >> 16); }
  __device__ bool bias_atomic() const { return (upflags >> 8) & 1; }
  __device__ long out_off(int n, int k) const {
    if (ncol2) { const int ni = ncol2 & 0xffff, q = n / ni; return (long)(n - q * ni) * son + (long)q * (ncol2 >> 16) + (long)k * sok; }
    return (long)n * son + (long)k * sok;
  }
  __device__ int bias_col(int n) const { return ncol2 ? n % (ncol2 & 0xffff) : n; }
};
static_assert(sizeof(Prob) == 96, "kernel-argument budget");
struct Args { Prob p[MAXP]; int wbegin[MAXP]; float* ws; int nprob; int xcd; };   // wbegin[i]: first workgroup of problem i
static_assert(sizeof(Args) <= 4096, "HIP kernel-argument segment");
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

// BIG: 192x192 output tile, 8 waves (one workgroup per CU, LDS ring 3 x 48 KB).  The 96x96 kernel moves 24 KB into LDS per 1.18 MFLOP
// and, with two chunks per workgroup in flight, is bound by the bytes it can keep in flight (measured ~8 TB/s of L2->LDS traffic at
// 0.3-0.4 PF); the large tile halves the bytes per FLOP at the same bytes in flight.  The LDS image stays a set of 64-row x 96-column
// sub-tiles (2 per operand), so the fragment addressing and the DMA-side swizzle are those of the small kernel.
// REG: the chunk images travel global -> registers -> LDS (global_load_dwordx4 + ds_write_b128, two LDS stages) instead of by LDS-DMA.
template <bool RSC, bool BIG, bool REG>   // RSC: the launch contains a problem with row scales (second accumulator set)
__global__ __launch_bounds__(BIG ? 512 : 256) void gemm_tn_grouped_kernel(tng::Args ga) {
  using namespace tng;
  constexpr int NW = BIG ? 8 : 4, NSUB = BIG ? 4 : 2, BT = BIG ? 192 : 96, NB = BIG ? 6 : 3, STG = NSUB * TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave >> 1, wk = wave & 1, g = lane >> 4, li = lane & 15;
  // wave tile: 48 A columns x (48 | 96) B columns.  sa / sb: the operand sub-tile, acol0 / bcol0: first column inside it
  const int sa = BIG ? (wn >> 1) : 0, acol0 = (BIG ? (wn & 1) : wn) * 48, sb = BIG ? wk : 0, bcol0 = BIG ? 0 : wk * 48;
  // workgroup id -> work item.  The hardware deals workgroup ids round-robin over the 8 XCDs, each with its own 4 MB L2; the tiles of
  // one problem walk the same A / B panels in loose lockstep, so XCD x takes a CONTIGUOUS range of work items (a few whole problems):
  // a panel chunk is then fetched into one L2 once and hit by the problem's other tiles instead of being pulled into all eight.
  int bid = (int)blockIdx.x;
  if (ga.xcd) {
    const int G = (int)gridDim.x, x = bid & 7, q = G >> 3, rem = G & 7;
    bid = x * q + (x < rem ? x : rem) + (bid >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i)   // (wbegin is padded with INT_MAX past nprob: the scan is a few wide scalar loads)
    if (bid >= ga.wbegin[i]) pi = i;
  const Prob& P = ga.p[pi];
  const int lw = bid - ga.wbegin[pi];
  const int ptk = P.tk(BT), pnt = P.ntile(BT), pmps = P.mps();
  float* const ppart = ga.ws + P.partoff;
  const int z = lw / pnt, t = lw - z * pnt;
  const int nt = t / ptk, kt = t - nt * ptk;
  const int n0 = nt * BT, k0 = kt * BT;
  const int N = P.N, K = P.K;
  const bf16_t* __restrict__ A = P.A;
  const bf16_t* __restrict__ Bm = P.B;
  const long lda = P.lda, ldb = P.ldb;
  const int up_k = P.up_k(), up_v = P.up_v();

  // segments: rows [seg_lo(s), seg_hi(s)) for s in [s0, s1); flat: one segment per sample; split: a single sub-range of one sample
  int s0, s1, sublo = 0, subhi = P.rps;
  if (P.zs == 1) { s0 = 0; s1 = P.nsamp; }
  else {
    const int b = z / P.sub, j = z - b * P.sub;
    s0 = b; s1 = b + 1;
    sublo = j * pmps;
    subhi = sublo + pmps < P.rps ? sublo + pmps : P.rps;
    if (sublo > subhi) sublo = subhi;
  }
  const int seglen = subhi - sublo;
  const int cps = (seglen + CH - 1) / CH;          // chunks per segment
  const int nc = cps * (s1 - s0);

  // DMA piece i (of 6) of this wave: 1-KB piece q = wave + NW*i of the chunk image; sub-tile q / 12 (A sub-tiles first), piece q % 12
  int prow[PCS], pcol[PCS], psub[PCS];
#pragma unroll
  for (int i = 0; i < PCS; ++i) {
    const int q = wave + NW * i, sub = q / 12, off = 1024 * (q - sub * 12) + 16 * lane;
    const int row = off / RS, u = (off - row * RS) >> 4;
    psub[i] = sub;
    prow[i] = row;
    pcol[i] = (u ^ (((row >> 2) & 1) << 1)) * 8 + (sub % (NSUB / 2)) * 96;
  }
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_tng;
  asm volatile("" : "+v"(zpage));
  // issue stream state: (segment, chunk within segment) of the next chunk to request
  int iseg = s0, ilc = 0, islot = 0;
  uint4 stg[REG ? PCS : 1];
  auto issue = [&]() {
    const long segbase = (long)iseg * P.rps + sublo;
    const int r0 = ilc * CH;
    char* slot = smem + islot * STG;
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
      const int r = r0 + prow[i];
      const bool isA = BIG ? (psub[i] < NSUB / 2) : (i < 3);
      unsigned long long src = zpage;
      if (isA) {
        const int n = n0 + pcol[i];
        if (r < seglen && n < N) {
          long arow = segbase + r;
          if (up_k) {   // pixel-shuffled view (ConvTranspose3d k = stride backward): coarse voxel -> the fine row of this problem's tap
            const unsigned vv = (unsigned)up_v, kk = (unsigned)up_k, m = (unsigned)arow;
            const unsigned q = m / vv, x = m - q * vv, q2 = q / vv, y = q - q2 * vv, bb = q2 / vv, zq = q2 - bb * vv;
            const long Vf = (long)vv * kk;
            arow = (((long)bb * Vf + zq * kk) * Vf + y * kk) * Vf + x * kk;
          }
          if (up_k && P.ncol2) {   // folded tap row: column n = (tx, co) sits at fine row arow + tx, channel co (contiguous when lda == Cout)
            const int ni = P.ncol2 & 0xffff, q = n / ni;
            src = (unsigned long long)(A + (arow + q) * lda + (n - q * ni));
          } else src = (unsigned long long)(A + arow * lda + n);
        }
      } else {
        const int k = k0 + pcol[i];
        if (r < seglen && k < K) src = (unsigned long long)(Bm + (segbase + r) * ldb + k);
      }
      const int q = wave + NW * i;
      if constexpr (REG) stg[i] = src != zpage ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
      else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + q * 1024), 16, 0, 0);
    }
    if (++ilc == cps) { ilc = 0; ++iseg; }
    if (!REG && ++islot == ST) islot = 0;
  };
  auto sstore = [&](int slotidx) {   // REG: the staged chunk -> LDS stage `slotidx` (same image as the DMA writes: piece q at q * 1024 + lane * 16)
#pragma unroll
    for (int i = 0; i < PCS; ++i) *reinterpret_cast<uint4*>(smem + slotidx * STG + (wave + NW * i) * 1024 + lane * 16) = stg[REG ? i : 0];
  };

  f32x4 acc[3][NB], tot[RSC ? 3 : 1][RSC ? NB : 1];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (RSC) tot[RSC ? a : 0][RSC ? b : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const bool want_bias = P.dbias != nullptr && kt == 0 && wk == 0;
  f32x4 bacc[3], btot[RSC ? 3 : 1];
  Frag<bf16_t> ones;
#pragma unroll
  for (int a = 0; a < 3; ++a) { bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f}; if (RSC) btot[RSC ? a : 0] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.v[j] = (short)0x3F80;
  const bool scaled = RSC && P.rowscale != nullptr;

  auto compute = [&](const char* stage) {
    const char* sA = stage + sa * TILE;
    const char* sB = stage + (NSUB / 2 + sb) * TILE;
#pragma unroll
    for (int s = 0; s < CH / 32; ++s) {
      Frag<bf16_t> bf[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) bf[b] = tng_frag(sB, s * 32, bcol0 + b * 16, lane);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        Frag<bf16_t> af = tng_frag(sA, s * 32, acol0 + a * 16, lane);
#pragma unroll
        for (int b = 0; b < NB; ++b) mma(acc[a][b], af, bf[b]);
        if (want_bias) mma(bacc[a], af, ones);
      }
    }
  };
  int cseg = s0, clc = 0, cslot = 0;
  auto sample_end = [&]() {
    if (++clc == cps) {
      if (scaled) {   // end of a sample: fold its accumulators into the totals with the sample's stochastic-depth factor
        const float sc = P.rowscale[cseg];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { tot[RSC ? a : 0][RSC ? b : 0][r] += sc * acc[a][b][r]; acc[a][b][r] = 0.f; }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { btot[RSC ? a : 0][r] += sc * bacc[a][r]; bacc[a][r] = 0.f; }
        }
      }
      clc = 0; ++cseg;
    }
  };
  if constexpr (REG) {
    // chunk c+1 sits in registers (in flight) while chunk c is multiplied out of LDS stage c & 1; it is written to the other stage -- whose
    // last readers passed the barrier that closed iteration c-1 -- before the loads of chunk c+2 are issued.  One barrier per chunk.
    if (nc > 0) { issue(); sstore(0); }
    if (nc > 1) issue();
    __syncthreads();
    for (int c = 0; c < nc; ++c) {
      compute(smem + (c & 1) * STG);
      if (c + 1 < nc) {
        sstore((c + 1) & 1);
        if (c + 2 < nc) issue();
      }
      __syncthreads();
      sample_end();
    }
  } else {
#pragma unroll
    for (int s = 0; s < ST - 1; ++s)
      if (s < nc) issue();
    for (int c = 0; c < nc; ++c) {
      if (nc - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PCS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (c + ST - 1 < nc) issue();
      compute(smem + cslot * STG);
      if (++cslot == ST) cslot = 0;
      sample_end();
    }
  }
  // acc[a][b][r]: row n = n0 + sa*96 + acol0 + a*16 + 4g + r, col k = k0 + sb*96 + bcol0 + b*16 + li
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + sa * 96 + acol0 + a * 16 + 4 * g + r, k = k0 + sb * 96 + bcol0 + b * 16 + li;
        if (n < N && k < K) {
          const float v = scaled ? tot[RSC ? a : 0][RSC ? b : 0][r] : acc[a][b][r];
          if (P.zs > 1) ppart[((long)z * N + n) * K + k] = v;
          else P.Out[P.out_off(n, k)] += v;   // sole owner of this output element
        }
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + sa * 96 + acol0 + a * 16 + 4 * g + r;
        if (n < N) {
          const float v = scaled ? btot[RSC ? a : 0][r] : bacc[a][r];
          if (P.zs > 1) ppart[(long)P.zs * N * K + (long)z * N + n] = v;
          else if (P.bias_atomic()) atomicAdd(P.dbias + P.bias_col(n), v);
          else P.dbias[P.bias_col(n)] += v;
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
  const float* const ppart = ga.ws + P.partoff;
  const long NK = (long)P.N * P.K, NK4 = NK >> 2;   // K % 8 == 0
  const long i4 = (long)((int)blockIdx.x - P.rbegin) * 256 + threadIdx.x;
  if (i4 < NK4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < P.zs; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(ppart + (long)z * NK + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long i = i4 * 4;
    const int n = (int)(i / P.K), k = (int)(i - (long)n * P.K);
    float* o = P.Out + P.out_off(n, k);
    o[0] += s.x; o[P.sok] += s.y; o[2 * P.sok] += s.z; o[3 * P.sok] += s.w;
  } else if (P.dbias && i4 < NK4 + P.N) {
    const int n = (int)(i4 - NK4);
    float s = 0.f;
    for (int z = 0; z < P.zs; ++z) s += ppart[(long)P.zs * NK + (long)z * P.N + n];
    if (P.bias_atomic()) atomicAdd(P.dbias + P.bias_col(n), s);
    else P.dbias[P.bias_col(n)] += s;
  }
}

// host side: chunk the problem list into launches of <= 16 problems, decide the contraction splits per launch
int k_gemm_tn_grouped(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st, bool foreground) {
  using namespace tng;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipSuccess;
    auto setattr = [&](const void* f, int bytes) { if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    setattr((const void*)gemm_tn_grouped_kernel<false, false, false>, LDS_BYTES);
    setattr((const void*)gemm_tn_grouped_kernel<true, false, false>, LDS_BYTES);
    setattr((const void*)gemm_tn_grouped_kernel<false, true, false>, 2 * LDS_BYTES);
    setattr((const void*)gemm_tn_grouped_kernel<true, true, false>, 2 * LDS_BYTES);
    setattr((const void*)gemm_tn_grouped_kernel<false, true, true>, 2 * 4 * TILE);
    setattr((const void*)gemm_tn_grouped_kernel<true, true, true>, 2 * 4 * TILE);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  // workgroups a split launch aims at.  Round 3: 256 instead of 640 -- the launches run on the side stream UNDER the input-gradient chain (since
  // the queues really overlap), where fewer, longer workgroups take less from the chain's latency-bound kernels: 8 grids 51.1-51.2 -> 50.6-50.8 ms,
  // 4 grids 28.9 -> 28.7, 2 grids 17.67 -> 17.54, 1 grid 11.54 -> 11.47 (1280: 51.6-51.8; 192: 51.3-51.7; 320: 51.2)
  static const int target_bg = getenv("NMH_TNG_TARGET") ? atoi(getenv("NMH_TNG_TARGET")) : 256;
  // foreground launches (the last flushes of a backward pass: nothing but the patch-embedding backward is left to run beside them)
  static const int target_fg = getenv("NMH_TNG_TARGET_FG") ? atoi(getenv("NMH_TNG_TARGET_FG")) : 640;
  const int target_wgs = foreground ? target_fg : target_bg;
  static const int flat_tiles = getenv("NMH_TNG_FLAT") ? atoi(getenv("NMH_TNG_FLAT")) : 384;
  // longest contraction first: the workgroups of a launch are dealt in order, so the short tiles fill the tail of the last round
  std::vector<int> order(nprob);
  for (int i = 0; i < nprob; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return probs[a].M > probs[b].M; });
  const int nlaunch = (nprob + MAXP - 1) / MAXP, per = (nprob + nlaunch - 1) / nlaunch;   // equal shares rather than 40 + remainder
  for (int base = 0; base < nprob; base += per) {
    const int np = std::min(per, nprob - base);
    Args ga{};
    ga.nprob = np;
    ga.ws = ws;
    static const int xcd_map = getenv("NMH_TNG_XCD") ? atoi(getenv("NMH_TNG_XCD")) : 1;
    ga.xcd = xcd_map;
    long tiles = 0, tiles_big = 0;
    bool any_rs = false, all192 = true;
    for (int i = 0; i < np; ++i) {
      const TnProblemHost& h = probs[order[base + i]];
      if (h.K % 8 || h.lda % 8 || h.ldb % 8 || h.rows_per_sample <= 0 || h.M % h.rows_per_sample || !h.A || !h.B || !h.dW) return -4;
      if (h.up_k > 0 && (h.up_v <= 0 || h.up_k > 255 || h.up_v > 65535 || h.rows_per_sample != (long)h.up_v * h.up_v * h.up_v)) return -4;
      if (h.lda >= (1L << 31) || h.ldb >= (1L << 31) || h.ldo >= (1L << 31) || h.stride_k >= (1L << 31)) return -4;
      Prob& p = ga.p[i];
      p.A = (const bf16_t*)h.A; p.B = (const bf16_t*)h.B; p.Out = h.dW; p.dbias = h.dbias; p.rowscale = h.rowscale;
      p.lda = (int)h.lda; p.ldb = (int)h.ldb; p.N = h.N; p.K = h.K;
      p.son = (int)h.ldo; p.sok = h.stride_k > 0 ? (int)h.stride_k : 1;
      if (h.n_inner < 0 || h.n_inner > 65535 || (h.n_inner > 0 && (h.stride_n2 < 0 || h.stride_n2 > 32767 || h.N % h.n_inner))) return -4;
      p.ncol2 = h.n_inner > 0 ? (h.n_inner | ((int)h.stride_n2 << 16)) : 0;
      p.upflags = (h.up_k & 0xff) | ((h.bias_atomic || h.n_inner > 0 ? 1 : 0) << 8)   /* folded columns share dbias entries */ | (h.up_k > 0 ? (h.up_v << 16) : 0);
      p.rps = h.rows_per_sample; p.nsamp = (int)(h.M / h.rows_per_sample);
      tiles += p.ntile(96);
      tiles_big += p.ntile(192);
      all192 &= h.N % 192 == 0 && h.K % 192 == 0;
      any_rs |= h.rowscale != nullptr;
    }
    // 192x192 tiles (8 waves, one workgroup per CU) when they tile every problem exactly and still fill the chip a few times over
    // (minimum number of large tiles; 0 disables.  Round 3, end: 256 -- stage 2's and stage 3's launches.  With the two queues really overlapping the
    //  large tile's halved L2->LDS traffic shows in the step: 8 grids 49.74 -> 49.27 ms, 4 grids 28.05 -> 27.81, 2 grids 16.95 -> 16.80, 1 grid
    //  11.15 -> 11.11 (64: the same within noise); a timing-only run without ANY grouped launch reads 45.4 ms at 8 grids -- 4.4 ms of the step is what
    //  these "background" launches still cost the input-gradient chain)
    const int big_min = getenv("NMH_TNG_BIG") ? atoi(getenv("NMH_TNG_BIG")) : 256;
    const bool big = big_min > 0 && all192 && tiles_big >= big_min;
    const int bt = big ? 192 : 96;
    if (big) tiles = tiles_big;
    // enough output tiles to fill the chip: nobody splits.  Otherwise every problem of the launch is split at sample-aligned row
    // ranges so that the launch has ~target_wgs workgroups (bounded by 16 chunks of work per split and by the workspace)
    int S = tiles >= flat_tiles ? 1 : (int)((target_wgs + tiles - 1) / tiles);
    long wsoff = 0;
    int w = 0, rb = 0;
    bool any_split = false;
    for (int i = 0; i < MAXP; ++i) ga.wbegin[i] = 0x7fffffff;
    for (int i = 0; i < np; ++i) {
      Prob& p = ga.p[i];
      p.zs = 1; p.sub = 1; p.partoff = 0;
      if (S > 1) {
        int sub = (S + p.nsamp - 1) / p.nsamp;
        const int maxsub = std::max(1, p.rps / 1024);
        sub = std::max(1, std::min(sub, maxsub));
        int zs = p.nsamp * sub;
        const long need = (long)zs * p.N * (p.K + 1);
        if (zs > 1 && ws && wsoff + need <= ws_floats && wsoff + need < (1L << 31)) {
          p.zs = zs; p.sub = sub;
          p.partoff = (int)wsoff;
          wsoff += (need + 3) / 4 * 4;
          any_split = true;
        }
      }
      ga.wbegin[i] = w;
      w += p.ntile(bt) * p.zs;
      p.rbegin = rb;
      if (p.zs > 1) rb += (int)(((long)p.N * p.K / 4 + (p.dbias ? p.N : 0) + 255) / 256);
    }
    // Measured (tools/bench_tng.py, 16-40 stage-2 problems per launch): LDS-DMA 96x96 0.30-0.40 PF, register-staged 96x96 0.33-0.40 PF,
    // 192x192 0.38-0.50 PF; PMC: L2 hit rate 86 %, no LDS bank conflicts, waves parked in s_waitcnt / s_barrier 55-63 % of their cycles
    // (every tile of a problem waits for the same first-touch lines of the next chunk).  Inside the training step of rounds 1-2 the three variants
    // were indistinguishable (the two queues took turns); since the queue fix the large tile pays (above) and the register-staged transport costs
    // (+0.5 ms): LDS-DMA stays the transport, NMH_TNG_REG=1 / NMH_TNG_BIG=<min tiles> select the others (all covered by the parity tests).
    const int reg_stage = getenv("NMH_TNG_REG") ? atoi(getenv("NMH_TNG_REG")) : 0;
    if (reg_stage) {
      if (big) {
        if (any_rs) hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, true, true>), dim3(w), dim3(512), 2 * 4 * TILE, st, ga);
        else hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, true, true>), dim3(w), dim3(512), 2 * 4 * TILE, st, ga);
      } else {
        if (any_rs) hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, false, true>), dim3(w), dim3(256), 2 * 2 * TILE, st, ga);
        else hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, false, true>), dim3(w), dim3(256), 2 * 2 * TILE, st, ga);
      }
    } else if (big) {
      if (any_rs) hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, true, false>), dim3(w), dim3(512), 2 * LDS_BYTES, st, ga);
      else hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, true, false>), dim3(w), dim3(512), 2 * LDS_BYTES, st, ga);
    } else {
      if (any_rs) hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, false, false>), dim3(w), dim3(256), LDS_BYTES, st, ga);
      else hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, false, false>), dim3(w), dim3(256), LDS_BYTES, st, ga);
    }
    NMH_CHECK_LAUNCH();
    if (any_split) {
      hipLaunchKernelGGL(gemm_tn_grouped_reduce_kernel, dim3(rb), dim3(256), 0, st, ga);
      NMH_CHECK_LAUNCH();
    }
  }
  return 0;
}
