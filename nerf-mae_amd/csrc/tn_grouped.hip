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

// byte offset (inside a tile of dense 192-byte rows) of this lane's transpose-read address for contraction rows m0.. and columns col0..col0+15: tng_frag's address
__device__ __forceinline__ unsigned tng_frag_ofs(int m0, int col0, int lane) {
  const int g = lane >> 4, p = lane & 15;
  const int row = m0 + 4 * g + (p >> 2);
  const int ch = ((col0 >> 2) + (p & 3)) ^ (((row >> 2) & 1) << 2);
  return (unsigned)(row * tng::RS + ch * 8);
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
  // (round 5: a pointer-bumped issue path as in gemm_tn_stream_kernel was measured here and dropped: 12 live 64-bit pointers took the rowscale variant
  //  from 171 to 195 VGPRs = one workgroup less per CU, and its stage-1/2 launches ran 2x longer inside the step; profiles/r5c_step_kernels_by_shape_8grids.txt)
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

  // fragment addresses: per-lane offsets inside a stage, fixed for the kernel (the DMA variant reads them with raw ds_read_b64_tr_b16: common.hpp)
  unsigned aofs[3], bofs[NB];
#pragma unroll
  for (int a = 0; a < 3; ++a) aofs[a] = (unsigned)(sa * TILE) + tng_frag_ofs(0, acol0 + a * 16, lane);
#pragma unroll
  for (int b = 0; b < NB; ++b) bofs[b] = (unsigned)((NSUB / 2 + sb) * TILE) + tng_frag_ofs(0, bcol0 + b * 16, lane);
  const unsigned smem_u = lds_addr_u(smem);
  auto compute = [&](const char* stage) {
    if constexpr (REG) {   // (no LDS-DMA in flight: compiler-visible reads)
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
    } else {
      const unsigned st_u = smem_u + (unsigned)(stage - smem);
      TrFrag fb[2][NB], fa[2][3];
#pragma unroll
      for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) tr_read_raw<16 * RS>(fb[s][b], st_u + bofs[b] + s * 32 * RS);
#pragma unroll
        for (int a = 0; a < 3; ++a) tr_read_raw<16 * RS>(fa[s][a], st_u + aofs[a] + s * 32 * RS);
      }
      tr_wait();
#pragma unroll
      for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) tr_pin(fb[s][b]);
#pragma unroll
        for (int a = 0; a < 3; ++a) tr_pin(fa[s][a]);
      }
#pragma unroll
      for (int s = 0; s < CH / 32; ++s) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const Frag<bf16_t> af = tr_frag(fa[s][a]);
#pragma unroll
          for (int b = 0; b < NB; ++b) mma(acc[a][b], af, tr_frag(fb[s][b]));
          if (want_bias) mma(bacc[a], af, ones);
        }
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

// sums the split partials of every split problem of the group into dW / dbias.  A wave: 16 items (four consecutive output elements, or one bias entry)
// x 4 interleaved shares of the splits (lane = item + 16 * share), combined by two cross-lane adds in a fixed order -- four times the loads in flight of the
// one-thread-per-item form, whose 27-63 workgroups walked 100+ slabs as one dependent-latency chain each (28-66 us per reduce behind the streaming launches).
// No LDS, no barrier: the launch has to fit beside whatever persistent kernel owns the CUs' LDS at that moment.
constexpr int TNG_RED_ITEMS = 64;   // items per 256-thread workgroup
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, item = wave * 16 + (lane & 15), share = lane >> 4;
  const long i4 = (long)((int)blockIdx.x - P.rbegin) * TNG_RED_ITEMS + item;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool is_w = i4 < NK4, is_b = !is_w && P.dbias && i4 < NK4 + P.N;
  if (is_w) {
    const float* src = ppart + i4 * 4;
    int z = share;
    for (; z + 12 < P.zs; z += 16) {   // four slabs of this share in flight
      const float4 v0 = *reinterpret_cast<const float4*>(src + (long)z * NK), v1 = *reinterpret_cast<const float4*>(src + (long)(z + 4) * NK);
      const float4 v2 = *reinterpret_cast<const float4*>(src + (long)(z + 8) * NK), v3 = *reinterpret_cast<const float4*>(src + (long)(z + 12) * NK);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y); s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; z < P.zs; z += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)z * NK);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (is_b) {
    const int n = (int)(i4 - NK4);
    for (int z = share; z < P.zs; z += 4) s.x += ppart[(long)P.zs * NK + (long)z * P.N + n];
  }
  // shares 0+1, 2+3, then the two pairs (the same order for every item: results do not depend on the launch shape)
  s.x += __shfl_xor(s.x, 16, 64); s.y += __shfl_xor(s.y, 16, 64); s.z += __shfl_xor(s.z, 16, 64); s.w += __shfl_xor(s.w, 16, 64);
  s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64); s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
  if (share) return;
  if (is_w) {
    const long i = i4 * 4;
    const int n = (int)(i / P.K), k = (int)(i - (long)n * P.K);
    float* o = P.Out + P.out_off(n, k);
    o[0] += s.x; o[P.sok] += s.y; o[2 * P.sok] += s.z; o[3 * P.sok] += s.w;
  } else if (is_b) {
    const int n = (int)(i4 - NK4);
    if (P.bias_atomic()) atomicAdd(P.dbias + P.bias_col(n), s.x);
    else P.dbias[P.bias_col(n)] += s.x;
  }
}

// ------------------------------------------------------------------------------------------------
// Streaming variant for LONG contractions with a SMALL output (round 5): the stage-0/1 Linear gradients (64 k - 512 k rows, 96..768 x 96..768
// outputs), the patch-merging reduction and the decoder-1 transpose-conv gradient (512 k coarse voxels x 16 problems of 192 x 96).  With 96x96 output tiles
// those problems have 1-32 tiles: each tile's workgroups re-read the B panel once per N tile and fetch 192-byte slices of 768-byte A rows, and the
// launch ran at 1.3-2.0 TB/s of HBM traffic (2.35 ms for the transpose-conv gradient's 3.15 GB).  They are HBM-bound by two orders of magnitude
// (48-96 FLOP per operand byte at full-output blocks, MFMA busy < 10 %), so here a workgroup owns a ROW RANGE and a whole output block of up to
// 384 x 192 (8 waves x up to 36 accumulator tiles): every operand byte is fetched once, in full rows, through a ring of 32-row chunks with
// ST-1 chunks (54-108 KB per CU) in flight; the partial blocks go to the launch workspace and the grouped reduce launch sums them (the
// same [split][N][K] layout as the split path above).  A workgroup's rows lie inside one sample, so the stochastic-depth factor is one multiply
// at the end.
//   NA / NB: 96-column sub-tiles of the A / B operand per stage; WN x WK = 8: wave grid over the (6 NA) x (6 NB) accumulator tiles.
// ------------------------------------------------------------------------------------------------
namespace tns {
constexpr int CH = 32, SUB = CH * tng::RS;   // one sub-tile: 32 rows x 96 columns = 6 x 1 KB DMA pieces
}
template <int NA, int NB, int WN, int WK, int ST, bool UP>   // UP: the launch's A operands are pixel-shuffled views (ConvTranspose3d backward)
__global__ __launch_bounds__(512) void gemm_tn_stream_kernel(tng::Args ga) {
  using namespace tng;
  constexpr int NS = NA + NB, PIECES = NS * 6, NPW = (PIECES + 7) / 8, STG = NPW * 8 * 1024;   // (stage stride: whole pieces per wave; the surplus is a trash area)
  constexpr int TN = 6 * NA / WN, TK = 6 * NB / WK;
  static_assert(WN * WK == 8 && (6 * NA) % WN == 0 && (6 * NB) % WK == 0, "wave grid");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave / WK, wk = wave - wn * WK, g = lane >> 4, li = lane & 15;
  const int bid = (int)blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAXP; ++i)
    if (bid >= ga.wbegin[i]) pi = i;
  const Prob& P = ga.p[pi];
  const int lw = bid - ga.wbegin[pi];
  const int N = P.N, K = P.K;
  const int nblk = (N + 96 * NA - 1) / (96 * NA), kblk = (K + 96 * NB - 1) / (96 * NB);
  const int z = lw / (nblk * kblk), t = lw - z * (nblk * kblk);
  const int nb = t / kblk, kb = t - nb * kblk;
  const int n0 = nb * 96 * NA, k0 = kb * 96 * NB;
  // rows: split sj of sample sb takes the 32-row chunks sj, sj + sub, sj + 2 sub, ... of the sample: the workgroups of a problem stream ADJACENT chunks at
  // any moment (a grid-stride loop) rather than row ranges a fixed multi-MB stride apart
  const int sb = z / P.sub, sj = z - sb * P.sub, psub = P.sub;
  const int seglen = P.rps;
  const long segbase = (long)sb * P.rps;
  const int tc = (seglen + tns::CH - 1) / tns::CH;
  const int nc = tc > sj ? (tc - sj + psub - 1) / psub : 0;
  const long lda = P.lda, ldb = P.ldb;
  const int up_k = P.up_k(), up_v = P.up_v();
  const float up_rcp = UP ? 1.0f / (float)up_v : 0.f;

  // DMA piece i of this wave: 1-KB piece q = wave + 8 i of the stage image (sub-tile q / 6, piece q % 6); q >= PIECES: zero page -> trash.
  // The issue path is one pointer bump per piece and chunk (first version: ~150 instructions of 64-bit address arithmetic and branches per piece -- 1.2 us
  // per chunk and wave, the whole HBM budget of a 30-KB chunk; 2.5 TB/s).  pptr: source of the piece in the split's first chunk; pstep: bytes per chunk step.
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_tng;
  asm volatile("" : "+v"(zpage));
  unsigned long long pptr[NPW];
  int prow[NPW];
  bool pok[NPW], pisA[NPW];
  const long stepA = (long)psub * tns::CH * lda * 2, stepB = (long)psub * tns::CH * ldb * 2;
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = wave + 8 * i, sub = q / 6, off = 1024 * (q - sub * 6) + 16 * lane;
    const int row = off / RS, u = (off - row * RS) >> 4;
    prow[i] = row;
    pisA[i] = sub < NA;
    const int c = (u ^ (((row >> 2) & 1) << 1)) * 8 + (pisA[i] ? sub : sub - NA) * 96;
    const int col = pisA[i] ? n0 + c : k0 + c;
    pok[i] = q < PIECES && (pisA[i] ? col < N : col < K);
    const long r = segbase + (long)sj * tns::CH + row;
    if (pisA[i]) {
      if (UP) {   // the column part of the folded tap row: column n = (tx, co) sits at fine row arow + tx, channel co (arow: per chunk, below)
        long cofs = col;
        if (P.ncol2) { const int ni = P.ncol2 & 0xffff, q1 = col / ni; cofs = (long)q1 * lda + (col - q1 * ni); }
        pptr[i] = (unsigned long long)(P.A + cofs);
      } else pptr[i] = (unsigned long long)(P.A + r * lda + col);
    } else pptr[i] = (unsigned long long)(P.B + r * ldb + col);
  }
  int ichunk = 0, islot = 0;
  auto issue = [&]() {
    const int r0 = (sj + ichunk * psub) * tns::CH;
    const bool full = r0 + tns::CH <= seglen;   // (wave-uniform) every row of the chunk exists
    char* slot = smem + islot * STG;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      unsigned long long src = pptr[i];
      if (UP && pisA[i]) {   // pixel-shuffled view: coarse voxel m -> the fine row of this problem's tap row
        const unsigned vv = (unsigned)up_v, kk = (unsigned)up_k, m = (unsigned)(segbase + r0 + prow[i]);
        // m < 2^24: floor(m / vv) from the float reciprocal, corrected by one step either way
        auto divv = [&](unsigned a, unsigned& qo, unsigned& ro) {
          unsigned qq = (unsigned)((float)a * up_rcp);
          int rr = (int)(a - qq * vv);
          if (rr < 0) { --qq; rr += (int)vv; }
          if (rr >= (int)vv) { ++qq; rr -= (int)vv; }
          qo = qq; ro = (unsigned)rr;
        };
        unsigned q1, x, q2, y, bb, zq;
        divv(m, q1, x); divv(q1, q2, y); divv(q2, bb, zq);
        const long Vf = (long)vv * kk;
        const long arow = (((long)bb * Vf + zq * kk) * Vf + y * kk) * Vf + x * kk;
        src += (unsigned long long)(arow * lda * 2);
      } else pptr[i] += (unsigned long long)(pisA[i] ? stepA : stepB);
      const bool ok = pok[i] && (full || r0 + prow[i] < seglen);
      src = ok ? src : zpage;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(slot + (wave + 8 * i) * 1024), 16, 0, 0);
    }
    ++ichunk;
    if (++islot == ST) islot = 0;
  };

  f32x4 acc[TN][TK], bacc[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = P.dbias != nullptr && kb == 0 && wk == 0;
  Frag<bf16_t> ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones.v[j] = (short)0x3F80;

  // fragment addresses (raw transpose reads, common.hpp): per-lane offsets inside a stage
  unsigned aofs[TN], bofs[TK];
#pragma unroll
  for (int a = 0; a < TN; ++a) { const int tn = wn * TN + a; aofs[a] = (unsigned)((tn / 6) * tns::SUB) + tng_frag_ofs(0, (tn % 6) * 16, lane); }
#pragma unroll
  for (int b = 0; b < TK; ++b) { const int tk = wk * TK + b; bofs[b] = (unsigned)((NA + tk / 6) * tns::SUB) + tng_frag_ofs(0, (tk % 6) * 16, lane); }
  const unsigned smem_u = lds_addr_u(smem);
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (s < nc) issue();
  int cslot = 0;
  for (int c = 0; c < nc; ++c) {
    if (nc - 1 - c >= ST - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // chunk c is visible to everyone; everyone is done reading chunk c-1 (whose slot the next issue refills)
    if (c + ST - 1 < nc) issue();
    const unsigned st_u = smem_u + (unsigned)(cslot * STG);
    if (++cslot == ST) cslot = 0;
    TrFrag fa[TN], fb[TK];
#pragma unroll
    for (int b = 0; b < TK; ++b) tr_read_raw<16 * RS>(fb[b], st_u + bofs[b]);
#pragma unroll
    for (int a = 0; a < TN; ++a) tr_read_raw<16 * RS>(fa[a], st_u + aofs[a]);
    tr_wait();
#pragma unroll
    for (int b = 0; b < TK; ++b) tr_pin(fb[b]);
#pragma unroll
    for (int a = 0; a < TN; ++a) tr_pin(fa[a]);
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const Frag<bf16_t> af = tr_frag(fa[a]);
#pragma unroll
      for (int b = 0; b < TK; ++b) mma(acc[a][b], af, tr_frag(fb[b]));
      if (want_bias) mma(bacc[a], af, ones);
    }
  }
  // acc[a][b][r]: row n = n0 + (wn TN + a) 16 + 4 g + r, column k = k0 + (wk TK + b) 16 + li  ->  this split's partial slab
  const float sc = P.rowscale ? P.rowscale[sb] : 1.f;
  float* const ppart = ga.ws + P.partoff;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * TN + a) * 16 + 4 * g + r, k = k0 + (wk * TK + b) * 16 + li;
        if (n < N && k < K) ppart[((long)z * N + n) * K + k] = sc * acc[a][b][r];
      }
  if (want_bias && li == 0) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * TN + a) * 16 + 4 * g + r;
        if (n < N) ppart[(long)P.zs * N * K + (long)z * N + n] = sc * bacc[a][r];
      }
  }
}

namespace tns {
struct Cfg { int NA, NB; };
// output-block shapes on offer (columns of A x columns of B per workgroup): {192,96} {384,96} {96,384} {384,192} {192,384} {192,192}
static const Cfg kCfg[] = {{2, 1}, {4, 1}, {1, 4}, {4, 2}, {2, 4}, {2, 2}};
constexpr int NCFG = 6;
// the block shape a problem takes: the one that covers [N, K] with the fewest operand re-reads (blocks along N re-read B, blocks along K re-read A),
// then the least padding
static int pick_cfg(int N, int K) {
  int best = -1;
  double bestcost = 1e30;
  for (int c = 0; c < NCFG; ++c) {
    const int bn = 96 * kCfg[c].NA, bk = 96 * kCfg[c].NB;
    const int nblk = (N + bn - 1) / bn, kblk = (K + bk - 1) / bk;
    const double bytes = (double)N * kblk + (double)K * nblk;                    // operand columns fetched per row
    const double pad = (double)nblk * bn * kblk * bk / ((double)N * K);           // MFMA / DMA-issue work relative to the useful part
    const double cost = bytes * (1.0 + 0.02 * pad);
    if (cost < bestcost) { bestcost = cost; best = c; }
  }
  return best;
}
template <int NA, int NB, int WN, int WK, int ST, bool UP> static int launch(const tng::Args& ga, int wgs, hipStream_t st) {
  constexpr int lds = ST * ((((NA + NB) * 6 + 7) / 8) * 8 * 1024);
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_stream_kernel<NA, NB, WN, WK, ST, UP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  hipLaunchKernelGGL((gemm_tn_stream_kernel<NA, NB, WN, WK, ST, UP>), dim3(wgs), dim3(512), lds, st, ga);
  NMH_CHECK_LAUNCH();
  return 0;
}
template <bool UP> static int launch_cfg(int c, const tng::Args& ga, int wgs, hipStream_t st) {
  switch (c) {
    // ring of ST = 2 stages (48-80 KB of LDS): the probe (tools/probes/ldsdma_stream_probe.hip) reads 6.2-6.3 TB/s with any ring of 32-KB stages, and the launches
    // run on the side stream under the input-gradient chain, whose GEMM workgroups need 40-80 KB of the same CUs' LDS (with 4 stages the chain stood still)
    case 0: return launch<2, 1, 4, 2, 2, UP>(ga, wgs, st);
    case 1: return launch<4, 1, 8, 1, 2, UP>(ga, wgs, st);
    case 2: return launch<1, 4, 1, 8, 2, UP>(ga, wgs, st);
    case 3: return launch<4, 2, 4, 2, 2, UP>(ga, wgs, st);
    case 4: return launch<2, 4, 2, 4, 2, UP>(ga, wgs, st);
    case 5: return launch<2, 2, 4, 2, 2, UP>(ga, wgs, st);
  }
  return -2;
}
}  // namespace tns

// the streaming launches of a grouped call: problems -> (block shape) groups -> one launch each + one grouped reduce.  Returns the number of problems taken
// (they are removed from `rest`), or a negative error.
static int tn_stream_launches(const TnProblemHost* probs, int nprob, std::vector<int>& rest, float* ws, long ws_floats, hipStream_t st, bool foreground) {
  using namespace tng;
  const char* e_on = getenv("NMH_TNS");   // (read per call: tools/bench_tns.py and the parity tests switch it inside one process)
  const int on = e_on ? atoi(e_on) : 1;
  static const long min_m = getenv("NMH_TNS_MINM") ? atol(getenv("NMH_TNS_MINM")) : 32768;
  const char* e_ratio = getenv("NMH_TNS_RATIO");   // (per call, like NMH_TNS: the parity tests force the streaming path at small sizes with 0)
  const double ratio = e_ratio ? atof(e_ratio) : 6.0;
  // workgroups per launch.  Background (the side stream under the input-gradient chain): 192 -- a persistent 512-thread workgroup per CU leaves the chain's
  // small kernels waiting for a CU, a quarter of the chip stays theirs; foreground (the last flushes of a backward pass): one per CU
  static const int target_b = getenv("NMH_TNS_TARGET") ? atoi(getenv("NMH_TNS_TARGET")) : 192;
  static const int target_f = getenv("NMH_TNS_TARGET_FG") ? atoi(getenv("NMH_TNS_TARGET_FG")) : 256;
  const int target_bg = foreground ? target_f : target_b;
  rest.clear();
  std::vector<int> take[2 * tns::NCFG];   // [block shape][plain | pixel-shuffled A]
  for (int i = 0; i < nprob; ++i) {
    const TnProblemHost& h = probs[i];
    bool ok = on && ws && h.M >= min_m && h.rows_per_sample >= 2048 && h.M % h.rows_per_sample == 0 && (long)h.N * h.K <= 768L * 768 &&
              h.K % 8 == 0 && h.lda % 8 == 0 && h.ldb % 8 == 0 && h.N % 8 == 0 && h.A && h.B && h.dW;
    if (ok && h.n_inner > 0 && (h.N % h.n_inner || h.n_inner % 8 || h.n_inner > 65535 || h.stride_n2 < 0 || h.stride_n2 > 32767)) ok = false;
    if (ok && h.up_k > 0 && (h.up_v <= 0 || h.up_k > 255 || h.up_v > 65535 || h.rows_per_sample != (long)h.up_v * h.up_v * h.up_v)) ok = false;
    if (ok && (h.lda >= (1L << 31) || h.ldb >= (1L << 31) || h.ldo >= (1L << 31) || h.stride_k >= (1L << 31))) ok = false;
    if (ok && h.up_k > 0 && h.M >= (1L << 24)) ok = false;   // (the kernel's float-reciprocal row decomposition)
    int cfg = -1;
    if (ok) {
      // worth it only while the partial blocks (one per workgroup, written and read back by the reduce) stay small against the operands: the stage-1 problems
      // (64 k rows, 768 x 192 outputs) measured 302 us here against 200 us with the tile kernels, stage 0 and the transpose conv (512 k rows) 1.2-2x faster
      // (Also measured and dropped: 384 x 192 blocks for the stage-2 problems -- 8000 rows as one range split two ways.  1353 us for 18 blocks against
      //  1129 us with the 96 x 96 tiles, +1.4 ms in the step: one 8-wave workgroup per CU with a two-stage ring cannot hide the chunk latency that three
      //  co-resident tile workgroups hide for each other.  profiles/r5a_ab_tns_mid_class.txt)
      cfg = tns::pick_cfg(h.N, h.K);
      const int bn = 96 * tns::kCfg[cfg].NA, bk = 96 * tns::kCfg[cfg].NB;
      const int nblk = (h.N + bn - 1) / bn, kblk = (h.K + bk - 1) / bk;
      const double operand = (double)h.M * ((double)h.N * kblk + (double)h.K * nblk) * 2.0 / (nblk * kblk);
      const double partial = (double)target_bg * std::min(h.N, bn) * std::min(h.K, bk) * 4.0;
      if (operand < ratio * partial) ok = false;
    }
    if (ok) { take[2 * cfg + (h.up_k > 0 ? 1 : 0)].push_back(i); continue; }
    rest.push_back(i);
  }
  int ntaken = 0;
  auto fill = [](tng::Prob& p, const TnProblemHost& h) {
    p.A = (const bf16_t*)h.A; p.B = (const bf16_t*)h.B; p.Out = h.dW; p.dbias = h.dbias; p.rowscale = h.rowscale;
    p.lda = (int)h.lda; p.ldb = (int)h.ldb; p.N = h.N; p.K = h.K;
    p.son = (int)h.ldo; p.sok = h.stride_k > 0 ? (int)h.stride_k : 1;
    p.ncol2 = h.n_inner > 0 ? (h.n_inner | ((int)h.stride_n2 << 16)) : 0;
    p.upflags = (h.up_k & 0xff) | ((h.bias_atomic || h.n_inner > 0 ? 1 : 0) << 8) | (h.up_k > 0 ? (h.up_v << 16) : 0);
  };
  for (int cu = 0; cu < 2 * tns::NCFG; ++cu) {
    const int c = cu >> 1;
    std::vector<int>& idx = take[cu];
    for (size_t base = 0; base < idx.size(); base += MAXP) {
      const int np = (int)std::min((size_t)MAXP, idx.size() - base);
      Args ga{};
      ga.nprob = np; ga.ws = ws; ga.xcd = 0;
      for (int i = 0; i < MAXP; ++i) ga.wbegin[i] = 0x7fffffff;
      const int bn = 96 * tns::kCfg[c].NA, bk = 96 * tns::kCfg[c].NB;
      double bytes_tot = 0.0;
      std::vector<double> bytes(np);
      for (int i = 0; i < np; ++i) {
        const TnProblemHost& h = probs[idx[base + i]];
        const int nblk = (h.N + bn - 1) / bn, kblk = (h.K + bk - 1) / bk;
        bytes[i] = (double)h.M * ((double)h.N * kblk + (double)h.K * nblk);
        bytes_tot += bytes[i];
      }
      // every launch is followed by its own reduce on the same stream, so each launch has the whole workspace
      long wsoff = 0;
      int w = 0, rb = 0;
      bool fits = true;
      for (int i = 0; i < np && fits; ++i) {
        const TnProblemHost& h = probs[idx[base + i]];
        Prob& p = ga.p[i];
        fill(p, h);
        p.rps = h.rows_per_sample; p.nsamp = (int)(h.M / h.rows_per_sample);
        const int nblk = (h.N + bn - 1) / bn, kblk = (h.K + bk - 1) / bk;
        // this problem's share of the launch's workgroups -> row splits per sample (>= 1024 rows each; at least 2 splits in all: the reduce path)
        const double share = target_bg * bytes[i] / bytes_tot / (double)(nblk * kblk);
        int sub = (int)(share / p.nsamp + 0.5);
        sub = std::max(1, std::min(sub, std::max(1, p.rps / 1024)));
        if (p.nsamp * sub < 2) sub = 2;
        long need = (long)p.nsamp * sub * p.N * (p.K + 1);
        while (sub > 1 && p.nsamp * (sub - 1) >= 2 && (wsoff + need > ws_floats || wsoff + need >= (1L << 31))) { --sub; need = (long)p.nsamp * sub * p.N * (p.K + 1); }
        if (wsoff + need > ws_floats || wsoff + need >= (1L << 31)) { fits = false; break; }
        p.sub = sub; p.zs = p.nsamp * sub;
        p.partoff = (int)wsoff;
        wsoff += (need + 3) / 4 * 4;
        ga.wbegin[i] = w;
        w += p.zs * nblk * kblk;
        p.rbegin = rb;
        rb += (int)(((long)p.N * p.K / 4 + (p.dbias ? p.N : 0) + TNG_RED_ITEMS - 1) / TNG_RED_ITEMS);
      }
      if (!fits) {   // workspace too small for this launch: its problems take the tile kernels
        for (int i = 0; i < np; ++i) rest.push_back(idx[base + i]);
        continue;
      }
      if (int e = (cu & 1) ? tns::launch_cfg<true>(c, ga, w, st) : tns::launch_cfg<false>(c, ga, w, st)) return e;
      hipLaunchKernelGGL(gemm_tn_grouped_reduce_kernel, dim3(rb), dim3(256), 0, st, ga);
      NMH_CHECK_LAUNCH();
      ntaken += np;
    }
  }
  return ntaken;
}

// host side: chunk the problem list into launches of <= 16 problems, decide the contraction splits per launch
static int k_gemm_tn_grouped_tiles(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st, bool foreground);
int k_gemm_tn_grouped(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st, bool foreground) {
  // long contractions with small outputs take the streaming kernel (each workgroup: a row range x a whole output block), everything else the tile kernels
  std::vector<int> rest;
  const int taken = tn_stream_launches(probs, nprob, rest, ws, ws_floats, st, foreground);
  if (taken < 0) return taken;
  if (taken == 0) return k_gemm_tn_grouped_tiles(probs, nprob, ws, ws_floats, st, foreground);
  if (rest.empty()) return 0;
  std::vector<TnProblemHost> r(rest.size());
  for (size_t i = 0; i < rest.size(); ++i) r[i] = probs[rest[i]];
  // (the two paths share the workspace: the tile launches follow the streaming launches and their reduce on the same stream)
  return k_gemm_tn_grouped_tiles(r.data(), (int)r.size(), ws, ws_floats, st, foreground);
}
static int k_gemm_tn_grouped_tiles(const TnProblemHost* probs, int nprob, float* ws, long ws_floats, hipStream_t st, bool foreground) {
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
    //  Round 5: 0 (off).  With the ring actually pipelined (raw transpose reads) the 96x96 tiles -- 48 KB of LDS, shorter-lived workgroups that let the
    //  input-gradient chain's kernels onto the CUs -- measure 46.49-46.57 ms against 46.67-46.89 with the large tiles (same box, two alternating runs))
    const int big_min = getenv("NMH_TNG_BIG") ? atoi(getenv("NMH_TNG_BIG")) : 0;
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
      if (p.zs > 1) rb += (int)(((long)p.N * p.K / 4 + (p.dbias ? p.N : 0) + TNG_RED_ITEMS - 1) / TNG_RED_ITEMS);
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
