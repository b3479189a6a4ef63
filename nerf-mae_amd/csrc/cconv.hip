// Composed forward of decoder1's first two linear ops (reference: unetr_block.py:151-158 ConvTranspose3d(96 -> 48, kernel = stride = 4)
// followed by unetr_block.py:35-44 Conv3d(48 -> 48, 3x3x3, pad 1), swin_mae3d.py:1246-1257), bf16, gfx950.
//
// u = ConvT(x) is piecewise linear in the coarse grid: u[4j + a] = Wt[a]^T x[j] + bt for phase a in {0..3}^3.  The 3x3x3 convolution
// of u at fine voxel 4j + a only touches the coarse cells j + n, n in N(a) = Nz x Ny x Nx with N(0) = {-1, 0}, N(1) = N(2) = {0},
// N(3) = {0, +1} per axis (3.375 cells on average instead of 27 fine taps):
//     y1[4j + a][c] = sum_{n in N(a)} sum_ci x[j + n][ci] Wc[a][n][ci][c]  +  border term,
//     Wc[a][n][ci][c] = sum_{d in {-1,0,1}^3 : floor((a + d) / 4) = n} sum_co Wt[ci][co][(a + d) mod 4] W1[c][co][d].
// 216 (a, n) blocks of 96 x 48 weights (2 MB) replace 27 x 48 x 48 per fine voxel: 31 instead of 133 kFLOP per voxel.  The transpose
// conv's bias enters as sum_{d inside the volume} sum_co bt[co] W1[c][co][d]: constant per channel in the interior -- removed by the
// affine-free InstanceNorm that follows, like conv1's own bias -- and different only on the border shell of the volume,
// where a (class, channel) table of DIFFERENCES to the interior value is added.  Zero padding of u = zero coarse cells outside the grid.
//
// Kernel: a persistent 512-thread workgroup per CU walks blocks of 4x8x8 coarse cells (16x32x32 fine voxels):
//   * the 6x10x10x96 coarse halo sits in LDS (117 KB; 192-byte cells, line stride 1952 B: conflict-free ds_read_b128 for the 2x8-cell MFMA
//     column tiles under the b128 service groups of the part);
//   * wave w owns 32 cells (z = w / 2, four y-lines) = two 16-cell column tiles; products are formed transposed (weights as the A
//     operand, rows permuted) so that a lane leaves with 12 consecutive channels of one fine voxel: a 16-byte + an 8-byte store;
//   * the 64 phases are walked as 16 (a_z, a_y) groups with the four a_x phases in flight together (96 accumulators): one x fragment
//     then feeds up to four phases;
//   * the 2 MB of composed weights stream through a 2 x 18 KB LDS ring by LDS-DMA in 108 chunks of 18 fragments (one (group, n_z, n_y,
//     k-step): the six (a_x, n_x) blocks x three channel tiles), fragment-ordered by the pack kernel below;
//   * InstanceNorm statistics of the (bf16-rounded) outputs are folded per group (DPP row sums -> LDS -> fp64 atomics per sample), as in
//     conv48.hip.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>

namespace cc {
constexpr int BZ = 4, BY = 8, BX = 8, HZ = BZ + 2, HY = BY + 2, HX = BX + 2;
constexpr int CELL = 192, LINE = HX * CELL + 32, PLANE = HY * LINE, HALO = HZ * PLANE;
constexpr int WCHUNK = 18 * 1024, NCHUNK = 108, WHALF = 9 * 1024, NHALF = 2 * NCHUNK;   // ring: 4 slots of half a chunk (9 fragments)
constexpr int OFF_RING = HALO, OFF_SACC = OFF_RING + 2 * WCHUNK, OFF_DELTA = OFF_SACC + 2 * 96 * 4, OFF_MHAT = OFF_DELTA + 27 * 48 * 4, LDS_BYTES = OFF_MHAT + 48 * 4;
static_assert(LDS_BYTES <= 163840, "LDS budget");
constexpr int HCH = HZ * HY * HX * 12;          // 16-byte chunks of the halo (7200)
constexpr int HREG = (HCH + 511) / 512;         // 15
// the six (a_x, n_x) blocks of a chunk
__device__ constexpr int BLK_AX[6] = {0, 0, 1, 2, 3, 3};
__device__ constexpr int BLK_NX[6] = {0, 1, 1, 1, 1, 2};   // index into xf[]: n_x + 1
}  // namespace cc

struct CConvArgs {
  const bf16_t* X; const bf16_t* Wcp; const float* delta; bf16_t* Y; double* stats_acc;
  int B, v, nbz, nby, nbx; long total;
  // centered variant (CZ): mhat [B][48] = the per-(sample, channel) mean of the output, known BEFORE the launch (k_cconv_mean: the output is linear in x);
  // the kernel stores z = lrelu(y1 - mhat) and accumulates the statistics of t = y1 - mhat
  const float* mhat; float slope;
};

// neighbour offsets of phase component a: count and first offset
__host__ __device__ __forceinline__ int cc_ncnt(int a) { return (a == 0 || a == 3) ? 2 : 1; }
__host__ __device__ __forceinline__ int cc_nfirst(int a) { return a == 0 ? -1 : 0; }

// DBG (diagnostic builds, NMH_CCONV_DBG): 1 = no output stores, 2 = no weight DMA / waits (stale weights), 4 = no MFMAs, 8 = no epilogue at all.  0 = product.
template <int DBG, bool CZ = false>
__global__ __launch_bounds__(512) void cconv_fwd_kernel(CConvArgs a) {
  using namespace cc;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + OFF_RING;
  float* const sacc = reinterpret_cast<float*>(smem + OFF_SACC);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int V = a.v, F = 4 * a.v;

  if (tid < 192) sacc[tid] = 0.f;
  for (int i = tid; i < 27 * 48; i += 512) reinterpret_cast<float*>(smem + OFF_DELTA)[i] = a.delta[i];
  // the line pads are never read (a fragment read stays inside its cell); nothing to clear

  // XCD-contiguous block ranges (as conv48.hip): workgroup w runs on XCD w % 8
  const int nx8 = 8, xcd = blockIdx.x % nx8, jb = blockIdx.x / nx8, jstride = gridDim.x / nx8;
  const long per = (a.total + nx8 - 1) / nx8;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < a.total) ? tbeg + per : a.total;

  // half-chunk h (9 lane-linear 1-KB images: blocks 3 (h & 1) .. +2 of chunk h / 2) -> ring slot: wave w takes image w, wave 0 also image 8
  auto w_dma = [&](int h, int slot) {
    if (DBG & 2) return;
    const char* src = reinterpret_cast<const char*>(a.Wcp) + (long)h * WHALF;
    char* dst = wbuf + slot * WHALF;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wave * 1024 + lv * 16),
                                     (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * 1024 + lv * 16),
                                       (__attribute__((address_space(3))) void*)(dst + 8 * 1024), 16, 0, 0);
  };
  auto row_sum = [](float v) -> float {  // inclusive scan over the 16-lane row: lane 15 ends up with the row total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
  };
  int st_b = -1, scur = 0;
  auto stats_flush = [&]() {
    if (st_b >= 0 && tid < 96) {
      const float v = sacc[scur * 96 + tid];
      sacc[scur * 96 + tid] = 0.f;
      atomicAdd(a.stats_acc + (long)st_b * 96 + tid, (double)v);
    }
  };

  // per-lane geometry inside the block: wave = (z_l, y half), column tile i = two y-lines, lane li = (y line, x)
  const int z_l = wave >> 1, y_l = (wave & 1) * 4, ly = li >> 3, lx = li & 7;
  const int xbase = (z_l + 1) * PLANE + (y_l + 1 + ly) * LINE + (lx + 1) * CELL + g * 16;   // halo offset of this lane's cell, column tile 0, k-group g

  long t = tbeg + jb;
  if (t >= tend) return;
  int cz_b = -1;
  w_dma(0, 0); w_dma(1, 1); w_dma(2, 2);   // three half-chunks ahead
  for (; t < tend; t += jstride) {
    // ---- block origin
    const unsigned tu = (unsigned)t;
    unsigned r1 = tu / (unsigned)a.nbx; const int xb = (int)(tu - r1 * (unsigned)a.nbx);
    unsigned r2 = r1 / (unsigned)a.nby; const int yb = (int)(r1 - r2 * (unsigned)a.nby);
    unsigned r3 = r2 / (unsigned)a.nbz; const int zb = (int)(r2 - r3 * (unsigned)a.nbz);
    const int b = __builtin_amdgcn_readfirstlane((int)r3);
    const int z0 = __builtin_amdgcn_readfirstlane(zb * BZ), y0 = __builtin_amdgcn_readfirstlane(yb * BY), x0 = __builtin_amdgcn_readfirstlane(xb * BX);
    // ---- coarse halo -> LDS (cells outside the grid are zero: the zero padding of the fine convolution)
    {
      // all requests first, then all LDS stores: in front of an LDS store the backend waits for every outstanding vector-memory operation
      // (it cannot tell the store from the LDS-DMA weight prefetch in flight) -- one drain for the 15 loads instead of one per load
      const bf16_t* Xb = a.X + (long)b * V * V * V * 96;
      uint4 hv[HREG];
#pragma unroll
      for (int i = 0; i < HREG; ++i) {
        const int cid = tid + 512 * i;
        const int cell = cid / 12, c12 = cid - cell * 12;
        const int hz = cell / (HY * HX), rem = cell - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
        hv[i] = make_uint4(0, 0, 0, 0);
        if (cid < HCH && (unsigned)z < (unsigned)V && (unsigned)y < (unsigned)V && (unsigned)x < (unsigned)V)
          hv[i] = *reinterpret_cast<const uint4*>(Xb + ((long)(z * V + y) * V + x) * 96 + c12 * 8);
      }
#pragma unroll
      for (int i = 0; i < HREG; ++i) {
        const int cid = tid + 512 * i;
        if (cid < HCH) {
          const int cell = cid / 12, c12 = cid - cell * 12;
          const int hz = cell / (HY * HX), rem = cell - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
          *reinterpret_cast<uint4*>(halo + hz * PLANE + hy * LINE + hx * CELL + c12 * 16) = hv[i];
        }
      }
      if constexpr (CZ) {   // minus the sample's predicted means: the accumulators of every group start from them (next to the halo stores: one drain for all)
        if (b != cz_b && tid < 48) reinterpret_cast<float*>(smem + OFF_MHAT)[tid] = -a.mhat[(long)b * 48 + tid];
        cz_b = b;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // the halo is complete before anyone reads it (the x fragments of a chunk are requested in front of its weight barrier)
    if (a.stats_acc && b != st_b) {
      stats_flush();
      if (st_b >= 0) scur ^= 1;
      st_b = b;
    }
    int ck = 0;
#pragma unroll 1
    for (int gi = 0; gi < 16; ++gi) {
      const int az = gi >> 2, ay = gi & 3;
      f32x4 acc[4][3][2];
      if constexpr (CZ) {   // rows 4 g + r of channel tile n <-> channel 12 g + 4 n + r: three raw 16-byte reads (a visible LDS read would drain the weight prefetch)
        f32x4 m0, m1, m2;
        const unsigned maddr = (unsigned)(OFF_MHAT + 12 * g * 4);
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(m0), "=&v"(m1), "=&v"(m2) : "v"(maddr) : "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int m = 0; m < 2; ++m) { acc[p][0][m] = m0; acc[p][1][m] = m1; acc[p][2][m] = m2; }
      } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[p][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int cz = cc_ncnt(az), cy = cc_ncnt(ay), fz = cc_nfirst(az), fy = cc_nfirst(ay);
#pragma unroll 1
      for (int iz = 0; iz < cz; ++iz)
#pragma unroll 1
        for (int iy = 0; iy < cy; ++iy) {
          const int noff = xbase + (fz + iz) * PLANE + (fy + iy) * LINE;
#pragma unroll 1
          for (int s = 0; s < 3; ++s, ++ck) {
            Frag<bf16_t> xf[3][2];
#pragma unroll
            for (int nx = 0; nx < 3; ++nx)
#pragma unroll
              for (int m = 0; m < 2; ++m) xf[nx][m].v = *reinterpret_cast<const bf16x8*>(halo + noff + (nx - 1) * CELL + m * 2 * LINE + s * 64);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int h = 2 * ck + half;
              // half-chunk h has landed once at most the pieces of the two younger half-chunks (and, in front of a group's first one, the 16
              // output stores of the previous group, which are younger still) are outstanding: vector-memory operations retire in order
              const bool after_epi = half == 0 && gi > 0 && iz == 0 && iy == 0 && s == 0;
              if (DBG & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              else if (wave == 0) {
                if (after_epi) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
              } else {
                if (after_epi) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
              }
              __builtin_amdgcn_s_barrier();     // everyone's pieces of h are visible; everyone is done with half-chunk h - 1, whose slot is refilled
              {
                const int hn = h + 3 >= NHALF ? h + 3 - NHALF : h + 3;
                w_dma(hn, (h + 3) & 3);
              }
              // all nine weight fragments of the half-chunk are requested before the first MFMA: left to itself the compiler reuses ONE fragment
              // register and serialises read -> wait -> 2 MFMAs (an LDS round trip per 34 MFMA cycles: measured 3x the MFMA time)
              const char* wsrc = wbuf + (h & 3) * WHALF + lane * 16;
              Frag<bf16_t> wf[9];
#pragma unroll
              for (int i = 0; i < 9; ++i) wf[i].v = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                const int k = 3 * half + kk;
#pragma unroll
                for (int n = 0; n < 3; ++n)
#pragma unroll
                  for (int m = 0; m < 2; ++m) {
                    if (DBG & 4) { asm volatile("" ::"v"(wf[kk * 3 + n].v), "v"(xf[BLK_NX[k]][m].v)); }
                    else mma(acc[BLK_AX[k]][n][m], wf[kk * 3 + n], xf[BLK_NX[k]][m]);
                  }
              }
            }
          }
        }
      if (DBG & 8) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("" ::"v"(acc[p][n][m]));
        continue;
      }
      // ---- epilogue of the group: lane (li, g) owns channels 12 g .. 12 g + 11 (rows 4 g + r of channel tile n <-> channel 12 g + 4 n + r)
      const int zf = 4 * (z0 + z_l) + az;
      const int kz = zf == 0 ? 0 : (zf == F - 1 ? 2 : 1);
      // InstanceNorm statistics: per-lane partial sums of the group's fp32 outputs (the bf16 rounding of the stored values moves mean and
      // variance over 4 M voxels by ~1e-6 relative), folded over the 16 voxel columns by DPP row sums
      float st1[3][4], st2[3][4];
#pragma unroll
      for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[n][r] = 0.f; st2[n][r] = 0.f; }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int yf = 4 * (y0 + y_l + 2 * m + ly) + ay;
        const int ky = yf == 0 ? 0 : (yf == F - 1 ? 2 : 1);
        bf16_t* const drow = a.Y + ((((long)b * F + zf) * F + yf) * F + 4 * (x0 + lx)) * 48 + 12 * g;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int xf_ = 4 * (x0 + lx) + p;
          const int kx = xf_ == 0 ? 0 : (xf_ == F - 1 ? 2 : 1);
          float vv[3][4];
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[n][r] = acc[p][n][m][r];
          const int cls = (kz * 3 + ky) * 3 + kx;
          if (cls != 13) {   // border shell: the transpose conv's bias reaches fewer taps than in the interior
            // raw ds_read + own wait: a compiler-visible LDS (or global) load here is preceded by `s_waitcnt vmcnt(0)` -- the weight prefetch is an
            // LDS-DMA in flight -- which drains the output stores of the previous voxel every time (measured: the epilogue then dominates the kernel)
            f32x4 d0, d1, d2;
            const unsigned daddr = (unsigned)(OFF_DELTA + (cls * 48 + 12 * g) * 4);
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(daddr) : "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) { vv[0][r] += d0[r]; vv[1][r] += d1[r]; vv[2][r] += d2[r]; }
          }
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[n][r] += vv[n][r]; st2[n][r] += vv[n][r] * vv[n][r]; }
          if constexpr (CZ) {   // z = lrelu(t), t = y1 - mhat (the statistics above are those of t)
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
              for (int r = 0; r < 4; ++r) vv[n][r] = __builtin_fmaxf(vv[n][r], a.slope * vv[n][r]);
          }
          unsigned w6[6];
#pragma unroll
          for (int q = 0; q < 6; ++q) w6[q] = pk_bf16(vv[q >> 1][(q & 1) * 2], vv[q >> 1][(q & 1) * 2 + 1]);
          bf16_t* dst = drow + p * 48;
          // (non-temporal stores measured slower: 1.95 vs 1.48 ms at 8 grids -- the 24-byte pieces of a voxel row no longer merge in L2)
          if (!(DBG & 1)) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(w6[0], w6[1], w6[2], w6[3]);
            *reinterpret_cast<uint2*>(dst + 8) = make_uint2(w6[4], w6[5]);
          } else asm volatile("" ::"v"(w6[0]), "v"(w6[1]), "v"(w6[2]), "v"(w6[3]), "v"(w6[4]), "v"(w6[5]));
        }
      }
      if (a.stats_acc) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a1 = row_sum(st1[n][r]), a2 = row_sum(st2[n][r]);
            if (li == 15) {
              // raw ds_add_f32: behind a compiler-visible LDS atomic the backend puts `s_waitcnt vmcnt(0)` (an LDS-DMA weight prefetch is in flight and
              // it cannot prove the two LDS regions apart) -- which would drain the 16 output stores just issued, once per group
              const unsigned addr = (unsigned)(OFF_SACC + (scur * 96 + (12 * g + 4 * n + r) * 2) * 4);
              asm volatile("ds_add_f32 %0, %1\n\tds_add_f32 %0, %2 offset:4" ::"v"(addr), "v"(a1), "v"(a2) : "memory");
            }
          }
      }
    }
    __syncthreads();   // everyone is done with the halo (and the statistics slot) before the next block overwrites it
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing weight prefetch
  __syncthreads();
  if (a.stats_acc) stats_flush();
}

// ------------------------------------------------------------------------------------------------
// composed weights in fragment order + the border-bias difference table, from the fp32 master parameters:
//   Wt [96][48][4][4][4] (ConvTranspose3d weight), W1 [48][48][3][3][3] (Conv3d weight), bt [48] (ConvTranspose3d bias)
//   -> Wcp bf16 [108 chunks][18 fragments][64 lanes][8], delta fp32 [27][48]
// Two launches (on the side stream under the encoder): (1) both weights transposed to co-contiguous rows WtT [64 ph][96 ci][48 co],
// W1T [27 d][48 c][48 co] (+ the border table); (2) one thread per packed element: up to 27 taps x 48-term dot products of contiguous rows.
// ------------------------------------------------------------------------------------------------
constexpr long CC_WTT = 64L * 96 * 48, CC_W1T = 27L * 48 * 48;

__global__ __launch_bounds__(256) void cconv_tr_kernel(const float* __restrict__ Wt, const float* __restrict__ W1, float* __restrict__ ws) {
  // one workgroup per source row block: contiguous reads, 192-byte contiguous writes (an LDS tile in between)
  __shared__ float tile[64 * 49];
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk < 96) {                          // WtT[ph][ci][co] = Wt[ci][co][ph], ci = blk: 48 x 64 source floats
    const float* src = Wt + (long)blk * 48 * 64;
    for (int i = tid; i < 48 * 64; i += 256) { const int co = i >> 6, ph = i & 63; tile[ph * 49 + co] = src[i]; }
    __syncthreads();
    for (int i = tid; i < 64 * 48; i += 256) { const int ph = i / 48, co = i - ph * 48; ws[((long)ph * 96 + blk) * 48 + co] = tile[ph * 49 + co]; }
  } else if (blk < 96 + 48) {              // W1T[d][c][co] = W1[c][co][d], c = blk - 96: 48 x 27 source floats
    const int c = blk - 96;
    const float* src = W1 + (long)c * 48 * 27;
    for (int i = tid; i < 48 * 27; i += 256) { const int co = i / 27, d = i - co * 27; tile[d * 49 + co] = src[i]; }
    __syncthreads();
    for (int i = tid; i < 27 * 48; i += 256) { const int d = i / 48, co = i - d * 48; ws[CC_WTT + ((long)d * 48 + c) * 48 + co] = tile[d * 49 + co]; }
  }
}

// one workgroup per fragment (16 channels x 32 input channels = 512 elements, one per thread); per contributing tap the 32 WtT rows and the
// 16 W1T rows are staged in LDS (coalesced 16-byte loads) and every thread forms its 48-term dot product from there
__global__ __launch_bounds__(512) void cconv_pack_kernel(const float* __restrict__ ws, const float* __restrict__ bt, bf16_t* __restrict__ Wcp, float* __restrict__ delta, float* __restrict__ Mtab) {
  constexpr int RSF = 52;                                  // padded row (floats)
  __shared__ __attribute__((aligned(16))) float swt[32 * RSF];
  __shared__ __attribute__((aligned(16))) float sw1[16 * RSF];
  const float* WtT = ws;
  const float* W1T = ws + CC_WTT;
  const int blk = blockIdx.x, tid = threadIdx.x;
  if (blk >= 108 * 18) {   // border table, one workgroup per class: delta[class][c] = -(sum over the taps that fall outside the volume for the class)
    const int cls = blk - 108 * 18, kz = cls / 9, ky = (cls / 3) % 3, kx = cls % 3;
    const int c = tid % 48, part = tid / 48;               // 10 tap groups x 48 channels (32 threads idle)
    float acc = 0.f;
    if (part < 10)
      for (int d = part; d < 27; d += 10) {
        const int dz = d / 9 - 1, dy = (d / 3) % 3 - 1, dx = d % 3 - 1;
        const bool valid = !((kz == 0 && dz < 0) || (kz == 2 && dz > 0) || (ky == 0 && dy < 0) || (ky == 2 && dy > 0) || (kx == 0 && dx < 0) || (kx == 2 && dx > 0));
        if (valid) continue;
        const float* w1 = W1T + ((long)d * 48 + c) * 48;
        for (int co = 0; co < 48; ++co) acc -= bt[co] * w1[co];
      }
    swt[tid] = acc;
    __syncthreads();
    if (tid < 48) {
      float v = 0.f;
      for (int p = 0; p < 10; ++p) v += swt[p * 48 + tid];
      delta[cls * 48 + tid] = v;
    }
    return;
  }
  const int ck = blk / 18, fr = blk - ck * 18;
  int gi = 0, iz = 0, iy = 0, s = 0, run = 0;
  for (gi = 0; gi < 16; ++gi) {
    const int cnt = cc_ncnt(gi >> 2) * cc_ncnt(gi & 3) * 3;
    if (ck < run + cnt) break;
    run += cnt;
  }
  {
    const int loc = ck - run, cy = cc_ncnt(gi & 3);
    s = loc % 3;
    const int zy = loc / 3;
    iz = zy / cy; iy = zy - iz * cy;
  }
  const int az = gi >> 2, ay = gi & 3, nz = cc_nfirst(az) + iz, ny = cc_nfirst(ay) + iy;
  const int k6 = fr / 3, nt = fr - k6 * 3;
  const int ax = cc::BLK_AX[k6], nx = cc::BLK_NX[k6] - 1;
  const int lane = tid >> 3, j = tid & 7, li = lane & 15, g = lane >> 4;
  const int cil = 8 * g + j;                               // input channel 32 s + cil
  float acc = 0.f;
  for (int dz = -1; dz <= 1; ++dz) {
    const int tz = az + dz, qz = tz < 0 ? -1 : (tz > 3 ? 1 : 0);
    if (qz != nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int ty = ay + dy, qy = ty < 0 ? -1 : (ty > 3 ? 1 : 0);
      if (qy != ny) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int tx = ax + dx, qx = tx < 0 ? -1 : (tx > 3 ? 1 : 0);
        if (qx != nx) continue;                            // (block-uniform branches)
        const int ph = ((tz & 3) * 4 + (ty & 3)) * 4 + (tx & 3), d = ((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1);
        __syncthreads();
        if (tid < 384) {                                   // 32 contiguous rows of 48 floats
          const int r = tid / 12, q = tid - r * 12;
          *reinterpret_cast<float4*>(swt + r * RSF + q * 4) = *reinterpret_cast<const float4*>(WtT + ((long)ph * 96 + 32 * s + r) * 48 + q * 4);
        } else if (tid < 384 + 128) {                      // 16 rows: channel of accumulator row r = 12 (r >> 2) + 4 nt + (r & 3); 128 threads x 1.5 float4
          for (int i = tid - 384; i < 192; i += 128) {
            const int r = i / 12, q = i - r * 12, c = 12 * (r >> 2) + 4 * nt + (r & 3);
            *reinterpret_cast<float4*>(sw1 + r * RSF + q * 4) = *reinterpret_cast<const float4*>(W1T + ((long)d * 48 + c) * 48 + q * 4);
          }
        }
        __syncthreads();
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          const float4 u = *reinterpret_cast<const float4*>(swt + cil * RSF + q * 4), w = *reinterpret_cast<const float4*>(sw1 + li * RSF + q * 4);
          s2 += u.x * w.x + u.y * w.y + u.z * w.z + u.w * w.w;
        }
        acc += s2;
      }
    }
  }
  const bf16_t wq = f2bf(acc);
  Wcp[((long)blk * 64 + lane) * 8 + j] = wq;
  // M[n][ci][c] = sum over the phases a with n in N(a) of the (bf16-rounded) composed weights: the output summed over a sample's fine voxels is
  // sum_n (sum of x over the cells j with j + n inside the grid) . M[n]  (k_cconv_mean)
  if (Mtab) atomicAdd(Mtab + (((long)((nz + 1) * 3 + (ny + 1)) * 3 + (nx + 1)) * 96 + 32 * s + cil) * 48 + 12 * (li >> 2) + 4 * nt + (li & 3), bf2f(wq));
}

long k_cconv_pack_ws_floats() { return CC_WTT + CC_W1T; }

int k_cconv_pack(const float* Wt, const float* W1, const float* bt, void* Wcp, float* delta, float* ws, hipStream_t st, float* Mtab) {
  hipLaunchKernelGGL(cconv_tr_kernel, dim3(96 + 48), dim3(256), 0, st, Wt, W1, ws);
  NMH_CHECK_LAUNCH();
  if (Mtab) {
    hipError_t e = nmh_zero_async(Mtab, sizeof(float) * 27 * 96 * 48, st);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(cconv_pack_kernel, dim3(108 * 18 + 27), dim3(512), 0, st, (const float*)ws, bt, (bf16_t*)Wcp, delta, Mtab);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ---- the mean of the composed output, per (sample, channel), from the COARSE tensor ------------------------------------------------
// y1 is linear in x, so its sum over a sample's fine voxels is  sum_n S_n . M[n]  +  sum_class count(class) delta[class], with S_n[ci] the sum of x over
// the source cells s for which s - n is a cell (n = 0: all cells; n_axis = +1: all but the first plane of that axis; -1: all but the last) and M from
// the pack above.  (1) class sums C27[b][class][ci] of x by position class (first / interior / last per axis), one pass over the 98-MB coarse tensor;
// (2) S_n from the classes, the 2592-term dot products and the border constant.  With the mean known BEFORE the fine tensor is written, the InstanceNorm
// that follows needs no pass of its own over it: the producer stores lrelu(y1 - mean), the scale 1 / std goes into the consumer's weights (positive, so
// it commutes with the LeakyReLU).
__device__ __forceinline__ void unpack8_cc(const uint4& u, float (&v)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__global__ __launch_bounds__(256) void cconv_class_sums_kernel(const bf16_t* __restrict__ x, double* __restrict__ C27, int v, int ysplit) {
  // block = (z plane x y chunk, sample); wave = rows y of the chunk, lane = (8-channel chunk cl of 12, x lane xl of 5).  The row's (z, y) classes are wave-uniform
  // (scalar branch into one of 9 accumulator sets of three x classes); only the first / last cell of a row leave the interior x class.
  __shared__ float red[27 * 96];
  const int zc = blockIdx.x, z = zc / ysplit, yc = zc - z * ysplit, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = lane % 12, xl = lane / 12;
  for (int i = tid; i < 27 * 96; i += 256) red[i] = 0.f;
  __syncthreads();
  const int rows = (v + ysplit - 1) / ysplit, y0 = yc * rows, y1 = y0 + rows < v ? y0 + rows : v;
  const int kz = z == 0 ? 0 : (z == v - 1 ? 2 : 1);
  float acc[3][3][8];   // [y class][x class][channel]
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[a][k][j] = 0.f;
  if (xl < 5) {
    for (int y = y0 + wave; y < y1; y += 4) {
      const bf16_t* xp = x + ((((long)b * v + z) * v + y) * v) * 96 + cl * 8;
      float r0[8], r1[8], r2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { r0[j] = 0.f; r1[j] = 0.f; r2[j] = 0.f; }
      for (int xb = 0; xb < v; xb += 40) {   // eight cells per lane in flight (the row of a 40^3 grid in one go)
        uint4 raw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int xx = xb + xl + 5 * k;
          raw[k] = make_uint4(0, 0, 0, 0);
          if (xx < v) raw[k] = *reinterpret_cast<const uint4*>(xp + (long)xx * 96);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int xx = xb + xl + 5 * k;
          float f[8];
          unpack8_cc(raw[k], f);
          const bool first = xx == 0, last = xx == v - 1;
#pragma unroll
          for (int j = 0; j < 8; ++j) { r0[j] += first ? f[j] : 0.f; r2[j] += last ? f[j] : 0.f; r1[j] += (first || last) ? 0.f : f[j]; }
        }
      }
      const int ky = __builtin_amdgcn_readfirstlane(y == 0 ? 0 : (y == v - 1 ? 2 : 1));
      if (ky == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[0][0][j] += r0[j]; acc[0][1][j] += r1[j]; acc[0][2][j] += r2[j]; }
      } else if (ky == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[2][0][j] += r0[j]; acc[2][1][j] += r1[j]; acc[2][2][j] += r2[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[1][0][j] += r0[j]; acc[1][1][j] += r1[j]; acc[1][2][j] += r2[j]; }
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) if (acc[a][k][j] != 0.f) atomicAdd(&red[(a * 3 + k) * 96 + cl * 8 + j], acc[a][k][j]);
  }
  __syncthreads();
  for (int i = tid; i < 9 * 96; i += 256) if (red[i] != 0.f) atomicAdd(C27 + ((long)b * 27 + kz * 9 + i / 96) * 96 + i % 96, (double)red[i]);
}
// block (n, b): S_n [96] from the class sums, then its 96 x 48 share of the dot products -> fp64 accumulators macc [B][48]
__global__ __launch_bounds__(256) void cconv_mean_dot_kernel(const double* __restrict__ C27, const float* __restrict__ Mtab, double* __restrict__ macc) {
  __shared__ float S[96];
  __shared__ float part[5 * 48];
  const int n = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int nz = n / 9 - 1, ny = (n / 3) % 3 - 1, nx = n % 3 - 1;
  if (tid < 96) {
    double sum = 0.0;
    for (int k = 0; k < 27; ++k) {
      const int kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
      // a source cell s feeds the output cell j = s - n: counted when j is a cell.  n = +1: s >= 1 (not the first plane); n = -1: s <= v - 2 (not the last)
      const bool in = !((nz > 0 && kz == 0) || (nz < 0 && kz == 2) || (ny > 0 && ky == 0) || (ny < 0 && ky == 2) || (nx > 0 && kx == 0) || (nx < 0 && kx == 2));
      if (in) sum += C27[((long)b * 27 + k) * 96 + tid];
    }
    S[tid] = (float)sum;
  }
  __syncthreads();
  const int c = tid % 48, pt = tid / 48;
  if (pt < 5) {
    float acc = 0.f;
    for (int r = pt; r < 96; r += 5) acc += S[r] * Mtab[((long)n * 96 + r) * 48 + c];
    part[pt * 48 + c] = acc;
  }
  __syncthreads();
  if (tid < 48) atomicAdd(macc + (long)b * 48 + tid, (double)part[tid] + part[48 + tid] + part[96 + tid] + part[144 + tid] + part[192 + tid]);
}
__global__ void cconv_mean_final_kernel(const double* __restrict__ macc, const float* __restrict__ delta, float* __restrict__ mhat, int B, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 48) return;
  const int c = i % 48;
  const double F = 4.0 * v;
  double tot = macc[i];
  for (int k = 0; k < 27; ++k) {
    const int kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
    const double cnt = (kz == 1 ? F - 2 : 1.0) * (ky == 1 ? F - 2 : 1.0) * (kx == 1 ? F - 2 : 1.0);
    tot += cnt * (double)delta[k * 48 + c];
  }
  mhat[i] = (float)(tot / (F * F * F));
}
int k_cconv_mean(const void* x, const float* Mtab, const float* delta, double* C27, float* mhat, int B, int v, hipStream_t st) {
  // C27: [B][27][96] class sums followed by [B][48] dot-product accumulators (fp64)
  if (v < 2) return -2;
  hipError_t e = nmh_zero_async(C27, sizeof(double) * (27 * 96 + 48) * B, st);
  if (e != hipSuccess) return (int)e;
  double* macc = C27 + (long)B * 27 * 96;
  const int ysplit = v > 64 ? 4 : 1;   // few workgroups: each ends with 864 fp64 atomics onto the same 2592 addresses per sample (3200 workgroups: 82 us, the atomics)
  hipLaunchKernelGGL(cconv_class_sums_kernel, dim3(v * ysplit, B), dim3(256), 0, st, (const bf16_t*)x, C27, v, ysplit);
  NMH_CHECK_LAUNCH();
  hipLaunchKernelGGL(cconv_mean_dot_kernel, dim3(27, B), dim3(256), 0, st, (const double*)C27, Mtab, macc);
  NMH_CHECK_LAUNCH();
  hipLaunchKernelGGL(cconv_mean_final_kernel, dim3((B * 48 + 255) / 256), dim3(256), 0, st, (const double*)macc, delta, mhat, B, v);
  NMH_CHECK_LAUNCH();
  return 0;
}

long k_cconv_pack_numel() { return 108L * 18 * 512; }

int k_cconv_fwd(const void* X, const void* Wcp, const float* delta, void* Y, int B, int v, double* stats_acc, hipStream_t st, const float* mhat, float slope) {
  using namespace cc;
  if (v % BY || v % BX || v % BZ) return -2;
  CConvArgs a;
  a.mhat = mhat; a.slope = slope;
  a.X = (const bf16_t*)X; a.Wcp = (const bf16_t*)Wcp; a.delta = delta; a.Y = (bf16_t*)Y; a.stats_acc = stats_acc;
  a.B = B; a.v = v; a.nbz = v / BZ; a.nby = v / BY; a.nbx = v / BX;
  a.total = (long)B * a.nbz * a.nby * a.nbx;
  if (a.total >= (1L << 31) || (long)B * 64 * v * v * v * 48 >= (1L << 40)) return -2;
  if (stats_acc) {
    hipError_t e = nmh_zero_async(stats_acc, sizeof(double) * 2 * 48 * B, st);
    if (e != hipSuccess) return (int)e;
  }
  // the XCD-contiguous ranges need a multiple of 8 workgroups; rounded UP: one grid has 250 blocks -- 248 workgroups would walk 8 ranges of 32 blocks with
  // a stride of 31 and the first workgroup of every range would run two blocks (the launch then takes two block times instead of one)
  long nb = (a.total + 7) / 8 * 8;
  if (nb > 256) nb = 256;
  if (mhat) {
    static NmhPerDeviceOnce attr_cz;
    if (attr_cz.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)cconv_fwd_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return (int)e;
      attr_cz.set();
    }
    hipLaunchKernelGGL((cconv_fwd_kernel<0, true>), dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  static const int dbg = getenv("NMH_CCONV_DBG") ? atoi(getenv("NMH_CCONV_DBG")) : 0;
#define CC_LAUNCH(D)                                                                                                              \
  {                                                                                                                               \
    static NmhPerDeviceOnce attr;                                                                                                     \
    if (attr.need()) {                                                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)cconv_fwd_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      if (e != hipSuccess) return (int)e;                                                                                         \
      attr.set();                                                                                                                \
    }                                                                                                                             \
    hipLaunchKernelGGL(cconv_fwd_kernel<D>, dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);                                     \
  }
  if (dbg == 1) CC_LAUNCH(1) else if (dbg == 2) CC_LAUNCH(2) else if (dbg == 4) CC_LAUNCH(4) else if (dbg == 8) CC_LAUNCH(8) else if (dbg == 6) CC_LAUNCH(6)
  else if (dbg == 10) CC_LAUNCH(10) else if (dbg == 12) CC_LAUNCH(12) else if (dbg == 14) CC_LAUNCH(14) else CC_LAUNCH(0)
#undef CC_LAUNCH
  NMH_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================
// The transpose convolution itself (the residual branch u = ConvT(x) of decoder1: unetr_block.py:151-158, 193-200), same skeleton without
// the halo: u[4j + a] = Wt[a]^T x[j] + bt.  A wave keeps the x fragments of its 32 cells in REGISTERS for the whole block (no neighbours),
// the 64 x (96 x 48) phase weights stream through a 3-slot LDS ring in 48 chunks of 12 fragments ((a_z, a_y) group x k-step: four a_x
// phases x three channel tiles), and a lane leaves with the same 4 voxels x 12 channels as above: 384 contiguous bytes per (lane, line).
// ================================================================================================
namespace up4 {
constexpr int NCHUNK = 48, WCH = 12 * 1024, RING = 3, LDS_BYTES = RING * WCH;
}
struct Up4Args { const bf16_t* X; const bf16_t* W; const float* bt; bf16_t* Y; int B, v, nbz, nby, nbx; long total; };

__global__ __launch_bounds__(512) void upconv4_fwd_kernel(Up4Args a) {
  using namespace up4;
  using cc::BZ; using cc::BY; using cc::BX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int V = a.v, F = 4 * a.v;
  const int nx8 = 8, xcd = blockIdx.x % nx8, jb = blockIdx.x / nx8, jstride = gridDim.x / nx8;
  const long per = (a.total + nx8 - 1) / nx8;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < a.total) ? tbeg + per : a.total;
  // chunk c (12 lane-linear 1-KB images) -> ring slot: wave w takes image w, waves 0..3 also image 8 + w
  auto w_dma = [&](int c, int slot) {
    const char* src = reinterpret_cast<const char*>(a.W) + (long)c * WCH;
    char* dst = smem + slot * WCH;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wave * 1024 + lv * 16),
                                     (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
    if (wave < 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (8 + wave) * 1024 + lv * 16),
                                       (__attribute__((address_space(3))) void*)(dst + (8 + wave) * 1024), 16, 0, 0);
  };
  float bb[3][4];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) bb[n][r] = a.bt[12 * g + 4 * n + r];
  const int z_l = wave >> 1, y_l = (wave & 1) * 4, ly = li >> 3, lx = li & 7;
  long t = tbeg + jb;
  if (t >= tend) return;
  w_dma(0, 0); w_dma(1, 1);
  int slot = 0;
  for (; t < tend; t += jstride) {
    const unsigned tu = (unsigned)t;
    unsigned r1 = tu / (unsigned)a.nbx; const int xb = (int)(tu - r1 * (unsigned)a.nbx);
    unsigned r2 = r1 / (unsigned)a.nby; const int yb = (int)(r1 - r2 * (unsigned)a.nby);
    unsigned r3 = r2 / (unsigned)a.nbz; const int zb = (int)(r2 - r3 * (unsigned)a.nbz);
    const int b = __builtin_amdgcn_readfirstlane((int)r3);
    const int z0 = __builtin_amdgcn_readfirstlane(zb * BZ), y0 = __builtin_amdgcn_readfirstlane(yb * BY), x0 = __builtin_amdgcn_readfirstlane(xb * BX);
    Frag<bf16_t> xf[3][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bf16_t* xc = a.X + ((((long)b * V + z0 + z_l) * V + y0 + y_l + 2 * m + ly) * V + x0 + lx) * 96 + 8 * g;
#pragma unroll
      for (int s = 0; s < 3; ++s) xf[s][m].v = *reinterpret_cast<const bf16x8*>(xc + 32 * s);
    }
    int ck = 0;
#pragma unroll 1
    for (int gi = 0; gi < 16; ++gi) {
      const int az = gi >> 2, ay = gi & 3;
      f32x4 acc[4][3][2];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[p][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 3; ++s, ++ck) {
        // chunk ck has landed once at most the pieces of the next chunk (and, in front of a group's first chunk, the 16 output stores of the
        // previous group, younger still) are outstanding: vector-memory operations retire in order
        const bool after_epi = s == 0 && gi > 0;
        if (wave < 4) {
          if (after_epi) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        } else {
          if (after_epi) asm volatile("s_waitcnt vmcnt(17) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        {
          const int cn = ck + 2 >= NCHUNK ? ck + 2 - NCHUNK : ck + 2;
          const int sn = slot == 0 ? 2 : slot - 1;          // (slot + 2) % 3
          w_dma(cn, sn);
        }
        const char* wsrc = smem + slot * WCH + lane * 16;
        slot = slot == 2 ? 0 : slot + 1;
        Frag<bf16_t> wf[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) wf[i].v = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) mma(acc[p][n][m], wf[p * 3 + n], xf[s][m]);
      }
      const int zf = 4 * (z0 + z_l) + az;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int yf = 4 * (y0 + y_l + 2 * m + ly) + ay;
        bf16_t* const drow = a.Y + ((((long)b * F + zf) * F + yf) * F + 4 * (x0 + lx)) * 48 + 12 * g;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned w6[6];
#pragma unroll
          for (int q = 0; q < 6; ++q)
            w6[q] = pk_bf16(acc[p][q >> 1][m][(q & 1) * 2] + bb[q >> 1][(q & 1) * 2], acc[p][q >> 1][m][(q & 1) * 2 + 1] + bb[q >> 1][(q & 1) * 2 + 1]);
          bf16_t* dst = drow + p * 48;
          *reinterpret_cast<uint4*>(dst) = make_uint4(w6[0], w6[1], w6[2], w6[3]);
          *reinterpret_cast<uint2*>(dst + 8) = make_uint2(w6[4], w6[5]);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing weight prefetch
  __syncthreads();
}

// Wup bf16 [48 chunks = (a_z, a_y) group x k-step][12 fragments = a_x x channel tile][64 lanes][8] from WtT (workspace of the pack above)
__global__ __launch_bounds__(512) void upconv4_pack_kernel(const float* __restrict__ WtT, bf16_t* __restrict__ Wup) {
  const int blk = blockIdx.x, tid = threadIdx.x;          // one workgroup per fragment
  const int ck = blk / 12, fr = blk - ck * 12, gi = ck / 3, s = ck - gi * 3, ax = fr / 3, n = fr - ax * 3;
  const int lane = tid >> 3, j = tid & 7, li = lane & 15, g = lane >> 4;
  const int ph = gi * 4 + ax, ci = 32 * s + 8 * g + j, c = 12 * (li >> 2) + 4 * n + (li & 3);
  Wup[((long)blk * 64 + lane) * 8 + j] = f2bf(WtT[((long)ph * 96 + ci) * 48 + c]);
}

long k_upconv4_pack_numel() { return 48L * 12 * 512; }

int k_upconv4_pack(const float* ws, void* Wup, hipStream_t st) {
  hipLaunchKernelGGL(upconv4_pack_kernel, dim3(48 * 12), dim3(512), 0, st, ws, (bf16_t*)Wup);
  NMH_CHECK_LAUNCH();
  return 0;
}

int k_upconv4_fwd(const void* X, const void* Wup, const float* bt, void* Y, int B, int v, hipStream_t st) {
  using namespace cc;
  if (v % BY || v % BX || v % BZ) return -2;
  Up4Args a;
  a.X = (const bf16_t*)X; a.W = (const bf16_t*)Wup; a.bt = bt; a.Y = (bf16_t*)Y;
  a.B = B; a.v = v; a.nbz = v / BZ; a.nby = v / BY; a.nbx = v / BX;
  a.total = (long)B * a.nbz * a.nby * a.nbx;
  if (a.total >= (1L << 31) || (long)B * 64 * v * v * v * 48 >= (1L << 40)) return -2;
  long nb = (a.total + 7) / 8 * 8;      // (rounded up: see k_cconv_fwd)
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(upconv4_fwd_kernel, dim3((unsigned)nb), dim3(512), up4::LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================
// Weight gradient of decoder1's conv1 THROUGH the composition (backward of the op above with respect to conv1.weight):
//     dW1[c][co][d] = sum_p dy1[p][c] u[p + d][co]      with u = ConvT(x)   (unetr_block.py:35-44 backward; the 4 TFLOP/step launch of conv48_wgrad)
//  =  sum_a sum_ci Wt[ci][co][(a + d) mod 4] . G[a][n(a, d)][ci][c],         G[a][n][ci][c] = sum_j x[j + n][ci] dy1[4j + a][c]
// (the transpose conv's bias drops out: its term sum_p dy1[p][c] is the sum of an InstanceNorm input gradient = 0).  G are the same 216
// (phase, neighbour) blocks as the forward's composed weights: a quarter of the FLOPs, contraction over the coarse cells.
// Kernel: workgroup = (unit, slab).  A unit is one (a_z, a_y) phase group with at most two of its (n_z, n_y) neighbour lines (corner groups,
// which have four, are two units) = 6 or 12 blocks of 96 x 48; a slab is every S-th PAIR of coarse lines (b, z, y..y+1).  Per pair the two
// fine gradient lines (4z + a_z, 4y + a_y: 2 x 160 x 48, stored phase-major [a_x][cell][48] in LDS) and the x lines (z + n_z, y + n_y, with a
// zero cell at both ends, stored as two 48-channel halves) go through a double-buffered LDS stage (register prefetch of the next pair); both
// MFMA operands are contraction-major and come out of the tiles by ds_read_b64_tr_b16 at computed addresses (96-byte rows: every bank once).
// 12 waves: wave = ((a_x, n_x) block, ci half) keeps 9 accumulator tiles per neighbour line.  Partials per workgroup -> workspace; the
// reduce + chain-rule kernels below fold them into dW1.
// ================================================================================================
namespace ccw {
constexpr int VMAX = 40;
constexpr int XHALF = (VMAX + 2) * 96, XLINE = 2 * XHALF, XCOMB = 2 * XLINE;      // x: [comb][line][ci half][cell + 1][48 ch]
constexpr int DPH = VMAX * 96, DLINE = 4 * DPH;                                   // dy: [line][a_x][cell][48 ch]
constexpr int OFF_X = 0, OFF_DY = 2 * XCOMB, STAGE = OFF_DY + 2 * DLINE, OFF_ZERO = 2 * STAGE, LDS_BYTES = OFF_ZERO + 256;
constexpr int MAXU = 24;
static_assert(LDS_BYTES <= 163840, "LDS budget");
}  // namespace ccw

struct CCWUnit { signed char az, ay, ncomb, nz0, ny0, nz1, ny1, pad; int wg0, nslab; int blk0, blk1; };   // blk: index of the comb's first G block (x 6 + k6)
struct CCWArgs {
  const bf16_t* X; const bf16_t* dY; float* part;
  int B, v, nunit; long npair;
  CCWUnit u[ccw::MAXU];
};

__global__ __launch_bounds__(768) void cconv_wgrad_kernel(CCWArgs a) {
  using namespace ccw;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, p = lane & 15;
  // unit / slab of this workgroup
  int ui = 0;
  for (int i = 0; i < a.nunit; ++i)
    if ((int)blockIdx.x >= a.u[i].wg0) ui = i;
  const CCWUnit U = a.u[ui];
  const int slab = blockIdx.x - U.wg0, nslab = U.nslab, ncomb = U.ncomb;
  const int v = a.v, F = 4 * v, hv = v >> 1;
  const int b6 = wave >> 1, half = wave & 1, ax = cc::BLK_AX[b6], nx = cc::BLK_NX[b6] - 1;

  // zero cells at both ends of every x line (never overwritten) and the zero rows
  for (int i = tid; i < 2 * STAGE / 16; i += 768) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 16) reinterpret_cast<uint4*>(smem + OFF_ZERO)[tid] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // chunk assignment of the stage loads: dy 2 * 4v * 6 chunks, then ncomb * 2 * v * 12 chunks of x.  Everything about a thread's chunks except
  // the pair's base addresses is invariant: global element offset, LDS byte offset and (for x) the line / plane shift are computed ONCE
  // (recomputing them per pair -- a dozen divisions by the runtime edge per chunk -- was most of the kernel's time: 3.9 ms at 8 grids)
  const int ndy = 2 * 4 * v * 6, nxc = 2 * v * 12, ntot = ndy + ncomb * nxc;
  constexpr int NLD = 5;   // 768 * 5 >= 1920 + 2 * 960
  uint4 hreg[NLD];
  int goff[NLD], loff[NLD], meta[NLD];      // meta: -1 none, 0 dy, else 1 + (dz + 1) * 8 + (dyl + 1) for x (dz in -1..1, dyl = line + n_y in -1..2)
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 768 * i;
    goff[i] = 0; loff[i] = 0; meta[i] = -1;
    if (idx < ndy) {
      const int line = idx / (24 * v), rem = idx - line * (24 * v), vox = rem / 6, c6 = rem - vox * 6;
      goff[i] = line * 4 * F * 48 + vox * 48 + c6 * 8;
      loff[i] = OFF_DY + line * DLINE + (vox & 3) * DPH + (vox >> 2) * 96 + c6 * 16;
      meta[i] = 0;
    } else if (idx < ntot) {
      const int k = idx - ndy, comb = k / nxc, r0 = k - comb * nxc, line = r0 / (12 * v), rem = r0 - line * (12 * v), cell = rem / 12, c12 = rem - cell * 12;
      const int dz = comb ? U.nz1 : U.nz0, dyl = line + (comb ? U.ny1 : U.ny0);
      const int hf = c12 >= 6, c6 = c12 - 6 * hf;
      goff[i] = ((dz * v + dyl) * v + cell) * 96 + c12 * 8;
      loff[i] = OFF_X + comb * XCOMB + line * XLINE + hf * XHALF + (cell + 1) * 96 + c6 * 16;
      meta[i] = 1 + (dz + 1) * 8 + (dyl + 1);
    }
  }
  auto gload = [&](long pair) {
    const unsigned pu = (unsigned)pair;
    const unsigned q1 = pu / (unsigned)hv, y2 = pu - q1 * (unsigned)hv, b = q1 / (unsigned)v, z = q1 - b * (unsigned)v;
    const bf16_t* dyb = a.dY + ((((long)b * F + 4 * (long)z + U.az) * F + 4 * (long)(2 * y2) + U.ay) * F) * 48;
    const bf16_t* xb = a.X + ((((long)b * v + z) * v + 2 * y2) * v) * 96;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      hreg[i] = make_uint4(0, 0, 0, 0);
      if (meta[i] == 0) hreg[i] = *reinterpret_cast<const uint4*>(dyb + goff[i]);
      else if (meta[i] > 0) {
        const int m = meta[i] - 1, zz = (int)z + (m >> 3) - 1, yy = (int)(2 * y2) + (m & 7) - 1;
        if ((unsigned)zz < (unsigned)v && (unsigned)yy < (unsigned)v) hreg[i] = *reinterpret_cast<const uint4*>(xb + goff[i]);
      }
    }
  };
  auto sstore = [&](int st) {
    char* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (meta[i] >= 0) *reinterpret_cast<uint4*>(base + loff[i]) = hreg[i];
  };

  f32x4 acc[2][3][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ksteps = (2 * v + 31) / 32;
  long pair = slab;
  int st = 0;
  if (pair < a.npair) { gload(pair); sstore(0); }
  __syncthreads();
  for (; pair < a.npair; pair += nslab, st ^= 1) {
    const long nxt = pair + nslab;
    if (nxt < a.npair) gload(nxt);
    const char* xs = smem + st * STAGE + OFF_X + half * XHALF;
    const char* ds = smem + st * STAGE + OFF_DY + ax * DPH;
    for (int t = 0; t < ksteps; ++t) {
      // rows of this lane in the k-step: cell i -> (line, x); rows past the pair read zeros
      int offx[2], offd[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 32 * t + 16 * h + 4 * g + (p >> 2);
        const int line = i >= v, xi = i - line * v;
        const bool ok = i < 2 * v;
        offx[h] = ok ? line * XLINE + (xi + 1 + nx) * 96 + (p & 3) * 8 : -1;
        offd[h] = ok ? line * DLINE + xi * 96 + (p & 3) * 8 : -1;
      }
      const char* zr = smem + OFF_ZERO + (p & 3) * 8;
      Frag<bf16_t> df[3];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const bf16x4 lo = ds_read_tr16(offd[0] >= 0 ? ds + offd[0] + ct * 32 : zr), hi = ds_read_tr16(offd[1] >= 0 ? ds + offd[1] + ct * 32 : zr);
        df[ct].v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c < ncomb) {
#pragma unroll
          for (int it = 0; it < 3; ++it) {
            const bf16x4 lo = ds_read_tr16(offx[0] >= 0 ? xs + c * XCOMB + offx[0] + it * 32 : zr), hi = ds_read_tr16(offx[1] >= 0 ? xs + c * XCOMB + offx[1] + it * 32 : zr);
            Frag<bf16_t> xf;
            xf.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) mma(acc[c][it][ct], xf, df[ct]);     // rows ci = 48 half + 16 it + 4 g + r, column c = 16 ct + p
          }
        }
      }
    }
    if (nxt < a.npair) sstore(st ^ 1);
    __syncthreads();
  }
  // partial blocks of this workgroup: part[wg][comb][b6][ci][c]
  float* out = a.part + (long)blockIdx.x * (2 * 6 * 96 * 48);
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < ncomb)
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[((c * 6 + b6) * 96 + 48 * half + 16 * it + 4 * g + r) * 48 + 16 * ct + p] = acc[c][it][ct][r];
}

// ---- the same kernel with the stage filled by LDS-DMA (round 5) --------------------------------------------------------------------------
// The register-staged version above spends a third of an iteration outside its MFMAs: every wave stalls on its five global loads, writes them
// to LDS (the phase-major dy layout puts four consecutive voxels 3840 bytes = 0 banks apart: 21 % of the LDS cycles were conflicts) and meets
// the others at the barrier -- all twelve waves in the same phase at the same time (PMC: MFMA busy 0.26, LDS busy 0.24, waves waiting 39 %).
// Here a stage is a ring slot filled by global_load_lds: a wave-load covers 1 KB of CONSECUTIVE stage bytes and each lane's source is the global
// chunk that belongs there (zero cells, rows of lines outside the volume and the padding come from a 16-byte zero page, so a slot needs no
// initialisation and holds its own zero rows), one barrier per pair and no register staging; units with one neighbour line run a ring of three
// slots (the wait leaves the youngest pair in flight), units with two a ring of two.  Both operands leave LDS by raw ds_read_b64_tr_b16 (the
// builtin would be ordered behind every DMA in flight: DESIGN 3.2) at addresses computed once per kernel.
namespace ccw2 {
constexpr int DPH = 40 * 96, DLINE = 4 * DPH, DYB = 2 * DLINE;                 // dy: [line][a_x][cell][48 ch]                      30 wave-loads
constexpr int XHALF = 42 * 96, XLINE = 2 * XHALF, XCOMB = 16384;               // x: [comb][line][ci half][cell + 1][48 ch] + 256 B  16 wave-loads per comb
constexpr int ZROW = DYB + 2 * XLINE;                                          // the padding of comb 0 (comb 1: + XCOMB): always zeros
constexpr int stage_bytes(int nc) { return DYB + nc * XCOMB; }
constexpr int DUMP = 3 * stage_bytes(1) > 2 * stage_bytes(2) ? 3 * stage_bytes(1) : 2 * stage_bytes(2);   // target of the wave-loads past the stage
constexpr int LDS_BYTES = DUMP + 1024;
static_assert(LDS_BYTES <= 163840, "LDS budget");
}  // namespace ccw2
__device__ uint4 g_zero16_ccw[1];

template <int NC, int KS>
__device__ __forceinline__ void ccw2_body(const CCWArgs& a, const CCWUnit& U, char* smem) {
  using namespace ccw2;
  constexpr int NST = NC == 1 ? 3 : 2, STG = stage_bytes(NC), NGRP = STG / 1024, NG = (NGRP + 11) / 12;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, p = lane & 15;
  const int slab = blockIdx.x - U.wg0, nslab = U.nslab;
  const int v = a.v, F = 4 * v, hv = v >> 1;
  const int b6 = wave >> 1, half = wave & 1, ax = cc::BLK_AX[b6], nx = cc::BLK_NX[b6] - 1;
  unsigned long long zpage = (unsigned long long)(const void*)g_zero16_ccw;
  asm volatile("" : "+v"(zpage));

  // source of this lane's chunk in each of the wave's loads, one register: (16-byte units from the pair's dy / x base) * 32 + kind; kind 0: zero page,
  // 1: dy, 2 + (dz + 1) * 8 + (dyl + 1): x with a plane / line shift (the chunk is zero when the shifted line is outside the volume)
  int desc[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int gi = wave + 12 * i, q = gi * 64 + lane;
    desc[i] = 0;
    if (q < DYB / 16) {
      const int line = q / 960, r = q - line * 960, axq = r / 240, r2 = r - axq * 240, cell = r2 / 6, c6 = r2 - cell * 6;
      if (cell < v) desc[i] = ((line * 4 * F * 48 + (4 * cell + axq) * 48 + c6 * 8) >> 3) * 32 + 1;
    } else if (gi < NGRP) {
      const int k = q - DYB / 16, comb = k >> 10, r = k & 1023;
      if (r < 1008) {
        const int line = r / 504, r1 = r - line * 504, hf = r1 / 252, r2 = r1 - hf * 252, cellp = r2 / 6, c6 = r2 - cellp * 6, cell = cellp - 1;
        const int dz = comb ? U.nz1 : U.nz0, dyl = line + (comb ? U.ny1 : U.ny0);
        if (cell >= 0 && cell < v) desc[i] = ((((dz * v + dyl) * v + cell) * 96 + (hf * 6 + c6) * 8) >> 3) * 32 + 2 + (dz + 1) * 8 + (dyl + 1);
      }
    }
  }
  auto issue = [&](long pair, int stg) {
    const bool live = pair < a.npair;
    const unsigned pu = live ? (unsigned)pair : 0u;
    const unsigned q1 = pu / (unsigned)hv, y2 = pu - q1 * (unsigned)hv, b = q1 / (unsigned)v, z = q1 - b * (unsigned)v;
    const char* dyb = reinterpret_cast<const char*>(a.dY + ((((long)b * F + 4 * (long)z + U.az) * F + 4 * (long)(2 * y2) + U.ay) * F) * 48);
    const char* xb = reinterpret_cast<const char*>(a.X + ((((long)b * v + z) * v + 2 * y2) * v) * 96);
    // which (dz, dyl) shifts stay inside the volume for this pair: bit (dz + 1) * 8 + (dyl + 1) + 2; bit 1: dy; bit 0 (zero page) never
    unsigned okm = 0u;
    if (live) {
      unsigned ym = 0u;
#pragma unroll
      for (int d = -1; d <= 2; ++d) ym |= ((unsigned)((int)(2 * y2) + d) < (unsigned)v ? 1u : 0u) << (d + 1);
#pragma unroll
      for (int d = -1; d <= 1; ++d)
        if ((unsigned)((int)z + d) < (unsigned)v) okm |= ym << ((d + 1) * 8 + 2);
      okm |= 2u;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int gi = wave + 12 * i, kind = desc[i] & 31;
      const long off = (long)(desc[i] >> 5) << 4;
      const bool ok = (okm >> kind) & 1u;
      const unsigned long long src = ok ? (unsigned long long)((kind == 1 ? dyb : xb) + off) : zpage;
      char* dst = gi < NGRP ? smem + stg * STG + gi * 1024 : smem + DUMP;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // operand rows of this lane in the k-steps (stage-relative byte addresses; rows past the pair read the zero rows)
  unsigned adx[KS][2], add[KS][2];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 32 * t + 16 * h + 4 * g + (p >> 2);
      const int line = i >= v, xi = i - line * v;
      const bool ok = i < 2 * v;
      adx[t][h] = (unsigned)((ok ? DYB + half * XHALF + line * XLINE + (xi + 1 + nx) * 96 : ZROW) + (p & 3) * 8);
      add[t][h] = (unsigned)((ok ? ax * DPH + line * DLINE + xi * 96 : ZROW) + (p & 3) * 8);
    }
  const unsigned smem_u = lds_addr_u(smem);

  f32x4 acc[NC][3][3];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads of k-step t: the dy tiles and the x tiles of neighbour line c; a set is requested one half-step ahead of its MFMAs
  TrFrag df[2][3], xf[2][3];   // df: by parity of t; xf: NC == 2: by neighbour line, NC == 1: by parity of t
  auto read_d = [&](TrFrag (&f)[3], unsigned sb, int t) {
    const unsigned d0 = sb + add[t][0], d1 = sb + add[t][1];
    tr_read1<0>(f[0].lo, d0); tr_read1<0>(f[0].hi, d1);
    tr_read1<32>(f[1].lo, d0); tr_read1<32>(f[1].hi, d1);
    tr_read1<64>(f[2].lo, d0); tr_read1<64>(f[2].hi, d1);
  };
  auto read_x0 = [&](TrFrag (&f)[3], unsigned sb, int t) {
    const unsigned x0 = sb + adx[t][0], x1 = sb + adx[t][1];
    tr_read1<0>(f[0].lo, x0); tr_read1<0>(f[0].hi, x1);
    tr_read1<32>(f[1].lo, x0); tr_read1<32>(f[1].hi, x1);
    tr_read1<64>(f[2].lo, x0); tr_read1<64>(f[2].hi, x1);
  };
  auto read_x1 = [&](TrFrag (&f)[3], unsigned sb, int t) {
    const unsigned x0 = sb + adx[t][0], x1 = sb + adx[t][1];
    tr_read1<XCOMB>(f[0].lo, x0); tr_read1<XCOMB>(f[0].hi, x1);
    tr_read1<XCOMB + 32>(f[1].lo, x0); tr_read1<XCOMB + 32>(f[1].hi, x1);
    tr_read1<XCOMB + 64>(f[2].lo, x0); tr_read1<XCOMB + 64>(f[2].hi, x1);
  };
  auto pin3 = [&](TrFrag (&f)[3]) { tr_pin(f[0]); tr_pin(f[1]); tr_pin(f[2]); };
  auto mma9 = [&](f32x4 (&ac)[3][3], const TrFrag (&x)[3], const TrFrag (&d)[3]) {
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) mma(ac[it][ct], tr_frag(x[it]), tr_frag(d[ct]));     // rows ci = 48 half + 16 it + 4 g + r, column c = 16 ct + p
    // the nine products are issued before whatever follows (the wait for the next set)
    asm volatile("" : "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[0][2]), "+v"(ac[1][0]), "+v"(ac[1][1]), "+v"(ac[1][2]), "+v"(ac[2][0]), "+v"(ac[2][1]), "+v"(ac[2][2]));
  };

  long pair = slab;
  int cur = 0;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(pair + (long)s * nslab, s);
  for (; pair < a.npair; pair += nslab) {
    if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned sb = smem_u + (unsigned)(cur * STG);
    read_d(df[0], sb, 0);
    read_x0(xf[0], sb, 0);
    {   // the next pair's loads: their address arithmetic runs under the latency of the first reads
      int nxt = cur + NST - 1;
      if (nxt >= NST) nxt -= NST;
      issue(pair + (long)(NST - 1) * nslab, nxt);
    }
    tr_wait();
    pin3(df[0]); pin3(xf[0]);
    if constexpr (NC == 2) {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        read_x1(xf[1], sb, t);
        mma9(acc[0], xf[0], df[t & 1]);
        tr_wait();
        pin3(xf[1]);
        if (t + 1 < KS) { read_d(df[(t + 1) & 1], sb, t + 1); read_x0(xf[0], sb, t + 1); }
        mma9(acc[NC - 1], xf[1], df[t & 1]);
        if (t + 1 < KS) { tr_wait(); pin3(df[(t + 1) & 1]); pin3(xf[0]); }
      }
    } else {
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        if (t + 1 < KS) { read_d(df[(t + 1) & 1], sb, t + 1); read_x0(xf[(t + 1) & 1], sb, t + 1); }
        mma9(acc[0], xf[t & 1], df[t & 1]);
        if (t + 1 < KS) { tr_wait(); pin3(df[(t + 1) & 1]); pin3(xf[(t + 1) & 1]); }
      }
    }
    if (++cur == NST) cur = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (zero-page) loads
  // partial blocks of this workgroup: part[wg][comb][b6][ci][c]
  float* out = a.part + (long)blockIdx.x * (2 * 6 * 96 * 48);
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((c * 6 + b6) * 96 + 48 * half + 16 * it + 4 * g + r) * 48 + 16 * ct + p] = acc[c][it][ct][r];
}

__global__ __launch_bounds__(768) void cconv_wgrad_dma_kernel(CCWArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int ui = 0;
  for (int i = 0; i < a.nunit; ++i)
    if ((int)blockIdx.x >= a.u[i].wg0) ui = i;
  const CCWUnit U = a.u[ui];
  const int ks = (2 * a.v + 31) / 32;   // k-steps of a pair of lines (v a multiple of 8, <= 40: 1..3)
  if (U.ncomb == 1) {
    if (ks == 3) ccw2_body<1, 3>(a, U, smem);
    else if (ks == 2) ccw2_body<1, 2>(a, U, smem);
    else ccw2_body<1, 1>(a, U, smem);
  } else {
    if (ks == 3) ccw2_body<2, 3>(a, U, smem);
    else if (ks == 2) ccw2_body<2, 2>(a, U, smem);
    else ccw2_body<2, 1>(a, U, smem);
  }
}

// G[block][ci][c] = sum over the slabs of the unit that owns the block (block = (group, neighbour line, (a_x, n_x)): the forward's order)
__global__ void cconv_wgrad_reduce_kernel(CCWArgs a, float* __restrict__ G) {
  // grid = (unit, neighbour line of the unit, (a_x, n_x) block, 18 pieces of 256 elements)
  const int piece = blockIdx.x % 18, r1 = blockIdx.x / 18, k6 = r1 % 6, r2 = r1 / 6, comb = r2 & 1, ui = r2 >> 1;
  const CCWUnit U = a.u[ui];
  if (comb >= U.ncomb) return;
  const int blk = (comb ? U.blk1 : U.blk0) * 6 + k6;
  const int e = piece * 256 + threadIdx.x;
  float s = 0.f;
  for (int w = 0; w < U.nslab; ++w) s += a.part[(long)(U.wg0 + w) * (2 * 6 * 96 * 48) + (comb * 6 + k6) * 4608 + e];
  G[(long)blk * 4608 + e] = s;
}

// The transpose conv's bias: dW1[c][co][d] contains bt[co] * sum_{p : p + d inside the volume} dy1[p][c].  dy1 is the input gradient of an
// affine-free InstanceNorm, so its sum over a whole sample is zero (the same fact that makes conv1's own bias gradient vanish) and the sum over
// the valid voxels is MINUS the sum over the border voxels that tap d excludes.  This kernel takes the sums of dy1 over the 26 border classes
// (k_z, k_y, k_x in {first, interior, last}^3) -- one workgroup per (sample, fine z plane), only the border lines are read in full.
__global__ __launch_bounds__(256) void cconv_dy_border_kernel(const bf16_t* __restrict__ dY, float* __restrict__ C, int F, int planes_per_block) {
  __shared__ float red[27 * 48];
  const int nblk = (F + planes_per_block - 1) / planes_per_block;
  const int b = blockIdx.x / nblk, z0 = (blockIdx.x - b * nblk) * planes_per_block, tid = threadIdx.x;
  const int vl = tid / 6, c8 = tid - vl * 6;          // 42 voxel lanes x 6 groups of 8 channels (threads 252..255 idle)
  for (int i = tid; i < 27 * 48; i += 256) red[i] = 0.f;
  __syncthreads();
  float acc[3][8];
  auto flush = [&](int kz, int ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (acc[kx][j] != 0.f) atomicAdd(&red[((kz * 3 + ky) * 3 + kx) * 48 + c8 * 8 + j], acc[kx][j]);
        acc[kx][j] = 0.f;
      }
  };
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[kx][j] = 0.f;
  if (vl < 42) {
    const int lpb = (F + (int)gridDim.y - 1) / (int)gridDim.y, ly0 = (int)blockIdx.y * lpb, ly1 = ly0 + lpb < F ? ly0 + lpb : F;
    // (1) border lines (first / last plane: every line; other planes: the first / last line): every voxel of them is a border voxel
    for (int fz = z0; fz < z0 + planes_per_block && fz < F; ++fz) {
      const int kz = fz == 0 ? 0 : (fz == F - 1 ? 2 : 1);
      for (int fy = ly0; fy < ly1; ++fy) {
        const int ky = fy == 0 ? 0 : (fy == F - 1 ? 2 : 1);
        if (kz == 1 && ky == 1) continue;
        const bf16_t* base = dY + ((((long)b * F + fz) * F + fy) * F) * 48 + c8 * 8;
        for (int fx = vl; fx < F; fx += 42) {
          float v8[8];
          Vec8<bf16_t>::load(base + (long)fx * 48, v8);
          const int kx = fx == 0 ? 0 : (fx == F - 1 ? 2 : 1);
#pragma unroll
          for (int q = 0; q < 3; ++q)
            if (q == kx)
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[q][j] += v8[j];
        }
        if (ky != 1 || fy == F - 2 || fy == ly1 - 1) flush(kz, ky);      // the class changes after line 0, after the interior run and after the last line
      }
    }
    // (2) interior lines of interior planes: their two end voxels -- all (line, end) pairs of the workgroup spread over the 42 voxel lanes (one pair per
    //     lane and iteration; walked line by line with two lanes, the pass was a chain of 100 dependent load round trips: 114 us for 15 MB)
    {
      const int zb = z0 < 1 ? 1 : z0, ze = z0 + planes_per_block < F - 1 ? z0 + planes_per_block : F - 1;      // interior planes [zb, ze)
      const int yb = ly0 < 1 ? 1 : ly0, ye = ly1 < F - 1 ? ly1 : F - 1;                                        // interior lines [yb, ye)
      const int ny = ye - yb, npair = (ze > zb && ny > 0) ? (ze - zb) * ny * 2 : 0;
      for (int p = vl; p < npair; p += 42) {
        const int end = p & 1, ln = p >> 1, fz = zb + ln / ny, fy = yb + ln % ny;
        float v8[8];
        Vec8<bf16_t>::load(dY + ((((long)b * F + fz) * F + fy) * F + (end ? F - 1 : 0)) * 48 + c8 * 8, v8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (end) acc[2][j] += v8[j];
          else acc[0][j] += v8[j];
        }
      }
      flush(1, 1);
    }
  }
  __syncthreads();
  for (int e = tid; e < 27 * 48; e += 256)
    if (red[e] != 0.f) atomicAdd(C + e, red[e]);
}

// dW1[c][co][d] += sum over the 8 phases of the group: sum_ci WtT[(a + d) mod 4][ci][co] . G[a][n(a, d)][ci][c]
// (- bt[co] * sum of dy1 over the border classes that tap d excludes: added by the first phase group)
__global__ __launch_bounds__(256) void cconv_wgrad_chain_kernel(const float* __restrict__ G, const float* __restrict__ WtT, const float* __restrict__ bt,
                                                                const float* __restrict__ Cb, float* __restrict__ dW1) {
  __shared__ float sg[96 * 49], sw[96 * 49];
  const int d = blockIdx.x / 8, agrp = blockIdx.x - d * 8, tid = threadIdx.x;
  const int dz = d / 9 - 1, dy = (d / 3) % 3 - 1, dx = d % 3 - 1;
  float acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.f;
  for (int ai = 0; ai < 8; ++ai) {
    const int aa = agrp * 8 + ai, az = aa >> 4, ay = (aa >> 2) & 3, ax = aa & 3;
    const int tz = az + dz, ty = ay + dy, tx = ax + dx;
    const int nz = tz < 0 ? -1 : (tz > 3 ? 1 : 0), ny = ty < 0 ? -1 : (ty > 3 ? 1 : 0), nxx = tx < 0 ? -1 : (tx > 3 ? 1 : 0);
    const int ph = ((tz & 3) * 4 + (ty & 3)) * 4 + (tx & 3);
    // block index of (a, n) in the forward's order
    int base = 0;
    const int gi = az * 4 + ay;
    for (int q = 0; q < gi; ++q) base += cc_ncnt(q >> 2) * cc_ncnt(q & 3);
    const int izy = (nz - cc_nfirst(az)) * cc_ncnt(ay) + (ny - cc_nfirst(ay));
    const int k6 = ax == 0 ? (nxx < 0 ? 0 : 1) : (ax == 3 ? (nxx > 0 ? 5 : 4) : ax + 1);
    const float* gsrc = G + (long)((base + izy) * 6 + k6) * 4608;
    const float* wsrc = WtT + (long)ph * 4608;
    __syncthreads();
    for (int i = tid; i < 4608; i += 256) { const int r = i / 48, c = i - r * 48; sg[r * 49 + c] = gsrc[i]; sw[r * 49 + c] = wsrc[i]; }
    __syncthreads();
    // thread (tc, to) of a 16 x 16 grid owns the 3 x 3 block c = 3 tc + i, co = 3 to + j: six LDS reads per nine products
    {
      const int c0 = 3 * (tid >> 4), o0 = 3 * (tid & 15);
#pragma unroll 4
      for (int ci = 0; ci < 96; ++ci) {
        const float g0 = sg[ci * 49 + c0], g1 = sg[ci * 49 + c0 + 1], g2 = sg[ci * 49 + c0 + 2];
        const float w0 = sw[ci * 49 + o0], w1 = sw[ci * 49 + o0 + 1], w2 = sw[ci * 49 + o0 + 2];
        acc[0] += g0 * w0; acc[1] += g0 * w1; acc[2] += g0 * w2;
        acc[3] += g1 * w0; acc[4] += g1 * w1; acc[5] += g1 * w2;
        acc[6] += g2 * w0; acc[7] += g2 * w1; acc[8] += g2 * w2;
      }
    }
  }
  if (agrp == 0) {
#pragma unroll
    for (int o = 0; o < 9; ++o) {
      const int c = 3 * (tid >> 4) + o / 3, co = 3 * (tid & 15) + o % 3;
      float sb = 0.f;
      for (int cls = 0; cls < 27; ++cls) {
        const int kz = cls / 9, ky = (cls / 3) % 3, kx = cls % 3;
        const bool excluded = (kz == 0 && dz < 0) || (kz == 2 && dz > 0) || (ky == 0 && dy < 0) || (ky == 2 && dy > 0) || (kx == 0 && dx < 0) || (kx == 2 && dx > 0);
        if (excluded) sb += Cb[cls * 48 + c];
      }
      acc[o] -= bt[co] * sb;
    }
  }
#pragma unroll
  for (int o = 0; o < 9; ++o) {
    const int c = 3 * (tid >> 4) + o / 3, co = 3 * (tid & 15) + o % 3;
    atomicAdd(dW1 + ((long)c * 48 + co) * 27 + d, acc[o]);
  }
}

// The transpose conv's OWN parameters through the composition (their gradient via u = ConvT(x) -> conv1, the part that used to need conv1's input
// gradient on the fine grid):  dWt[ci][co][ph] += sum_{(a, d) : (a + d) mod 4 = ph} sum_c G[a][n(a, d)][ci][c] W1[c][co][d]   (27 pairs per phase),
// dbt[co] += sum_p (conv1^T dy1)[p][co] = - sum_d sum_c W1[c][co][d] . (sum of dy1 over the border classes that tap d excludes)   (zero total sums, as above).
// grid = 64 phases x 3 (d_z): 9 pairs each, the G block and W1T[d] staged in LDS.
__global__ __launch_bounds__(256) void cconv_wgrad_chain_t_kernel(const float* __restrict__ G, const float* __restrict__ W1T, const float* __restrict__ Cb,
                                                                  float* __restrict__ part, float* __restrict__ dbt) {
  __shared__ float sg[96 * 49], sw[48 * 49], scb[27 * 48];
  const int ph = blockIdx.x / 3, dz = (int)(blockIdx.x % 3) - 1, tid = threadIdx.x;
  const int pz = ph >> 4, py = (ph >> 2) & 3, px = ph & 3;
  if (ph == 0)
    for (int i = tid; i < 27 * 48; i += 256) scb[i] = Cb[i];
  float acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.f;
  float bacc = 0.f;
  for (int dyx = 0; dyx < 9; ++dyx) {
    const int dy = dyx / 3 - 1, dx = dyx % 3 - 1, d = ((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1);
    const int az = (pz - dz) & 3, ay = (py - dy) & 3, ax = (px - dx) & 3;
    const int tz = az + dz, ty = ay + dy, tx = ax + dx;
    const int nz = tz < 0 ? -1 : (tz > 3 ? 1 : 0), ny = ty < 0 ? -1 : (ty > 3 ? 1 : 0), nxx = tx < 0 ? -1 : (tx > 3 ? 1 : 0);
    int base = 0;
    const int gi = az * 4 + ay;
    for (int q = 0; q < gi; ++q) base += cc_ncnt(q >> 2) * cc_ncnt(q & 3);
    const int izy = (nz - cc_nfirst(az)) * cc_ncnt(ay) + (ny - cc_nfirst(ay));
    const int k6 = ax == 0 ? (nxx < 0 ? 0 : 1) : (ax == 3 ? (nxx > 0 ? 5 : 4) : ax + 1);
    const float* gsrc = G + (long)((base + izy) * 6 + k6) * 4608;
    const float* wsrc = W1T + (long)d * 2304;
    __syncthreads();
    for (int i = tid; i < 4608; i += 256) { const int r = i / 48, c = i - r * 48; sg[r * 49 + c] = gsrc[i]; }
    for (int i = tid; i < 2304; i += 256) { const int r = i / 48, c = i - r * 48; sw[r * 49 + c] = wsrc[i]; }
    __syncthreads();
    // (c outermost, two at a time: fully unrolled the other way round the 18 x 48 products are all hoisted and spill -- measured 2.4 ms)
    // thread (ti, to) of a 16 x 16 grid owns the 6 x 3 block ci = 6 ti + i, co = 3 to + j: nine LDS reads per eighteen products
    {
      const int i0 = 6 * (tid >> 4), o0 = 3 * (tid & 15);
#pragma unroll 2
      for (int c = 0; c < 48; ++c) {
        const float w0 = sw[c * 49 + o0], w1 = sw[c * 49 + o0 + 1], w2 = sw[c * 49 + o0 + 2];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float gv = sg[(i0 + i) * 49 + c];
          acc[3 * i] += gv * w0; acc[3 * i + 1] += gv * w1; acc[3 * i + 2] += gv * w2;
        }
      }
    }
    if (ph == 0) {     // the bias (once per tap d: the phase-0 workgroups); the class sums sit in LDS
      __syncthreads();
      if (tid < 48) {    // sb[c] = sum of dy1 over the border classes that tap d excludes
        float sb = 0.f;
        for (int cls = 0; cls < 27; ++cls) {
          const int kz = cls / 9, ky = (cls / 3) % 3, kx = cls % 3;
          const bool excluded = (kz == 0 && dz < 0) || (kz == 2 && dz > 0) || (ky == 0 && dy < 0) || (ky == 2 && dy > 0) || (kx == 0 && dx < 0) || (kx == 2 && dx > 0);
          if (excluded) sb += scb[cls * 48 + tid];
        }
        sg[tid] = sb;    // (the G block is done with)
      }
      __syncthreads();
      if (tid < 48)
        for (int c = 0; c < 48; ++c) bacc -= sw[c * 49 + tid] * sg[c];
    }
  }
  // partial [workgroup][ci][co] (contiguous): the 192 workgroups walk the SAME elements in the same order -- atomics on dWt[e][ph] would all land on
  // one 256-byte line at a time (measured 2.4 ms); the kernel below folds the three d_z parts into dWt
#pragma unroll
  for (int o = 0; o < 18; ++o) part[(long)blockIdx.x * 4608 + (6 * (tid >> 4) + o / 3) * 48 + 3 * (tid & 15) + o % 3] = acc[o];
  if (ph == 0 && tid < 48 && dbt) atomicAdd(dbt + tid, bacc);
}

__global__ __launch_bounds__(256) void cconv_wgrad_chain_t_reduce_kernel(const float* __restrict__ part, float* __restrict__ dWt) {
  const int i = blockIdx.x * 256 + threadIdx.x;      // dWt element (e, ph), ph fastest
  if (i >= 4608 * 64) return;
  const int e = i >> 6, ph = i & 63;
  const float* p = part + (long)ph * 3 * 4608 + e;
  dWt[i] += p[0] + p[4608] + p[2 * 4608];
}

long k_cconv_wgrad_ws_floats() { return 256L * (2 * 6 * 96 * 48) + 216L * 4608 + 27 * 48; }

// dW1 [48][48][3][3][3] += conv1 weight gradient from x [B][v^3][96] and dy1 [B][(4v)^3][48]; WtT = first part of the pack workspace of
// nmh_cconv_pack (the transpose-conv weight as [64 phases][96][48]); ws: k_cconv_wgrad_ws_floats() floats
// (valid for a dy1 whose per-sample sums vanish: the gradient of an affine-free InstanceNorm's input, which is what conv1 feeds)
int k_cconv_wgrad(const void* X, const void* dY, const float* WtT, const float* bt, float* dW1, float* dWt, float* dbt, float* ws, int B, int v, int phase, hipStream_t st) {
  using namespace ccw;
  // phase 0: everything; 1: the G partials only (the persistent kernel: main stream); 2: reduce, border sums and the chain rules (small launches that
  // only feed weight gradients: the caller may issue them on a side stream behind phase 1)
  if (v % 8 || v > VMAX) return -2;
  CCWArgs a{};
  a.X = (const bf16_t*)X; a.dY = (const bf16_t*)dY; a.part = ws; a.B = B; a.v = v;
  a.npair = (long)B * v * (v / 2);
  // units: every (a_z, a_y) group with at most two of its neighbour lines; slabs in proportion to the time of a pair: wa + neighbour lines (the bytes a
  // unit streams per pair would give 1.9 + lines; measured with the LDS-DMA kernel: tools/bench_ccw.py)
  static const double wa = [] { const char* e = getenv("NMH_CCW_WA"); return e ? atof(e) : 1.9; }();
  int nu = 0, base = 0;
  double wsum = 0.0, wgt[MAXU];
  for (int gi = 0; gi < 16; ++gi) {
    const int az = gi >> 2, ay = gi & 3, cz = cc_ncnt(az), cy = cc_ncnt(ay), fz = cc_nfirst(az), fy = cc_nfirst(ay);
    const int ncombs = cz * cy;
    for (int first = 0; first < ncombs; first += 2) {
      CCWUnit& u = a.u[nu];
      u.az = (signed char)az; u.ay = (signed char)ay;
      u.ncomb = (signed char)(ncombs - first >= 2 ? 2 : 1);
      const int c0 = first, c1 = first + 1;
      u.nz0 = (signed char)(fz + c0 / cy); u.ny0 = (signed char)(fy + c0 % cy);
      u.nz1 = (signed char)(fz + c1 / cy); u.ny1 = (signed char)(fy + c1 % cy);
      u.blk0 = base + c0; u.blk1 = base + c1;
      wgt[nu] = wa + u.ncomb;
      wsum += wgt[nu];
      ++nu;
    }
    base += ncombs;
  }
  a.nunit = nu;
  // slabs: largest-remainder shares of the part's 256 CUs (one workgroup per CU)
  int wg = 0, ns[MAXU];
  double fr[MAXU];
  for (int i = 0; i < nu; ++i) {
    const double share = 256.0 * wgt[i] / wsum;
    int s = (int)share;
    if (s < 1) s = 1;
    fr[i] = share - s;
    ns[i] = s;
    wg += s;
  }
  while (wg < 256) {
    int best = 0;
    for (int i = 1; i < nu; ++i)
      if (fr[i] > fr[best]) best = i;
    ++ns[best]; fr[best] = -1.0; ++wg;
  }
  wg = 0;
  for (int i = 0; i < nu; ++i) {
    int s = ns[i];
    if ((long)s > a.npair) s = (int)a.npair;
    a.u[i].wg0 = wg; a.u[i].nslab = s;
    wg += s;
  }
  if (wg > 256) return -2;
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)cconv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  static NmhPerDeviceOnce attr2;
  if (attr2.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)cconv_wgrad_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ccw2::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr2.set();
  }
  static const int use_dma = [] { const char* e = getenv("NMH_CCW_DMA"); return e ? atoi(e) : 1; }();
  if (phase != 2) {
    if (use_dma) hipLaunchKernelGGL(cconv_wgrad_dma_kernel, dim3((unsigned)wg), dim3(768), ccw2::LDS_BYTES, st, a);
    else hipLaunchKernelGGL(cconv_wgrad_kernel, dim3((unsigned)wg), dim3(768), LDS_BYTES, st, a);
    NMH_CHECK_LAUNCH();
  }
  if (phase == 1) return 0;
  float* G = ws + 256L * (2 * 6 * 96 * 48);
  float* Cb = G + 216L * 4608;
  hipLaunchKernelGGL(cconv_wgrad_reduce_kernel, dim3((unsigned)(nu * 2 * 6 * 18)), dim3(256), 0, st, a, G);
  NMH_CHECK_LAUNCH();
  {
    hipError_t e = nmh_zero_async(Cb, sizeof(float) * 27 * 48, st);
    if (e != hipSuccess) return (int)e;
  }
  {
    const int F = 4 * v, ppb = 5;
    hipLaunchKernelGGL(cconv_dy_border_kernel, dim3((unsigned)(B * ((F + ppb - 1) / ppb)), 8), dim3(256), 0, st, (const bf16_t*)dY, Cb, F, ppb);
  }
  NMH_CHECK_LAUNCH();
  hipLaunchKernelGGL(cconv_wgrad_chain_kernel, dim3(27 * 8), dim3(256), 0, st, (const float*)G, WtT, bt, (const float*)Cb, dW1);
  NMH_CHECK_LAUNCH();
  if (dWt) {   // the transpose conv's own weight / bias gradient through conv1 (W1T sits behind WtT in the pack workspace)
    // (the per-slab partials at the head of ws are consumed: their space takes the 192 x 4608 partial sums)
    hipLaunchKernelGGL(cconv_wgrad_chain_t_kernel, dim3(64 * 3), dim3(256), 0, st, (const float*)G, WtT + CC_WTT, (const float*)Cb, ws, dbt);
    NMH_CHECK_LAUNCH();
    hipLaunchKernelGGL(cconv_wgrad_chain_t_reduce_kernel, dim3(4608 * 64 / 256), dim3(256), 0, st, (const float*)ws, dWt);
    NMH_CHECK_LAUNCH();
  }
  return 0;
}

// ================================================================================================
// Input gradient THROUGH the composition: dx = ConvT^T(conv1^T(dy1)) is a stride-4 convolution of the fine gradient with a 6x6x6 kernel of 48 -> 96
// matrices,   dx[j][ci] = sum_{p in [0,6)^3} sum_c dy1[4j - 1 + p][c] Wd[p][c][ci],   Wd[p] = Wc[a(p)][n(p)]^T  (p = 0: (a, n) = (3, +1); 1..4: (p - 1, 0);
// 5: (0, -1) per axis): the same 216 blocks as the forward, the same 31 kFLOP per fine voxel instead of 133 -- and conv1's input gradient on the
// fine grid (a full 48 -> 48 conv48 pass that only fed the transpose conv's backward) is never formed.
// Kernel = the forward mirrored: persistent 512-thread workgroup per block of 4x8x8 coarse cells, wave = 32 cells with 6 x 2 accumulator tiles
// (96 channels), the 2 MB of weights through the same 4-slot LDS-DMA ring (108 chunks of 18 fragments = (fine line offset p_z, p_y; three
// k-steps)); the contraction operand of a cell is 576 CONTIGUOUS bytes of a fine line (voxels 4x - 1 .. 4x + 4, 9 k-steps) and is read straight
// from global memory into a 4-chunk register ring, three chunks ahead of the MFMAs.
// ================================================================================================
#ifndef CD_RING
#define CD_RING 4
#endif
constexpr long CD_NUMEL = 36L * 54 * 512;     // + 64 zero elements behind them
// Order of the 36 fine-line offsets (p_z, p_y): a fine line 4z + 3 is offset 4 of cell z and offset 0 of cell z + 1 (4z + 4: offsets 5 and 1), likewise in
// y -- both readers sit in neighbouring waves / lanes of the same workgroup.  Walking the offsets as {0,4},{1,5},{2},{3} x {0,4},{1,5},{2},{3} puts the
// two reads of such a line at most three steps apart (they hit in L2 / the memory-side cache) instead of two thirds of a block apart (HBM twice).
__device__ constexpr unsigned char CD_ORDER[36] = {
    0 * 6 + 0, 0 * 6 + 4, 4 * 6 + 0, 4 * 6 + 4, 0 * 6 + 1, 0 * 6 + 5, 4 * 6 + 1, 4 * 6 + 5, 0 * 6 + 2, 4 * 6 + 2, 0 * 6 + 3, 4 * 6 + 3,
    1 * 6 + 0, 1 * 6 + 4, 5 * 6 + 0, 5 * 6 + 4, 1 * 6 + 1, 1 * 6 + 5, 5 * 6 + 1, 5 * 6 + 5, 1 * 6 + 2, 5 * 6 + 2, 1 * 6 + 3, 5 * 6 + 3,
    2 * 6 + 0, 2 * 6 + 4, 2 * 6 + 1, 2 * 6 + 5, 2 * 6 + 2, 2 * 6 + 3,
    3 * 6 + 0, 3 * 6 + 4, 3 * 6 + 1, 3 * 6 + 5, 3 * 6 + 2, 3 * 6 + 3};
struct CDArgs { const bf16_t* dY; const bf16_t* Wdp; const bf16_t* add; bf16_t* DX; int B, v, nbz, nby, nbx; long total; };

__global__ __launch_bounds__(512) void cconv_dgrad_kernel(CDArgs a) {
  using namespace cc;
  // operand ring: RING chunks of registers, RING - 1 chunks (2 RING - 2 half-iterations) ahead of the MFMAs; weight ring: NSLOT slots of half a chunk,
  // LEAD half-chunks ahead -- further than the operand ring, because vector-memory operations retire in order: waiting for a weight piece waits for
  // every older operand load.  The eight waves run in lockstep (a barrier per half-chunk): every operand load of every wave has to be on time.
  constexpr int RING = CD_RING, NSLOT = 2 * RING, LEAD = 2 * RING - 1;
  static_assert(108 % RING == 0 && NSLOT * cc::WHALF <= 160 * 1024 && 5 * LEAD - 2 <= 63, "ring geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wbuf = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  const int V = a.v, F = 4 * a.v;
  const int nx8 = 8, xcd = blockIdx.x % nx8, jb = blockIdx.x / nx8, jstride = gridDim.x / nx8;
  const long per = (a.total + nx8 - 1) / nx8;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < a.total) ? tbeg + per : a.total;
  auto w_dma = [&](int h, int slot) {
    const char* src = reinterpret_cast<const char*>(a.Wdp) + (long)h * WHALF;
    char* dst = wbuf + slot * WHALF;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wave * 1024 + lv * 16),
                                     (__attribute__((address_space(3))) void*)(dst + wave * 1024), 16, 0, 0);
    // the ninth image in eight 128-byte slices, one per wave (8 active lanes): every wave issues exactly two pieces per half-chunk, so the counted
    // waits below need no per-wave branch (with one, the register operands tied to the wait are copied around it -- before their data has landed)
    if (lane < 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 8 * 1024 + wave * 128 + lv * 16),
                                       (__attribute__((address_space(3))) void*)(dst + 8 * 1024 + wave * 128), 16, 0, 0);
  };
  const int z_l = wave >> 1, y_l = (wave & 1) * 4, ly = li >> 3, lx = li & 7;
  long t = tbeg + jb;
  if (t >= tend) return;
#pragma unroll
  for (int h0 = 0; h0 < LEAD; ++h0) w_dma(h0, h0);
  // block origin -> this lane's cell (z, y of column tile 0, x) and sample
  struct Org { int b, zc, yc0, xc; };
  auto origin = [&](long tt) {
    const unsigned tu = (unsigned)tt;
    unsigned r1 = tu / (unsigned)a.nbx; const int xb = (int)(tu - r1 * (unsigned)a.nbx);
    unsigned r2 = r1 / (unsigned)a.nby; const int yb = (int)(r1 - r2 * (unsigned)a.nby);
    unsigned r3 = r2 / (unsigned)a.nbz; const int zb = (int)(r2 - r3 * (unsigned)a.nbz);
    Org o;
    o.b = __builtin_amdgcn_readfirstlane((int)r3);
    o.zc = __builtin_amdgcn_readfirstlane(zb * BZ) + z_l;
    o.yc0 = yb * BY + y_l + ly;
    o.xc = xb * BX + lx;
    return o;
  };
  // the operand stream runs three chunks ahead of the MFMAs and straight on into the next block of this workgroup (when there is none: into the
  // same block again -- valid addresses, values unused), so that every half-iteration issues exactly three operand loads: the counted waits rely on it
  long tp = t;
  Org po = origin(tp);
  int gz = 0, gy = 0, iz = 0, iy = 0, ppz = 0, ppy = 0, ps = 0;     // CD_ORDER, generated: offset = group + 4 * index, groups {0,4} {1,5} {2} {3}
  Frag<bf16_t> xr[RING][3][2];
  // Operand addressing, cheap enough for the 648 loads a wave issues per block (64-bit per-lane address arithmetic with its quarter-rate multiplies
  // made the kernel VALU-bound: 1.64 ms): raw buffer loads.  The descriptor's base (scalar, per block and wave) is the first voxel of the window of
  // cell (z, y = 0, x = 0): line (4 z - 1, -1), voxel -1; the lane's offset inside it (per block) is 4 y lines + 4 x voxels + its k-group; the piece's
  // line / k-step offset is a scalar.  Pieces outside the volume get an out-of-range offset: the buffer load returns zeros.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  struct PState { i32x4 rsrc; unsigned voff0, voff1, ymask, xbad; };
  auto pstate = [&](const Org& q) {
    PState r;
    const long base = (long)(size_t)a.dY + 2 * ((long)q.b * F * F * F * 48 + (((long)(4 * q.zc - 1) * F - 1) * F - 1) * 48);
    r.rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base & 0xffffffffL));
    r.rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((base >> 32) & 0xffff));
    r.rsrc[2] = 0x7fffffff;
    r.rsrc[3] = 0x00020000;
    r.voff0 = (unsigned)((4 * q.yc0 * F + 4 * q.xc) * 96 + 16 * g);
    r.voff1 = r.voff0 + (unsigned)(8 * F * 96);
    // bits 0 / 1: column tile 0 sits in the first / last cell line of the volume, bits 2 / 3: column tile 1
    r.ymask = (q.yc0 == 0 ? 1u : 0u) | (q.yc0 == V - 1 ? 2u : 0u) | (q.yc0 + 2 == V - 1 ? 8u : 0u);
    // bit kk set: piece kk of this lane lies in a window voxel outside the line (voxel -1 of the first cell: 4 kk + g < 6; voxel F of the last: >= 30)
    r.xbad = (q.xc == 0 ? (g < 2 ? 0x3u : 0x1u) : 0u) | (q.xc == V - 1 ? (g >= 2 ? 0x180u : 0x100u) : 0u);
    return r;
  };
  PState pq = pstate(po);
  // chunk (p_z, p_y, s) of column tile m -> ring slot: the lane's three 16-byte pieces (k-steps 3 s .. 3 s + 2, k-group g) of the 576-byte window
  auto ld = [&](Frag<bf16_t> (&dst)[3][2], int m) {
    const bool zbad = (unsigned)(4 * po.zc - 1 + ppz) >= (unsigned)F;                    // (scalar)
    const unsigned ysel = (ppy == 0 ? 1u : (ppy == 5 ? 2u : 0u)) << (2 * m);            // (scalar)
    const bool lbad = zbad || (pq.ymask & ysel) != 0u;
    const int soff = (ppz * F + ppy) * F * 96 + 192 * ps;                                // (scalar)
    const unsigned xb = pq.xbad >> (3 * ps);
#pragma unroll
    for (int kl = 0; kl < 3; ++kl) {
      const unsigned vo = (lbad || ((xb >> kl) & 1u)) ? 0xfffffff0u : (m ? pq.voff1 : pq.voff0);
      // always ONE load per piece; raw: the waits are the counted ones in the main loop
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst[kl][m].v) : "v"(vo), "s"(pq.rsrc), "s"(soff + 64 * kl));
    }
  };
  auto advance = [&]() {
    if (++ps == 3) {
      ps = 0;
      if (++iy == (gy < 2 ? 2 : 1)) {
        iy = 0;
        if (++iz == (gz < 2 ? 2 : 1)) {
          iz = 0;
          if (++gy == 4) {
            gy = 0;
            if (++gz == 4) {
              gz = 0;
              tp = tp + jstride < tend ? tp + jstride : tp;
              po = origin(tp);
              pq = pstate(po);
            }
          }
        }
      }
      ppz = gz + 4 * iz; ppy = gy + 4 * iy;
    }
  };
#pragma unroll
  for (int u = 0; u < RING - 1; ++u) {
    ld(xr[u], 0); ld(xr[u], 1);
    advance();
  }
  // the counted waits of the main loop assume its steady-state issue order; the start-up order is different (all weight pieces, then all operand
  // pieces): drain it once
#pragma unroll
  for (int u = 0; u < RING - 1; ++u)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(xr[u][0][0].v), "+v"(xr[u][1][0].v), "+v"(xr[u][2][0].v), "+v"(xr[u][0][1].v), "+v"(xr[u][1][1].v), "+v"(xr[u][2][1].v)
                 :
                 : "memory");
  for (; t < tend; t += jstride) {
    const Org o = origin(t);
    f32x4 acc[6][2];
#pragma unroll
    for (int tt = 0; tt < 6; ++tt)
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[tt][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    int h = 0;
#pragma unroll 1
    for (int it = 0; it < 108 / RING; ++it) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          // In-order retirement, 5 operations per wave and half-iteration (2 weight pieces, 3 operand loads): half-chunk h of the weights (issued LEAD
          // half-iterations ago, in front of that half-iteration's operand loads) has landed once at most 5 LEAD - 2 operations are outstanding; the
          // operand pieces of this chunk (the last of them issued 2 RING - 3 half-iterations back) once at most 5 (2 RING - 4) are: the stricter
          // count in front of a chunk's first half.  The registers the raw loads fill pass through the wait as read-write operands: nothing that
          // uses them can be scheduled in front of it (no per-wave branch around the wait: tied operands would be copied around it -- before
          // their data has landed).
#define CD_WAIT(N)                                                                                                                                      \
  asm volatile("s_waitcnt vmcnt(%6) lgkmcnt(0)"                                                                                                        \
               : "+v"(xr[u][0][0].v), "+v"(xr[u][1][0].v), "+v"(xr[u][2][0].v), "+v"(xr[u][0][1].v), "+v"(xr[u][1][1].v), "+v"(xr[u][2][1].v) : "i"(N) \
               : "memory")
          if (half == 0) CD_WAIT(5 * (2 * RING - 4) < 5 * LEAD - 2 ? 5 * (2 * RING - 4) : 5 * LEAD - 2); else CD_WAIT(5 * LEAD - 2);
#undef CD_WAIT
          __builtin_amdgcn_s_barrier();     // everyone's pieces of half-chunk h are visible; everyone is done with half-chunk h - 1, whose slot is refilled
          {
            const int hn = h + LEAD >= NHALF ? h + LEAD - NHALF : h + LEAD;
            w_dma(hn, (2 * u + half + LEAD) % NSLOT);
          }
          ld(xr[(u + RING - 1) % RING], half);
          const char* wsrc = wbuf + ((2 * u + half) % NSLOT) * WHALF + lane * 16;
          Frag<bf16_t> wf[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) wf[i].v = *reinterpret_cast<const bf16x8*>(wsrc + i * 1024);
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const int f = 9 * half + i, kl = f / 6, tt = f - kl * 6;
#pragma unroll
            for (int m = 0; m < 2; ++m) mma(acc[tt][m], wf[i], xr[u][kl][m]);
          }
          ++h;
        }
        advance();
      }
    }
    // ---- dx of the block: lane (cell li, g) holds channels 24 g .. 24 g + 23 (row 4 g + r of tile tt <-> channel 24 g + 4 tt + r)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const long cell = (((long)o.b * V + o.zc) * V + o.yc0 + 2 * m) * V + o.xc;
      float vv[24];
#pragma unroll
      for (int tt = 0; tt < 6; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) vv[4 * tt + r] = acc[tt][m][r];
      if (a.add) {
        const bf16_t* ap = a.add + cell * 96 + 24 * g;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          float v8[8];
          Vec8<bf16_t>::load(ap + 8 * q, v8);
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[8 * q + j] += v8[j];
        }
      }
      bf16_t* dp = a.DX + cell * 96 + 24 * g;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        *reinterpret_cast<uint4*>(dp + 8 * q) = make_uint4(pk_bf16(vv[8 * q], vv[8 * q + 1]), pk_bf16(vv[8 * q + 2], vv[8 * q + 3]),
                                                           pk_bf16(vv[8 * q + 4], vv[8 * q + 5]), pk_bf16(vv[8 * q + 6], vv[8 * q + 7]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Wdp bf16 [36 (p_z, p_y)][9 k-steps][6 channel tiles][64 lanes][8] from the forward's fragment-ordered composed weights (a gather: the two passes
// use the same bf16 values)
__global__ __launch_bounds__(512) void cconv_dpack_kernel(const bf16_t* __restrict__ Wcp, bf16_t* __restrict__ Wdp) {
  const int blk = blockIdx.x, tid = threadIdx.x;
  const int pos = blk / 54, f = blk - pos * 54, kk = f / 6, tt = f - kk * 6, combo = CD_ORDER[pos], pz = combo / 6, py = combo - pz * 6;
  const int lane = tid >> 3, j = tid & 7, li = lane & 15, g = lane >> 4;
  const int ci = 24 * (li >> 2) + 4 * tt + (li & 3), kap = 32 * kk + 8 * g + j, px = kap / 48, c = kap - px * 48;
  auto an = [](int p, int& aa, int& nn) { if (p == 0) { aa = 3; nn = 1; } else if (p == 5) { aa = 0; nn = -1; } else { aa = p - 1; nn = 0; } };
  int az, nz, ay, ny, ax, nx;
  an(pz, az, nz); an(py, ay, ny); an(px, ax, nx);
  int base = 0;
  const int gi = az * 4 + ay;
  for (int q = 0; q < gi; ++q) base += cc_ncnt(q >> 2) * cc_ncnt(q & 3);
  const int izy = (nz - cc_nfirst(az)) * cc_ncnt(ay) + (ny - cc_nfirst(ay));
  const int k6 = ax == 0 ? (nx < 0 ? 0 : 1) : (ax == 3 ? (nx > 0 ? 5 : 4) : ax + 1);
  const int s = ci >> 5, gf = (ci & 31) >> 3, jf = ci & 7, nt = (c % 12) >> 2, lif = 4 * (c / 12) + (c & 3);
  const long src = ((((long)(base + izy) * 3 + s) * 18 + k6 * 3 + nt) * 64 + 16 * gf + lif) * 8 + jf;
  Wdp[((long)blk * 64 + lane) * 8 + j] = Wcp[src];
  if (blk == 0 && tid < 64) Wdp[CD_NUMEL + tid] = f2bf(0.f);
}

long k_cconv_dpack_numel() { return CD_NUMEL + 64; }

int k_cconv_dpack(const void* Wcp, void* Wdp, hipStream_t st) {
  hipLaunchKernelGGL(cconv_dpack_kernel, dim3(36 * 54), dim3(512), 0, st, (const bf16_t*)Wcp, (bf16_t*)Wdp);
  NMH_CHECK_LAUNCH();
  return 0;
}

// dx [B][v^3][96] = (add ? add : 0) + the composed input gradient of dy1 [B][(4v)^3][48]; add may alias dx
int k_cconv_dgrad(const void* dY, const void* Wdp, const void* add, void* DX, int B, int v, hipStream_t st) {
  using namespace cc;
  if (v % BY || v % BX || v % BZ) return -2;
  CDArgs a;
  a.dY = (const bf16_t*)dY; a.Wdp = (const bf16_t*)Wdp; a.add = (const bf16_t*)add; a.DX = (bf16_t*)DX;
  a.B = B; a.v = v; a.nbz = v / BZ; a.nby = v / BY; a.nbx = v / BX;
  a.total = (long)B * a.nbz * a.nby * a.nbx;
  if (a.total >= (1L << 31) || (long)B * 64 * v * v * v * 48 >= (1L << 40)) return -2;
  long nb = (a.total + 7) / 8 * 8;      // (rounded up: see k_cconv_fwd)
  if (nb > 256) nb = 256;
  static NmhPerDeviceOnce attr;
  if (attr.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)cconv_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CD_RING * WHALF);
    if (e != hipSuccess) return (int)e;
    attr.set();
  }
  hipLaunchKernelGGL(cconv_dgrad_kernel, dim3((unsigned)nb), dim3(512), 2 * CD_RING * WHALF, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}
