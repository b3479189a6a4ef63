// 3x3x3 convolution on 64-channel blocks (bf16): the LDS-halo implicit GEMM of conv48.hip for channel counts that are multiples of 64
// -- swin_b's decoder1 (64 -> 64 at 160^3, BASELINE configs[3]) and the 256 -> 256 convolutions of the nerf_rpn FPN neck.
//
// One persistent 512-thread workgroup per CU walks (output-channel block, 4x4x16 tile) items in XCD-contiguous ranges; per item it loops
// over the input-channel blocks cs = 0..ncs-1 with the accumulators held in registers, so a Cin x Cout layer reads every activation
// row once per output block and writes its output once.  Per (item, cs):
//   * the 6x6x18 x 64-channel input halo sits in LDS in 128-byte voxel rows whose 16-byte chunks are XOR-swizzled by (x & 7): the 16
//     lanes of every ds_read_b128 service group (voxels x .. x+15 of one line, two adjacent chunks) hit 16 distinct bank quads for all
//     three x-taps (checked exhaustively; unswizzled 128-byte rows are 4-way conflicted), 82,944 B;
//   * K = 27*64 = 1728 = 54 MFMA k-steps (tap, channel half) -- no padding slots;
//   * weights are pre-packed in A-fragment order [os][cs][step 54][co-tile 4][lane 64][8] and streamed through a 2 x 36 KB LDS ring in
//     6 chunks of 9 k-steps by LDS-DMA, the DMA instructions dealt over the first k-steps of the previous chunk;
//   * the NEXT (item, cs)'s halo is prefetched into 11 x 16 B registers per thread, one request per k-step slot behind counted vmcnt
//     waits (a burst stalls the CU's vector-memory issue, see conv48.hip), and written to LDS after the last chunk;
//   * swapped MFMA operands (acc = W . X^T): a lane owns 16 consecutive output channels of one voxel per x-line -> two 16-byte stores;
//     InstanceNorm statistics (sum, sum of squares per (sample, channel)) folded per tile as in conv48.
// 8 waves: wave w owns x-lines (z = w>>1, y = 2(w&1) + i), i = 0,1, all four co-tiles: 8 MFMAs per k-step from 2 A + 4 B fragments.
// LDS: 82,944 + 73,728 + 1,024 = 157,696 B of 163,840.
#include "common.hpp"
#include <type_traits>
#include "kernels.hpp"
#include <cstdlib>

__device__ __forceinline__ unsigned c64_fdiv(unsigned n, const FDiv& f) { return f.sh < 0 ? n : (__umulhi(n, f.M) >> f.sh); }

namespace c64 {
constexpr int TZ = 4, TY = 4, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
constexpr int LINE = HX * 128, PLANE = HY * LINE, HALO = HZ * PLANE;   // 2304, 13824, 82944
constexpr int NSTEP = 54, CSTEPS = 9, NCHUNK = 6, WCHUNK = CSTEPS * 4 * 1024;  // 36864
constexpr int WBLOCK = NSTEP * 4 * 64 * 8;   // packed elements per (os, cs) block: 110592
constexpr int LDS_BYTES = HALO + 2 * WCHUNK + 2 * 128 * 4;
constexpr int HCH = HZ * HY * HX * 8;        // 5184 16-byte chunks
constexpr int HREG = (HCH + 511) / 512;      // 11
static_assert(HREG == 11, "halo request schedule assumes 11 loads per thread (2 per weight chunk 0-4, 1 in chunk 5)");
static_assert(LDS_BYTES <= 163840, "LDS budget");
}  // namespace c64

struct C64Args {
  const bf16_t* X; const bf16_t* Wk; bf16_t* Y;
  int B, D, H, W, tz, ty, tx;
  int ldx, ldy;                // channels of X / Y rows (elements)
  int nos, ncs;                // output / input channel blocks of 64
  long tiles, total;           // tiles = B*tz*ty*tx; total = nos * tiles (< 2^31)
  FDiv dtx, dty, dtz, dtiles;
  int accumulate;
  double* stats_acc;           // optional [B][Cout][2]
  const float* bias;           // optional [Cout] fp32, added before the store
};

__device__ __forceinline__ void c64_item(const C64Args& a, long t, int& os, int& b, int& z0, int& y0, int& x0) {
  const unsigned tu = (unsigned)t;
  const unsigned o = c64_fdiv(tu, a.dtiles), tl = tu - o * (unsigned)a.tiles;
  const unsigned r1 = c64_fdiv(tl, a.dtx), xt = tl - r1 * (unsigned)a.tx;
  const unsigned r2 = c64_fdiv(r1, a.dty), yt = r1 - r2 * (unsigned)a.ty;
  const unsigned r3 = c64_fdiv(r2, a.dtz), zt = r2 - r3 * (unsigned)a.tz;
  os = __builtin_amdgcn_readfirstlane((int)o);
  b = __builtin_amdgcn_readfirstlane((int)r3);
  z0 = __builtin_amdgcn_readfirstlane((int)zt * c64::TZ);
  y0 = __builtin_amdgcn_readfirstlane((int)yt * c64::TY);
  x0 = __builtin_amdgcn_readfirstlane((int)xt * c64::TX);
}

__global__ __launch_bounds__(512) void conv64_kernel(C64Args a) {
  using namespace c64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO;
  float* const sacc = reinterpret_cast<float*>(smem + HALO + 2 * WCHUNK);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;

  const int nx = 8, xcd = blockIdx.x % nx, jb = blockIdx.x / nx, jstride = gridDim.x / nx;
  const long per = (a.total + nx - 1) / nx;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < a.total) ? tbeg + per : a.total;

  uint4 hreg[HREG];
  const unsigned sample_bytes = (unsigned)a.D * a.H * a.W * (unsigned)a.ldx * 2u;   // one sample of X, all channels (< 4 GiB: checked at launch)
  // halo request i (of 11) for the item at (b, z0, y0, x0), input-channel block cs: 16-byte chunk cid = tid + 512 i -> (voxel, c8)
  auto halo_gload_one = [&](int i, int b, int cs, int z0, int y0, int x0, unsigned bytes) {
    int tv = tid;
    asm volatile("" : "+v"(tv));  // opaque: keep the index math inside the loop (no long-lived VGPRs)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (long)b * (sample_bytes / 2) + cs * 64), 0, (int)bytes, 0x00020000);
    const int cid = tv + 512 * i;
    const int vox = cid >> 3, c8 = cid & 7;
    const int line = (vox * 3641) >> 16, hx = vox - line * HX;    // /18 (vox < 648)
    const int hz = (line * 43) >> 8, hy = line - hz * HY;          // /6
    const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = cid < HCH && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)(((z * a.H + y) * a.W + x) * (a.ldx * 2) + c8 * 16) : 0xFFFFFFF0u;
    hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  auto halo_sstore = [&]() {
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int i = 0; i < HREG; ++i) {
      const int cid = tv + 512 * i;
      if (cid < HCH) {
        const int vox = cid >> 3, c8 = cid & 7;
        const int line = (vox * 3641) >> 16, hx = vox - line * HX;
        *reinterpret_cast<uint4*>(halo + vox * 128 + ((c8 ^ (hx & 7)) << 4)) = hreg[i];   // (vox*128 = line*LINE + hx*128: the image is dense)
      }
    }
  };
  auto w_dma = [&](const bf16_t* wblk, int ck, int buf) {
    const char* src = reinterpret_cast<const char*>(wblk) + (long)ck * WCHUNK;
    char* dst = wbuf + buf * WCHUNK;
    for (int u0 = wave * 64; u0 < WCHUNK / 16; u0 += 512)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(u0 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + u0 * 16), 16, 0, 0);
  };
  auto w_dma_one = [&](const bf16_t* wblk, int ck, int buf, int j) {  // j-th (of <= 5) DMA instruction of this wave for chunk ck
    const int u0 = wave * 64 + 512 * j;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    if (u0 < WCHUNK / 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(wblk) + (long)ck * WCHUNK + (long)(u0 + lv) * 16),
                                       (__attribute__((address_space(3))) void*)(wbuf + buf * WCHUNK + u0 * 16), 16, 0, 0);
  };

  if (tid < 256) sacc[tid] = 0.f;
  long t = tbeg + jb;
  if (t >= tend) return;
  int cos_, cb, cz0, cy0, cx0;
  c64_item(a, t, cos_, cb, cz0, cy0, cx0);
  int ccs = 0;   // current input-channel block
#pragma unroll
  for (int i = 0; i < HREG; ++i) halo_gload_one(i, cb, 0, cz0, cy0, cx0, sample_bytes);
  w_dma(a.Wk + (long)(cos_ * a.ncs) * WBLOCK, 0, 0);
  halo_sstore();
  __syncthreads();

  // per-lane A addressing: voxel x = li + dx of line (z_l + dz, y_l + i + dy); chunk (half*4 + g) ^ (x & 7)
  const int z_l = wave >> 1, y_l = (wave & 1) * 2;
  const int base0 = z_l * PLANE + y_l * LINE;
  int xoff[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int h = 0; h < 2; ++h) xoff[dx][h] = base0 + (li + dx) * 128 + (((h * 4 + g) ^ ((li + dx) & 7)) << 4);

  int st_b = -1, st_os = 0, scur = 0;
  auto stats_flush = [&]() {
    if (st_b >= 0 && tid < 128) {
      const float v = sacc[scur * 128 + tid];
      sacc[scur * 128 + tid] = 0.f;
      atomicAdd(a.stats_acc + ((long)st_b * (a.nos * 64) + st_os * 64) * 2 + tid, (double)v);
    }
  };
  auto row_sum = [](float v) -> float {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
  };

  f32x4 acc[2][4];
  for (;;) {
    // next (item, cs): the next input-channel block of this tile, or the first block of the next item
    const bool last_cs = ccs + 1 == a.ncs;
    const long tn = last_cs ? t + jstride : t;
    const bool has_next = tn < tend;
    int nos_ = cos_, nb = cb, nz0 = cz0, ny0 = cy0, nx0 = cx0;
    const int ncs_ = last_cs ? 0 : ccs + 1;
    if (last_cs && has_next) c64_item(a, tn, nos_, nb, nz0, ny0, nx0);
    const unsigned nbytes = has_next ? sample_bytes : 0u;
    const bf16_t* wnext = a.Wk + (long)(nos_ * a.ncs + ncs_) * WBLOCK;
    const bf16_t* wcur = a.Wk + (long)(cos_ * a.ncs + ccs) * WBLOCK;
    if (ccs == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    Frag<bf16_t> bf[2][4], af[2][2];
    auto ld_frags = [&](int buf, const char* wsrc, int sl, int s) {
      const int tap = s >> 1, h = s & 1;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
#pragma unroll
      for (int n = 0; n < 4; ++n) bf[buf][n].v = *reinterpret_cast<const bf16x8*>(wsrc + (sl * 4 + n) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) af[buf][i].v = *reinterpret_cast<const bf16x8*>(halo + xoff[dx][h] + dz * PLANE + (dy + i) * LINE);
    };
#pragma unroll
    for (int ck = 0; ck < NCHUNK; ++ck) {
      const char* wsrc = wbuf + (ck & 1) * WCHUNK;
      const int hbase = 2 * ck;                 // first halo request index of this chunk
      const int hcnt = ck < 5 ? 2 : 1;
      ld_frags(0, wsrc, 0, ck * CSTEPS);
#pragma unroll
      for (int sl = 0; sl < CSTEPS; ++sl) {
        if (sl + 1 < CSTEPS) ld_frags((sl + 1) & 1, wsrc, sl + 1, ck * CSTEPS + sl + 1);
        // VMEM issue dealt over the k-steps: steps 0-4 carry this wave's (<= 5) DMA instructions for the next weight chunk (chunk 0 of the
        // next (item, cs) during the last chunk), steps 5 and 7 the halo requests of the next (item, cs)
        if (sl < 5) {
          if (ck + 1 < NCHUNK) w_dma_one(wcur, ck + 1, (ck + 1) & 1, sl);
          else w_dma_one(wnext, 0, 0, sl);
        } else if (sl == 5 || (sl == 7 && hcnt == 2)) {
          halo_gload_one(hbase + (sl == 5 ? 0 : 1), nb, ncs_, nz0, ny0, nx0, nbytes);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 4; ++n) mma(acc[i][n], bf[sl & 1][n], af[sl & 1][i]);  // rows co, cols voxel
      }
      // vmcnt retires in order: "vmcnt(#halo requests of this chunk)" = the DMA (and everything older) has landed
      if (ck < 5) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // every wave is done reading the halo (barrier above) and this wave's prefetch has landed (vmcnt(0)): refill for the next (item, cs)
    if (has_next) halo_sstore();

    if (last_cs) {
      // ---- epilogue: lane (li, g) owns channels 16g .. 16g+15 (row 4g+r of co-tile n <-> channel 16g + 4n + r) of voxel x = li ----
      const int b = cb, z = cz0 + z_l, x = cx0 + li;
      const bool stats = a.stats_acc != nullptr;
      if (stats && (b != st_b || cos_ != st_os)) {
        stats_flush();
        if (st_b >= 0) scur ^= 1;
        st_b = b; st_os = cos_;
      }
      float st1[4][4], st2[4][4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st1[n][r] = 0.f; st2[n][r] = 0.f; }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int y = cy0 + y_l + i;
        if (z < a.D && y < a.H && x < a.W) {
          bf16_t* dst = a.Y + ((((long)b * a.D + z) * a.H + y) * a.W + x) * a.ldy + cos_ * 64 + 16 * g;
          float v[4][4];
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[n][r] = acc[i][n][r];
          if (a.bias) {
            const float4* bp = reinterpret_cast<const float4*>(a.bias + cos_ * 64 + 16 * g);
#pragma unroll
            for (int n = 0; n < 4; ++n) { const float4 bv = bp[n]; v[n][0] += bv.x; v[n][1] += bv.y; v[n][2] += bv.z; v[n][3] += bv.w; }
          }
          if (a.accumulate) {
            const uint4 o0 = *reinterpret_cast<const uint4*>(dst), o1 = *reinterpret_cast<const uint4*>(dst + 8);
            const unsigned ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) { v[q >> 1][(q & 1) * 2] += __uint_as_float(ow[q] << 16); v[q >> 1][(q & 1) * 2 + 1] += __uint_as_float(ow[q] & 0xffff0000u); }
          }
          unsigned w8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) w8[q] = pk_bf16(v[q >> 1][(q & 1) * 2], v[q >> 1][(q & 1) * 2 + 1]);
          *reinterpret_cast<uint4*>(dst) = make_uint4(w8[0], w8[1], w8[2], w8[3]);
          *reinterpret_cast<uint4*>(dst + 8) = make_uint4(w8[4], w8[5], w8[6], w8[7]);
          if (stats) {  // statistics of exactly what the normalisation pass will read back
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float q0 = __uint_as_float(w8[q] << 16), q1 = __uint_as_float(w8[q] & 0xffff0000u);
              st1[q >> 1][(q & 1) * 2] += q0; st1[q >> 1][(q & 1) * 2 + 1] += q1;
              st2[q >> 1][(q & 1) * 2] += q0 * q0; st2[q >> 1][(q & 1) * 2 + 1] += q1 * q1;
            }
          }
        }
      }
      if (stats) {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s1 = row_sum(st1[n][r]), s2 = row_sum(st2[n][r]);
            if (li == 15) {
              float* dst = sacc + scur * 128 + (16 * g + 4 * n + r) * 2;
              atomicAdd(dst, s1);
              atomicAdd(dst + 1, s2);
            }
          }
      }
    }
    if (!has_next) break;
    t = tn; ccs = ncs_;
    cos_ = nos_; cb = nb; cz0 = nz0; cy0 = ny0; cx0 = nx0;
    __syncthreads();
  }
  if (a.stats_acc) {
    __syncthreads();   // every wave's additions of the last tile are in the LDS accumulator
    stats_flush();
  }
}

long k_conv64_pack_numel(int Cin, int Cout) { return (long)(Cin / 64) * (Cout / 64) * c64::WBLOCK; }

int k_conv64(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, double* stats_acc, const float* bias, hipStream_t st) {
  using namespace c64;
  if (Cin % 64 || Cout % 64 || Cin <= 0 || Cout <= 0 || (double)D * H * W * Cin * 2 >= 4294967296.0) return -2;
  C64Args a;
  a.X = (const bf16_t*)X; a.Wk = (const bf16_t*)Wk; a.Y = (bf16_t*)Y;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.tz = (D + TZ - 1) / TZ; a.ty = (H + TY - 1) / TY; a.tx = (W + TX - 1) / TX;
  a.ldx = Cin; a.ldy = Cout; a.nos = Cout / 64; a.ncs = Cin / 64;
  a.tiles = (long)B * a.tz * a.ty * a.tx;
  a.total = a.tiles * a.nos;
  if (a.total >= (1L << 31)) return -2;
  a.dtx = make_fdiv((unsigned)a.tx); a.dty = make_fdiv((unsigned)a.ty); a.dtz = make_fdiv((unsigned)a.tz); a.dtiles = make_fdiv((unsigned)a.tiles);
  a.accumulate = accumulate;
  a.stats_acc = stats_acc;
  a.bias = bias;
  if (stats_acc) {
    hipError_t e = nmh_zero_async(stats_acc, sizeof(double) * 2 * Cout * B, st);
    if (e != hipSuccess) return (int)e;
  }
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  long nb = a.total < 256 ? ((a.total + 7) / 8 * 8) : 256;
  hipLaunchKernelGGL(conv64_kernel, dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================================
// Weight gradient on 64-channel blocks: dW[co][ci][tap] += sum_v dY[v][co] * X[v + tap][ci]   (bf16, Cin, Cout multiples of 64)
//
// The structure of conv48_wgrad_kernel (conv48.hip): persistent 8-wave workgroups walk 4x4x16 voxel tiles; the X halo (6x6x18 voxels,
// 81 KB) and the DMA-double-buffered dY tile (256 voxels, 2 x 32 KB) sit in LDS as they lie in memory ([voxel][64 channels], 128-byte
// rows) and BOTH MFMA operands are read contraction-major with ds_read_b64_tr_b16.  128-byte rows put rows x and x+2 on the same
// banks, so the 32-byte blocks of a row are XORed with (x >> 1) & 3 (x = position in the x-line): the 32 lanes of a transpose-read group
// (8 rows x 4 pieces) then cover all 64 banks exactly once, for the dY tile and for every x-tap shift of the halo.  The 27 taps x 4
// ci-tiles = 108 output column blocks (x 4 co-tiles) do not fit one workgroup's registers: two workgroup groups split the taps
// (14 + 13), 7 column blocks x 4 co-tiles = 28 accumulator tiles per wave.  blockIdx.y = (channel sub-problem, tap group).
// ================================================================================================================
namespace w64 {
constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2;
constexpr int LINE = HX * 128, PLANE = HY * LINE, HALO = (TZ + 2) * PLANE;   // 2304, 13824, 82944
constexpr int DYT = TZ * TY * TX * 128;                                      // 32768
constexpr int LDS_BYTES = HALO + 2 * DYT;                                    // 148480
constexpr int HCH = HALO / 16, DCH = DYT / 16;                               // 5184, 2048
constexpr int NT = 512, HREG = (HCH + NT - 1) / NT, NDMA = DCH / NT;         // 11, 4
constexpr int UPW = 7, UMAX = 56, PARTIAL = UMAX * 4 * 256;                  // units per wave, per group; floats per workgroup
static_assert(HREG == 11 && NDMA == 4, "request schedule");
}  // namespace w64

__device__ uint4 g_zero16_w64[4];
template <int N, int I = 0, class F> __device__ __forceinline__ void w64_static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    w64_static_for<N, I + 1>(f);
  }
}

struct W64Args {
  const bf16_t* X; const bf16_t* dY; float* ws;
  int B, D, H, W, tz, ty, tx;
  long total;
  int ldx, ldy, nci;
  FDiv dtx, dty, dtz;
};

__global__ __launch_bounds__(512) void conv64_wgrad_kernel(W64Args a) {
  using namespace w64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* dyb = smem + HALO;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = blockIdx.y & 1, sub = blockIdx.y >> 1, cs = sub % a.nci, os = sub / a.nci;
  const int t0 = grp * 14, nunit = (grp ? 13 : 14) * 4;
  const bf16_t* Xs = a.X + cs * 64;
  const bf16_t* dYs = a.dY + os * 64;
  const long ldy = a.ldy;
  const unsigned sample_bytes_x = (unsigned)a.D * a.H * a.W * (unsigned)a.ldx * 2u;

  uint4 hreg[HREG];
  auto tile_origin = [&](long t, int& b, int& z0, int& y0, int& x0) {
    const unsigned tu = (unsigned)t;
    const unsigned r1 = c64_fdiv(tu, a.dtx), xt = tu - r1 * (unsigned)a.tx;
    const unsigned r2 = c64_fdiv(r1, a.dty), yt = r1 - r2 * (unsigned)a.ty;
    const unsigned r3 = c64_fdiv(r2, a.dtz), zt = r2 - r3 * (unsigned)a.tz;
    b = __builtin_amdgcn_readfirstlane((int)r3);
    z0 = __builtin_amdgcn_readfirstlane((int)zt * TZ);
    y0 = __builtin_amdgcn_readfirstlane((int)yt * TY);
    x0 = __builtin_amdgcn_readfirstlane((int)xt * TX);
  };
  auto halo_gload_one = [&](int i, int b, int z0, int y0, int x0, bool on) {
    int tv = tid;
    asm volatile("" : "+v"(tv));
    const int cid = tv + NT * i;
    const int vox = cid >> 3, c8 = cid & 7;
    const int line = (vox * 3641) >> 16, hx = vox - line * HX;   // /18
    const int hz = (line * 43) >> 8, hy = line - hz * HY;         // /6
    const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Xs + (long)b * (sample_bytes_x / 2)), 0, on ? (int)sample_bytes_x : 0, 0x00020000);
    const bool ok = cid < HCH && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)(((z * a.H + y) * a.W + x) * (a.ldx * 2) + c8 * 16) : 0xFFFFFFF0u;
    hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  auto halo_sstore = [&]() {
#pragma unroll
    for (int i = 0; i < HREG; ++i) {
      const int cid = tid + NT * i;
      if (cid < HCH) {
        const int vox = cid >> 3, c8 = cid & 7;
        const int line = (vox * 3641) >> 16, hx = vox - line * HX;
        *reinterpret_cast<uint4*>(halo + vox * 128 + ((c8 ^ (((hx >> 1) & 3) << 1)) << 4)) = hreg[i];
      }
    }
  };
  // dY tile by LDS-DMA: unit u (16 B) of the dense image = voxel u>>3 (x = voxel & 15), physical chunk u&7 holding logical chunk ^ 2*((x>>1)&3)
  auto dy_dma_one = [&](int j, int b, int z0, int y0, int x0, int buf, bool on) {
    const int u0 = wave * 64 + NT * j;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int u = u0 + lv, v = u >> 3, xx = v & 15, c8 = (u & 7) ^ (((xx >> 1) & 3) << 1);
    const int line = v >> 4, x = x0 + xx, z = z0 + (line >> 2), y = y0 + (line & 3);
    const void* src = (on && z < a.D && y < a.H && x < a.W)
                          ? (const void*)(dYs + ((((long)b * a.D + z) * a.H + y) * a.W + x) * ldy + c8 * 8)
                          : (const void*)g_zero16_w64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(dyb + buf * DYT + u0 * 16), 16, 0, 0);
  };

  f32x4 acc[UPW][4];
#pragma unroll
  for (int i = 0; i < UPW; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's output blocks: unit u = wave + 8*idx -> (tap = t0 + u/4, ci-tile = u%4); per-lane LDS offset of the shifted, swizzled window
  const int xr = 4 * g + (p >> 2);              // x position (row of the transpose read) supplied by this lane
  int uoff[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    int u = wave + 8 * i;
    if (u >= nunit) u = nunit - 1;             // (clamped units are computed but never flushed)
    const int tap = t0 + (u >> 2), cit = u & 3;
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;  // already +1 biased
    const int hx = xr + dx;
    uoff[i] = dz * PLANE + dy * LINE + hx * 128 + (((cit ^ ((hx >> 1) & 3)) << 5) + ((p & 3) << 3));
  }
  int offA[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) offA[c] = xr * 128 + (((c ^ ((xr >> 1) & 3)) << 5) + ((p & 3) << 3));

  const unsigned halo_u = lds_addr_u(halo), dyb_u = lds_addr_u(dyb);
  unsigned u_ad[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) u_ad[i] = halo_u + (unsigned)uoff[i];
  const int nbx = gridDim.x;
  long t = blockIdx.x;
  int cur = 0;
  if (t < a.total) {
    int b, z0, y0, x0;
    tile_origin(t, b, z0, y0, x0);
#pragma unroll
    for (int j = 0; j < NDMA; ++j) dy_dma_one(j, b, z0, y0, x0, 0, true);
#pragma unroll
    for (int i = 0; i < HREG; ++i) halo_gload_one(i, b, z0, y0, x0, true);
    halo_sstore();
  }
  __syncthreads();
  for (; t < a.total; t += nbx) {
    const long tn = t + nbx;
    const bool has_next = tn < a.total;
    int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;
    if (has_next) tile_origin(tn, nb, nz0, ny0, nx0);
    const char* dyc = dyb + cur * DYT;
    // Raw transpose reads (common.hpp; round 5, as conv48_wgrad_kernel): the builtin read is ordered behind EVERY pending vector-memory operation
    // (`s_waitcnt vmcnt(0)`), i.e. behind the next tile's dY DMA and halo requests issued one k-step earlier -- the prefetch this loop exists to hide.
    // The k-step's tile offsets are instruction immediates; the operands of unit pair j + 1 are requested before the MFMAs of pair j.
    unsigned dy_ad[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dy_ad[c] = dyb_u + (unsigned)(cur * DYT + offA[c]);
    w64_static_for<8>([&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      // lines 2ks, 2ks+1 of the tile: (z_l, y_l) = (ks>>1, (ks&1)*2) and y_l+1
      constexpr int LB = (ks >> 1) * PLANE + ((ks & 1) * 2) * LINE, DB = ks * 32 * 128;
      constexpr int GU = 2, NG = (UPW + GU - 1) / GU;
      TrFrag fa[4], fb[2][GU];
#pragma unroll
      for (int c = 0; c < 4; ++c) tr_read_raw<16 * 128, DB>(fa[c], dy_ad[c]);
#pragma unroll
      for (int u = 0; u < GU; ++u) tr_read_raw<LINE, LB>(fb[0][u], u_ad[u]);
      if (ks < NDMA) dy_dma_one(ks, nb, nz0, ny0, nx0, cur ^ 1, has_next);
      {
        constexpr int hs[9] = {0, 2, 4, 6, 8, 10, 11, 11, 11};
#pragma unroll
        for (int i = 0; i < HREG; ++i)
          if (i >= hs[ks] && i < hs[ks + 1]) halo_gload_one(i, nb, nz0, ny0, nx0, has_next);
      }
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        tr_wait();
        if (j == 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) tr_pin(fa[c]);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (j * GU + u < UPW) tr_pin(fb[j & 1][u]);
        if (j + 1 < NG) {
#pragma unroll
          for (int u = 0; u < GU; ++u)
            if ((j + 1) * GU + u < UPW) tr_read_raw<LINE, LB>(fb[(j + 1) & 1][u], u_ad[(j + 1) * GU + u]);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (j * GU + u < UPW) {
            const Frag<bf16_t> bfr = tr_frag(fb[j & 1][u]);
#pragma unroll
            for (int c = 0; c < 4; ++c) mma(acc[j * GU + u][c], tr_frag(fa[c]), bfr);
          }
      }
    });
    __syncthreads();  // everyone is done with halo / dY[cur]; the barrier also drains this wave's DMA + prefetch loads
    if (has_next) halo_sstore();
    __syncthreads();
    cur ^= 1;
  }
  // flush: ws[(blockIdx.y * gridDim.x + blockIdx.x)][u][ct][row 16][col 16]
  float* wsb = a.ws + ((long)blockIdx.y * gridDim.x + blockIdx.x) * PARTIAL;
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + 8 * i;
    if (u < nunit) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) wsb[(u * 4 + c) * 256 + (4 * g + r) * 16 + p] = acc[i][c][r];
    }
  }
}

// dW[(co*Cin+ci)*27+tap] += sum_blocks ws[y][block][u][ct][row][col], y = 2*sub + grp, tap = 14*grp + u/4, co = os*64+ct*16+row, ci = cs*64+(u%4)*16+col
__global__ void conv64_wgrad_reduce_kernel(const float* ws, float* dW, int nblocks, int nci, int Cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int grp = blockIdx.y & 1, sub = blockIdx.y >> 1, cs = sub % nci, os = sub / nci;
  const int nunit = (grp ? 13 : 14) * 4;
  if (i >= nunit * 4 * 256) return;
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* src = ws + (long)blockIdx.y * nblocks * w64::PARTIAL + i;
  int b = 0;
  for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += src[(long)(b + u) * w64::PARTIAL];
  }
  for (; b < nblocks; ++b) s8[0] += src[(long)b * w64::PARTIAL];
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  const int col = i & 15, row = (i >> 4) & 15, uc = i >> 8, ct = uc & 3, u = uc >> 2, tap = 14 * grp + (u >> 2), cit = u & 3;
  dW[((long)(os * 64 + ct * 16 + row) * Cin + cs * 64 + cit * 16 + col) * 27 + tap] += s;
}

long k_conv64_wgrad_ws_floats() { return 512L * w64::PARTIAL; }

int k_conv64_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st) {
  using namespace w64;
  const int nsub = (Cin / 64) * (Cout / 64);
  if (Cin % 64 || Cout % 64 || nsub < 1 || nsub > 128 || (double)D * H * W * Cin * 2 >= 4294967296.0) return -2;
  W64Args a;
  a.X = (const bf16_t*)X; a.dY = (const bf16_t*)dY; a.ws = ws;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.tz = (D + TZ - 1) / TZ; a.ty = (H + TY - 1) / TY; a.tx = (W + TX - 1) / TX;
  a.total = (long)B * a.tz * a.ty * a.tx;
  if (a.total >= (1L << 31)) return -2;
  a.dtx = make_fdiv((unsigned)a.tx); a.dty = make_fdiv((unsigned)a.ty); a.dtz = make_fdiv((unsigned)a.tz);
  a.ldx = Cin; a.ldy = Cout; a.nci = Cin / 64;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv64_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  int nb = 256 / (2 * nsub);                 // <= 512 workgroups in total (two per CU never co-reside: 145 KB of LDS each)
  if (nb < 1) nb = 1;
  if (nb > a.total) nb = (int)a.total;
  hipLaunchKernelGGL(conv64_wgrad_kernel, dim3(nb, 2 * nsub), dim3(512), LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv64_wgrad_reduce_kernel, dim3((UMAX * 4 * 256 + 255) / 256, 2 * nsub), dim3(256), 0, st, ws, dW, nb, a.nci, Cin);
  NMH_CHECK_LAUNCH();
  return 0;
}
