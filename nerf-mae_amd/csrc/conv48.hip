// Specialised 3x3x3 convolution for the decoder1 layers (Cin = Cout = 48, bf16, 160^3): the roofline kernel.
//
// One persistent 512-thread workgroup per CU walks 4x8x16 output tiles (XCD-contiguous tile ranges so halo overlap hits
// the same L2).  Per tile:
//   * the 6x10x18x48 input halo sits in LDS (104.7 KB; line stride 1744 B and plane stride 17456 B are = 16 mod 32 so the
//     two 8-lane halves of every ds_read_b128 service group land on complementary bank halves -> conflict-free);
//   * K = 27*48 = 1296 is walked in 41 MFMA k-steps of 32: 36 "row-quad" steps (lane group g reads tap-row 4q+g, all
//     groups the same 8-channel vector c of that row: address = base + rowoff_q[g] + immediate) + 5 steps for tap-row 8;
//     only 2 of 164 slots are padding (1.2 % waste instead of 33 % for channel padding to 64);
//   * weights are pre-packed on the host side in exact B-fragment order [step][ntile][lane][8] and streamed through a
//     2 x 27 KB LDS ring in 5 chunks (global -> registers under the MFMAs -> LDS);
//   * the NEXT tile's halo is prefetched into 13 x 16 B registers per thread while the current tile computes, then
//     written to LDS between tiles (HBM/L2 latency fully hidden, only the ds_write pass is exposed);
//   * epilogue: accumulators -> bf16 -> wave-private slice of the (now free) halo region -> 16-B row stores.
// The same kernel computes input gradients with the flipped/transposed pack.  LDS use 160,032 B of 163,840.
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>

// unsigned division by a launch-invariant divisor (kernels.hpp FDiv; 64-bit div/mod on the VALU costs ~150 instructions each)
__device__ __forceinline__ unsigned c48_fdiv(unsigned n, const FDiv& f) { return f.sh < 0 ? n : (__umulhi(n, f.M) >> f.sh); }

namespace c48 {
constexpr int TZ = 4, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;
constexpr int LINE = HX * 96 + 16, PLANE = HY * LINE + 16, HALO = (TZ + 2) * PLANE;
constexpr int NSTEP = 41, CSTEPS = 9, WCHUNK = CSTEPS * 3 * 1024;
constexpr int LDS_BYTES = HALO + 2 * WCHUNK + 2 * 96 * 4;  // halo, weight ring, statistics accumulators
constexpr int NLINE = (TZ + 2) * HY;         // 60 x-lines in the halo, 108 16-B chunks each
constexpr int LPW = (NLINE + 7) / 8;         // lines per wave (8 waves): 8
constexpr int HREG = 2 * LPW;                // two requests per line: 16
static_assert(HREG == 16 && HX * 6 > 64 && HX * 6 <= 128, "the halo request schedule (4 per weight chunk 0-3) assumes 16 loads per thread, two per line");
static_assert(LINE % 32 == 16 && PLANE % 32 == 16, "bank-half alternation");
static_assert(LDS_BYTES <= 163840, "LDS budget");
}  // namespace c48

struct C48Args {
  const bf16_t* X; const bf16_t* Wk; bf16_t* Y;
  int B, D, H, W, tz, ty, tx;  // tiles per axis
  long total;                  // B*tz*ty*tx (< 2^31: checked at launch)
  FDiv dtx, dty, dtz;
  int accumulate;
  double* stats_acc;           // optional [B][48][2] fp64 accumulators: per-channel sum / sum of squares of the (bf16-rounded) outputs
  // backward-reduce variant (RB: the launch is the input gradient of a conv whose INPUT was lrelu(InstanceNorm(Y1))): stats_acc receives the two
  // sums the InstanceNorm backward needs, sum g and sum g * yhat with g = out * lrelu'(Y1 - mean), yhat = (Y1 - mean) * rstd (nmh_instnorm_bwd_reduce)
  const bf16_t* Y1; const float* stats1; float slope;
  // CZ (with RB): Y1 holds z = lrelu(y1 - mean) instead of y1 (centered decoder1, csrc/cconv.hip): y1 - mean = z > 0 ? z : z / slope; stats1 = (residual mean, rstd)
  float inv_slope;
  long wk_sample_stride;       // plain forward: elements between the weight images of consecutive samples (per-sample scaled weights, k_conv48_pack_scaled); 0 = one image
  // multi-block variant (MB): Cin = 48 ncib, Cout = 48 ncob; a work item is (tile, output block cob, input block cib), cib innermost:
  // the accumulators persist over cib, the epilogue runs after the last one; Wk holds one fragment-ordered image per (cob, cib)
  int ncib, ncob, ldx, ldy;    // ldx / ldy: channels per voxel of X / Y
  int cosplit;                 // MB with few tiles: the unit a workgroup walks is (tile, cob) instead of the tile, so that tiles * ncob
                               // workgroups share the volume (one 40^3 grid has 150 tiles for 256 CUs)
};

__device__ __forceinline__ void c48_tile_origin(const C48Args& a, long t, int& b, int& z0, int& y0, int& x0) {
  const unsigned tu = (unsigned)t;
  const unsigned r1 = c48_fdiv(tu, a.dtx), xt = tu - r1 * (unsigned)a.tx;
  const unsigned r2 = c48_fdiv(r1, a.dty), yt = r1 - r2 * (unsigned)a.ty;
  const unsigned r3 = c48_fdiv(r2, a.dtz), zt = r2 - r3 * (unsigned)a.tz;
  // the tile index is wave-uniform, but the arithmetic above runs on the VALU: hand the results back to SGPRs so that the
  // buffer descriptor built from b is provably uniform (otherwise every buffer_load is wrapped in a waterfall loop: measured
  // 270 cycles per load, 3.5k cycles per tile)
  b = __builtin_amdgcn_readfirstlane((int)r3);
  z0 = __builtin_amdgcn_readfirstlane((int)zt * c48::TZ);
  y0 = __builtin_amdgcn_readfirstlane((int)yt * c48::TY);
  x0 = __builtin_amdgcn_readfirstlane((int)xt * c48::TX);
}

// DBG (diagnostic builds only, NMH_C48_DBG): 1 = no output stores, 2 = no halo prefetch / LDS refill, 4 = no weight DMA and no
// chunk barriers, 8 = no MFMAs (operand traffic only), 16 = no operand reads in the k-loop (MFMAs only).  DBG = 0 is the product.
template <int DBG, bool MB = false, bool RB = false, bool CZ = false>
__global__ __launch_bounds__(512) void conv48_kernel(C48Args a) {
  using namespace c48;
  constexpr long WBLK = (long)NSTEP * 3 * 512;   // elements of one (cob, cib) weight image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;

  // XCD-contiguous tile ranges: block b runs on XCD b%8 (speed only); its tiles are xcd*per + j, j = b/8, stride gridDim/8
  const int nx = 8, xcd = blockIdx.x % nx, jb = blockIdx.x / nx, jstride = gridDim.x / nx;
  const bool split = MB && a.cosplit;   // units are (tile, cob) pairs, cob fastest
  const long total = split ? a.total * a.ncob : a.total;
  const long per = (total + nx - 1) / nx;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < total) ? tbeg + per : total;

  uint4 hreg[HREG];
  const unsigned vox_bytes = MB ? (unsigned)a.ldx * 2u : 96u;
  const unsigned sample_bytes = (unsigned)a.D * a.H * a.W * vox_bytes;  // one sample of X (< 2 GiB: checked at launch)
  // The halo by LINES (round 6): its 60 x-lines of 18 voxels (1728 contiguous bytes = 108 16-byte chunks) are dealt to the waves, line l = wave + 8 j, and a
  // wave fetches a line with two requests (lanes = chunks 0-63, chunks 64-107).  Everything that depends on the line -- its (z, y), the range test, the byte
  // offset in the sample, its LDS row -- is wave-uniform and lives on the scalar unit (the line's offset is the buffer instruction's SGPR offset); the lane part
  // (x range test, byte offset of the chunk in the line) is the same for every line of a tile: two VGPRs per tile.  The request costs no vector instruction,
  // where the chunk-id mapping of rounds 2-5 (13 requests of tid + 512 i) spent ~18 per request on two divisions by multiply-shift, four multiplies and the
  // range tests: ~230 VALU instructions per tile and wave in a kernel whose MFMA stream has no idle issue slots (NMH_C48_DBG=2: the prefetch cost 11 % of it).
  // 16 requests instead of 13 (the second request of a line has 44 live lanes).  The requests of the NEXT tile are dealt one at a time over the k-steps of the
  // current one (a burst backs up the CU's vector-memory path); `bytes` = 0 when there is no next tile: still issued (no branch in the k-loop), reads zero.
  constexpr unsigned OOB = 0x80000000u;   // >= num_records with or without the SGPR offset added (samples are < 2 GiB)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned hv0 = OOB, hv1 = OOB;   // lane offsets of the two requests of a line of the tile IN FLIGHT (OOB: x outside the volume / lane beyond chunk 107)
  auto halo_voff = [&](int x0, int cib) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int c1 = lv + 64;
    const int hxa = (lv * 43) >> 8, hxb = (c1 * 43) >> 8;   // / 6
    const int xa = x0 - 1 + hxa, xb = x0 - 1 + hxb;
    const unsigned cofs = MB ? (unsigned)cib * 96u : 0u;
    hv0 = (unsigned)xa < (unsigned)a.W ? (unsigned)xa * vox_bytes + (unsigned)(lv - hxa * 6) * 16u + cofs : OOB;
    hv1 = (c1 < HX * 6 && (unsigned)xb < (unsigned)a.W) ? (unsigned)xb * vox_bytes + (unsigned)(c1 - hxb * 6) * 16u + cofs : OOB;
  };
  auto halo_gload_one = [&](int i, int b, int z0, int y0, unsigned bytes) {
    const int j = i >> 1, l = wv + 8 * j;
    const int hz = (l * 205) >> 11, hy = l - hz * HY;   // / 10 (l < 64)
    const int z = z0 - 1 + hz, y = y0 - 1 + hy;
    // (bitwise, not short-circuit: the scalar unit then selects instead of branching -- a branch per request cut the k-loop into basic blocks)
    const unsigned lok = (unsigned)(l < NLINE) & (unsigned)((unsigned)z < (unsigned)a.D) & (unsigned)((unsigned)y < (unsigned)a.H);
    const unsigned lmask = 0u - lok;
    // buffer resource over sample b; a line outside the volume (the conv's zero padding) gets an empty one: every lane reads zero without touching memory
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (long)b * (sample_bytes / 2)), 0, (int)(bytes & lmask), 0x00020000);
    const unsigned soff = ((unsigned)((z * a.H + y) * a.W) * vox_bytes) & lmask;
    hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((i & 1) ? hv1 : hv0), (int)soff, 0));
  };
  auto halo_sstore = [&]() {
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
      const int l = wv + 8 * j;
      if (l < NLINE) {
        const int hz = (l * 205) >> 11, hy = l - hz * HY;
        char* const p = halo + hz * PLANE + hy * LINE + lane * 16;
        *reinterpret_cast<uint4*>(p) = hreg[2 * j];
        if (lane < HX * 6 - 64) *reinterpret_cast<uint4*>(p + 1024) = hreg[2 * j + 1];
      }
    }
  };
  // weight chunk ck (steps [9ck, min(9ck+9,41))) -> LDS ring slot `buf` by LDS-DMA: the image is lane-linear
  // (dst = wave-uniform base + lane*16), so no VGPR staging and no ds_write pass; completes before the next barrier.
  auto w_dma = [&](const bf16_t* wblk, int ck, int buf) {
    const int nunits = ((ck < 4) ? CSTEPS : (NSTEP - 4 * CSTEPS)) * 192;  // 16-B units, multiple of 64
    const char* src = reinterpret_cast<const char*>(wblk) + (long)ck * CSTEPS * 3072;
    char* dst = wbuf + buf * WCHUNK;
    for (int u0 = wave * 64; u0 < nunits; u0 += 512) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(u0 + lane) * 16),
                                       (__attribute__((address_space(3))) void*)(dst + u0 * 16), 16, 0, 0);
    }
  };

  auto w_dma_one = [&](const bf16_t* wblk, int ck, int buf, int j) {  // j-th (of <= 4) DMA instruction of this wave for chunk ck
    const int nunits = ((ck < 4) ? CSTEPS : (NSTEP - 4 * CSTEPS)) * 192;
    const int u0 = wave * 64 + 512 * j;
    int lv = lane;
    asm volatile("" : "+v"(lv));  // opaque: the (tile-invariant) 64-bit source addresses would otherwise be hoisted out of the tile loop: 40 VGPRs
    if (u0 < nunits)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(wblk) + (long)ck * CSTEPS * 3072 + (long)(u0 + lv) * 16),
                                       (__attribute__((address_space(3))) void*)(wbuf + buf * WCHUNK + u0 * 16), 16, 0, 0);
  };

  // zero the 16-B pads once (they are only ever multiplied by zero weights, but must not hold NaN patterns)
  for (int i = tid; i < HALO / 16; i += 512) reinterpret_cast<uint4*>(halo)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 192) reinterpret_cast<float*>(smem + HALO + 2 * WCHUNK)[tid] = 0.f;  // statistics accumulators
  __syncthreads();

  long t = tbeg + jb;
  if (t >= tend) return;
  int cb, cz0, cy0, cx0;  // origin of the current tile; the next tile's origin is computed once (64-bit divisions) and carried over
  c48_tile_origin(a, split ? t / a.ncob : t, cb, cz0, cy0, cx0);
  int cob = split ? (int)(t % a.ncob) : 0, cib = 0;   // MB: output / input channel block of the current work item
  const bf16_t* const wfirst = MB ? a.Wk + (long)cob * a.ncib * ((long)NSTEP * 3 * 512) : a.Wk + (long)cb * a.wk_sample_stride;
  halo_voff(cx0, 0);
#pragma unroll
  for (int i = 0; i < HREG; ++i) halo_gload_one(i, cb, cz0, cy0, sample_bytes);
  w_dma(wfirst, 0, 0);
  if (DBG & 4) w_dma(wfirst, 1, 1);
  halo_sstore();
  __syncthreads();
  int wb = 0;  // LDS buffer holding chunk 0 of the current tile

  // per-lane A addressing: m-tile i = x-line (z_l, y_l + i), lanes li = x
  const int z_l = wave >> 1, y_l = (wave & 1) * 4;
  const int base0 = z_l * PLANE + y_l * LINE + li * 96;
  int rowoff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) { const int r = 4 * q + g; rowoff[q] = (r / 3) * PLANE + (r % 3) * LINE; }
  const int row8 = 2 * PLANE + 2 * LINE;

  // fused InstanceNorm statistics: lane (li, g) owns channels 12g + 4n + r of voxel column li.  Per tile the wave's partial sums are
  // folded over its 16 voxel columns (DPP row scan) and added to a workgroup accumulator in LDS (ds_add_f32, [2][48][2] floats: the
  // registers they used to occupy across tiles are what lets the k-loop keep its prefetch depth); the accumulator is flushed with fp64
  // atomics when the workgroup moves to another sample.  Two accumulators alternate, so the flush (threads 0-95, during the first
  // epilogue of the new sample) cannot race with that tile's additions.
  float* const sacc = reinterpret_cast<float*>(smem + HALO + 2 * WCHUNK);
  int st_b = -1, scur = 0;
  auto stats_flush = [&]() {
    if (st_b >= 0 && tid < 96) {
      float v = sacc[scur * 96 + tid];
      const float sacc_even = (RB && CZ) ? __shfl(v, (tid & 63) & ~1, 64) : 0.f;   // sum g of the same channel (the even neighbour lane)
      sacc[scur * 96 + tid] = 0.f;
      if (RB && (tid & 1)) {
        if constexpr (CZ) v -= a.stats1[(long)st_b * 96 + tid - 1] * sacc_even;   // minus (residual mean) x (sum g)
        v *= a.stats1[(long)st_b * 96 + tid];      // sum g (y - mean) -> sum g yhat: rstd of (sample, channel tid / 2)
      }
      atomicAdd(a.stats_acc + (long)st_b * 96 + tid, (double)v);
    }
  };
  auto row_sum = [](float v) -> float {  // inclusive scan over the 16-lane row: lane 15 ends up with the row total
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
  };
  // DBG & 32: per-wave cycle accounting (s_memtime) of the tile phases, written to stats_acc reinterpreted as int64 [block][wave][12]
  long long ph[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) ph[i] = 0;
  long long tlast = 0;
  auto stamp = [&](int slot) {
    if (DBG & 32) {
      const long long now = (long long)__builtin_amdgcn_s_memtime();
      ph[slot] += now - tlast;
      tlast = now;
    }
  };
  if (DBG & 32) tlast = (long long)__builtin_amdgcn_s_memtime();
  const long long tbegin = tlast;
  f32x4 acc[4][3];
  for (;;) {
    // next work item: (t, cob, cib + 1) -> (t, cob + 1, 0) -> (t + jstride, 0, 0); with split units t = (tile, cob):
    // (t, cib + 1) -> (t + jstride, 0)
    long tn = t + jstride;
    int nco = 0, nci = 0;
    if (MB) {
      nci = cib + 1; nco = cob; tn = t;
      if (nci == a.ncib) { nci = 0; ++nco; }
      if (split) {
        if (nci == 0) { tn = t + jstride; nco = (int)(tn % a.ncob); }
      } else if (nco == a.ncob) { nco = 0; tn = t + jstride; }
    }
    const bool has_next = tn < tend;
    const bool pf_next = has_next && !(DBG & 2);
    int nb = cb, nz0 = cz0, ny0 = cy0, nx0 = cx0;
    if (has_next && (!MB || tn != t)) c48_tile_origin(a, split ? tn / a.ncob : tn, nb, nz0, ny0, nx0);
    const unsigned nbytes = pf_next ? sample_bytes : 0u;
    halo_voff(nx0, nci);   // lane offsets of the next tile's requests
    const bf16_t* const wcur = MB ? a.Wk + (long)(cob * a.ncib + cib) * WBLK : a.Wk + (long)cb * a.wk_sample_stride;
    const bf16_t* const wnxt = MB ? a.Wk + (long)(nco * a.ncib + nci) * WBLK : a.Wk + (long)nb * a.wk_sample_stride;
    if (!MB || cib == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // k-steps are software-pipelined inside a weight chunk: the fragments of step s+1 (3 B, linear lane*16; 4 A, one per x-line)
    // are in flight while the 12 MFMAs of step s issue.  aoff(s): LDS byte offset of the A fragment of x-line 0.
    auto a_off = [&](int s) -> int {
      if (s < 36) {
        const int q = s / 18, c = s % 18;
        return base0 + rowoff[q] + (c / 6) * 96 + (c % 6) * 16;
      }
      int c = 4 * (s - 36) + g;
      c = c > 17 ? 17 : c;  // padded slots: weights are zero there, any finite operand will do
      const int c6 = (c * 43) >> 8;
      return base0 + row8 + c6 * 96 + (c - 6 * c6) * 16;
    };
    Frag<bf16_t> bf[2][3], af[2][4];
    auto ld_frags = [&](int buf, const char* wsrc, int sl, int s) {
      const int ao = a_off(s);
#pragma unroll
      for (int n = 0; n < 3; ++n) bf[buf][n].v = *reinterpret_cast<const bf16x8*>(wsrc + (sl * 3 + n) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[buf][i].v = *reinterpret_cast<const bf16x8*>(halo + ao + i * LINE);
    };
#pragma unroll
    for (int ck = 0; ck < 5; ++ck) {
      const int nxt = (ck < 4) ? ck + 1 : 0;
      // VMEM issue is dealt over the k-steps: steps 0-3 carry this wave's (<= 4) DMA instructions for the next weight chunk, later steps
      // the halo requests of the next tile (HSLOT: 4+3+3+3 = 13 over chunks 0-3).  vmcnt retires in order, so "vmcnt(#halo requests of
      // this chunk)" at the chunk barrier means "the DMA and everything older has landed" while the newest requests stay in flight.
      const bool do_dma = !(DBG & 4);  // (the chunk-0 image requested during the last tile is simply never read)
      const char* wsrc = wbuf + ((wb + ck) & 1) * WCHUNK;
      constexpr int NST4 = NSTEP - 4 * CSTEPS;
      const int nst = (ck < 4) ? CSTEPS : NST4;
      const int hbase = 4 * ck;          // first halo request index of this chunk
      const int hcnt = ck < 4 ? 4 : 0;   // (16 = 4 + 4 + 4 + 4 over chunks 0-3)
      if (!(DBG & 16) || ck == 0) ld_frags(0, wsrc, 0, ck * CSTEPS);
      if ((DBG & 16) && ck == 0) ld_frags(1, wsrc, 1, 1);
#pragma unroll
      for (int sl = 0; sl < CSTEPS; ++sl) {
        if (sl < nst) {
          if (!(DBG & 16) && sl + 1 < nst) ld_frags((sl + 1) & 1, wsrc, sl + 1, ck * CSTEPS + sl + 1);
          if (do_dma && sl < 4) w_dma_one(ck < 4 ? wcur : wnxt, nxt, (wb + ck + 1) & 1, sl);
          if (sl >= 4) {
            // chunk 0: steps 4,5,6,8; chunks 1-3: steps 4,6,8
            const int k = (hcnt == 4) ? (sl == 4 ? 0 : sl == 5 ? 1 : sl == 6 ? 2 : sl == 8 ? 3 : -1) : (hcnt == 3) ? (sl == 4 ? 0 : sl == 6 ? 1 : sl == 8 ? 2 : -1) : -1;
            if (k >= 0) halo_gload_one(hbase + k, nb, nz0, ny0, nbytes);
          }
          if (DBG & 8) {
#pragma unroll
            for (int n = 0; n < 3; ++n) asm volatile("" ::"v"(bf[sl & 1][n].v));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[sl & 1][i].v));
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int n = 0; n < 3; ++n) mma(acc[i][n], bf[sl & 1][n], af[sl & 1][i]);  // acc = (W . X^T) tile: rows co, cols voxel
          }
        }
      }
      stamp(ck == 0 ? 0 : ck == 1 ? 2 : 4);
      if (!(DBG & 4)) {
        if (hcnt == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (hcnt == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DBG & 32) stamp(ck == 0 ? 1 : ck == 1 ? 3 : 5);
        __builtin_amdgcn_s_barrier();
      }
      stamp(ck == 0 ? 9 : ck == 1 ? 10 : 11);
    }
    wb ^= 1;  // 5 chunks: chunk 0 of the next tile landed in the other buffer

    // the next tile's halo goes to LDS first (the barrier closing weight chunk 4 guarantees every wave is done reading the old one, its
    // vmcnt(0) that the registers have landed): the 13 ds_write_b128 are only issued here and drain while the epilogue below keeps the
    // VALU and the vector-memory path busy
    if (pf_next) halo_sstore();
    stamp(7);

    // ---- epilogue: transposed accumulators (row 4g + r of co-tile n = channel 12g + 4n + r, col = voxel x = li): every lane owns 12
    //      consecutive channels of one voxel per x-line -> a 16-byte and an 8-byte bf16 store straight from registers, no LDS restaging ----
    if (!MB || cib == a.ncib - 1) {
      const int b = cb, z0 = cz0, y0 = cy0, x0 = cx0;
      const int z = z0 + z_l, x = x0 + li;
      const bool stats = !MB && !(DBG & 32) && a.stats_acc;
      if (stats && b != st_b) {
        stats_flush();
        if (st_b >= 0) scur ^= 1;
        st_b = b;
      }
      // lane (li, g) owns channels 12g .. 12g+11 of voxel x = li (pack: row 4g+r of co-tile n <-> channel 12g + 4n + r).  One 64-bit
      // address per tile, + one row stride per x-line (the per-line form cost two 32-bit multiplies and three v_mad_u64_u32 each); the
      // statistics variant is a separate instantiation so that the plain one carries no accumulator moves
      const int ldy = MB ? a.ldy : 48;
      bf16_t* const dst0 = a.Y + ((((long)b * a.D + z) * a.H + (y0 + y_l)) * a.W + x) * ldy + (MB ? cob * 48 : 0) + 12 * g;
      const long rowstride = (long)a.W * ldy;
      const bool zx_ok = z < a.D && x < a.W && (!(DBG & 1) || a.accumulate == 77);
      auto epilogue = [&](auto with_stats) {
        constexpr bool ST = decltype(with_stats)::value;
        // RB: the matching values of Y1 (24 bytes per line) and the lane's 12 channel means are requested first -- the halo registers are free by now --
        // and used after the four lines have been converted and stored
        uint4 yA[RB ? 4 : 1]; uint2 yB[RB ? 4 : 1]; float4 mu4[RB ? 3 : 1];
        unsigned wk[RB ? 4 : 1][6];
        if constexpr (RB) {
          const bf16_t* const y1p = a.Y1 + (dst0 - a.Y);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (zx_ok && y0 + y_l + i < a.H) {
              yA[i] = *reinterpret_cast<const uint4*>(y1p + i * rowstride);
              yB[i] = *reinterpret_cast<const uint2*>(y1p + i * rowstride + 8);
            }
          // (mean, rstd) pairs of channels 12 g .. 12 g + 11: 24 floats, the means are the even ones
          const float4* sp = reinterpret_cast<const float4*>(a.stats1 + ((long)b * 48 + 12 * g) * 2);
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float4 p0 = sp[2 * q], p1 = sp[2 * q + 1];
            mu4[q] = make_float4(p0.x, p0.z, p1.x, p1.z);
          }
        }
        float st1[ST ? 3 : 1][4], st2[ST ? 3 : 1][4];
        if (ST) {
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st1[ST ? n : 0][r] = 0.f; st2[ST ? n : 0][r] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (zx_ok && y0 + y_l + i < a.H) {
            bf16_t* dst = dst0 + i * rowstride;
            float v[3][4];
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
              for (int r = 0; r < 4; ++r) v[n][r] = acc[i][n][r];
            if (a.accumulate) {
              const uint4 o = *reinterpret_cast<const uint4*>(dst);
              const uint2 o2 = *reinterpret_cast<const uint2*>(dst + 8);
              const unsigned ow[6] = {o.x, o.y, o.z, o.w, o2.x, o2.y};
#pragma unroll
              for (int q = 0; q < 6; ++q) { v[q >> 1][(q & 1) * 2] += __uint_as_float(ow[q] << 16); v[q >> 1][(q & 1) * 2 + 1] += __uint_as_float(ow[q] & 0xffff0000u); }
            }
            unsigned w6[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) w6[q] = pk_bf16(v[q >> 1][(q & 1) * 2], v[q >> 1][(q & 1) * 2 + 1]);
            *reinterpret_cast<uint4*>(dst) = make_uint4(w6[0], w6[1], w6[2], w6[3]);   // (8-byte aligned 16-byte store: dword alignment suffices)
            *reinterpret_cast<uint2*>(dst + 8) = make_uint2(w6[4], w6[5]);
            if constexpr (RB) {
#pragma unroll
              for (int q = 0; q < 6; ++q) wk[i][q] = w6[q];
            }
            if (ST && !RB) {  // statistics of exactly what the normalisation pass will read back
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                const float q0 = __uint_as_float(w6[q] << 16), q1 = __uint_as_float(w6[q] & 0xffff0000u);
                st1[ST ? q >> 1 : 0][(q & 1) * 2] += q0; st1[ST ? q >> 1 : 0][(q & 1) * 2 + 1] += q1;
                st2[ST ? q >> 1 : 0][(q & 1) * 2] += q0 * q0; st2[ST ? q >> 1 : 0][(q & 1) * 2 + 1] += q1 * q1;
              }
            }
          }
        }
        if constexpr (RB) {   // sums of the InstanceNorm backward over the (bf16-rounded) outputs just stored
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (zx_ok && y0 + y_l + i < a.H) {
              const unsigned yw[6] = {yA[i].x, yA[i].y, yA[i].z, yA[i].w, yB[i].x, yB[i].y};
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                const float d0 = __uint_as_float(wk[i][q] << 16), d1 = __uint_as_float(wk[i][q] & 0xffff0000u);
                const float m0 = (q & 1) ? mu4[q >> 1].z : mu4[q >> 1].x, m1 = (q & 1) ? mu4[q >> 1].w : mu4[q >> 1].y;
                const float y0v = __uint_as_float(yw[q] << 16), y1v = __uint_as_float(yw[q] & 0xffff0000u);
                if constexpr (CZ) {
                  // y holds z = lrelu(t), t = y1 - mean: g = d lrelu'(t) has the sign test on z, and g t = d lrelu'(t) t = d z -- no inverse, no mean (the residual
                  // mean of t enters at the flush: sum g (t - e) = sum d z - e sum g)
                  const float g0 = d0 * (y0v > 0.f ? 1.0f : a.slope), g1 = d1 * (y1v > 0.f ? 1.0f : a.slope);
                  st1[q >> 1][(q & 1) * 2] += g0; st1[q >> 1][(q & 1) * 2 + 1] += g1;
                  st2[q >> 1][(q & 1) * 2] += d0 * y0v; st2[q >> 1][(q & 1) * 2 + 1] += d1 * y1v;
                } else {
                const float t0 = y0v - m0, t1 = y1v - m1;
                const float g0 = d0 * (t0 > 0.f ? 1.0f : a.slope), g1 = d1 * (t1 > 0.f ? 1.0f : a.slope);
                st1[q >> 1][(q & 1) * 2] += g0; st1[q >> 1][(q & 1) * 2 + 1] += g1;
                st2[q >> 1][(q & 1) * 2] += g0 * t0; st2[q >> 1][(q & 1) * 2 + 1] += g1 * t1;
                }
              }
            }
        }
        if (ST) {
#pragma unroll
          for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a1 = row_sum(st1[ST ? n : 0][r]), a2 = row_sum(st2[ST ? n : 0][r]);
              if (li == 15) {
                float* dst = sacc + scur * 96 + (12 * g + 4 * n + r) * 2;
                atomicAdd(dst, a1);
                atomicAdd(dst + 1, a2);
              }
            }
        }
      };
      if (stats || RB) epilogue(std::true_type{});
      else epilogue(std::false_type{});
    }
    stamp(6);
    cb = nb; cz0 = nz0; cy0 = ny0; cx0 = nx0;
    __syncthreads();
    stamp(8);
    if (!has_next) break;
    t = tn; cob = nco; cib = nci;
  }
  if (DBG & 32) {
    if (lane == 0) {
      long long* o = reinterpret_cast<long long*>(a.stats_acc) + ((long)blockIdx.x * 8 + wave) * 16;
#pragma unroll
      for (int i = 0; i < 12; ++i) o[i] = ph[i];
      o[12] = (long long)__builtin_amdgcn_s_memtime() - tbegin;
    }
    return;
  }
  if (a.stats_acc) stats_flush();
}


int k_conv48(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int accumulate, double* stats_acc, hipStream_t st,
             const void* Y1, const float* stats1, float slope, long wk_sample_stride, int centered) {
  using namespace c48;
  C48Args a;
  a.Y1 = (const bf16_t*)Y1; a.stats1 = stats1; a.slope = slope;
  a.inv_slope = slope != 0.f ? 1.0f / slope : 0.f; a.wk_sample_stride = wk_sample_stride;
  if (Y1 && (!stats1 || !stats_acc || accumulate)) return -1;
  if ((centered && !Y1) || (wk_sample_stride && (Y1 || accumulate))) return -1;
  a.X = (const bf16_t*)X; a.Wk = (const bf16_t*)Wk; a.Y = (bf16_t*)Y;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.tz = (D + TZ - 1) / TZ; a.ty = (H + TY - 1) / TY; a.tx = (W + TX - 1) / TX;
  a.total = (long)B * a.tz * a.ty * a.tx;
  if (a.total >= (1L << 31) || (long)D * H * W * 96 >= (1L << 31)) return -2;
  a.dtx = make_fdiv((unsigned)a.tx); a.dty = make_fdiv((unsigned)a.ty); a.dtz = make_fdiv((unsigned)a.tz);
  a.accumulate = accumulate;
  a.stats_acc = stats_acc;
  a.ncib = a.ncob = 1; a.ldx = a.ldy = 48; a.cosplit = 0;
  static const int dbg = getenv("NMH_C48_DBG") ? atoi(getenv("NMH_C48_DBG")) : 0;
  if (stats_acc && !(dbg & 32)) {
    hipError_t e = nmh_zero_async(stats_acc, sizeof(double) * 2 * 48 * B, st);
    if (e != hipSuccess) return (int)e;
  }
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv48_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  long nb = a.total < 256 ? ((a.total + 7) / 8 * 8) : 256;
  if (dbg) {  // timing decomposition only: results are wrong by construction
#define C48_DBG_CASE(D) case D: { hipFuncSetAttribute((const void*)conv48_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL(conv48_kernel<D>, dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a); break; }
    switch (dbg) {
      C48_DBG_CASE(1) C48_DBG_CASE(2) C48_DBG_CASE(3) C48_DBG_CASE(4) C48_DBG_CASE(7) C48_DBG_CASE(8) C48_DBG_CASE(15) C48_DBG_CASE(16) C48_DBG_CASE(23) C48_DBG_CASE(32)
      default: return -1;
    }
#undef C48_DBG_CASE
    NMH_CHECK_LAUNCH();
    return 0;
  }
  if (Y1) {
    static NmhPerDeviceOnce attr_rb;
    if (attr_rb.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)conv48_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return (int)e;
      attr_rb.set();
    }
    if (centered) {
      static NmhPerDeviceOnce attr_cz;
      if (attr_cz.need()) {
        hipError_t e = hipFuncSetAttribute((const void*)conv48_kernel<0, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_cz.set();
      }
      hipLaunchKernelGGL((conv48_kernel<0, false, true, true>), dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
    } else hipLaunchKernelGGL((conv48_kernel<0, false, true>), dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(conv48_kernel<0>, dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

// The same kernel on 48-channel blocks: Cin = 48 ncib, Cout = 48 ncob (decoder level 40^3: 96 / 192 channels).  Every (tile, cob, cib) work
// item is one halo fill + one 41-step contraction against its own weight image; the halo of an input block is re-fetched per output block
// (L2 hits).  No fused statistics in this variant.
int k_conv48_mb(const void* X, const void* Wk, void* Y, int B, int D, int H, int W, int Cin, int Cout, int accumulate, hipStream_t st) {
  using namespace c48;
  if (Cin % 48 || Cout % 48 || Cin <= 0 || Cout <= 0) return -2;
  if ((long)D * H * W * Cin * 2 >= (1L << 31)) return -2;   // 32-bit buffer offsets inside one sample, below the out-of-range marker
  C48Args a;
  a.Y1 = nullptr; a.stats1 = nullptr; a.slope = 0.f; a.inv_slope = 0.f; a.wk_sample_stride = 0;
  a.X = (const bf16_t*)X; a.Wk = (const bf16_t*)Wk; a.Y = (bf16_t*)Y;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.tz = (D + TZ - 1) / TZ; a.ty = (H + TY - 1) / TY; a.tx = (W + TX - 1) / TX;
  a.total = (long)B * a.tz * a.ty * a.tx;
  if (a.total >= (1L << 31)) return -2;
  a.dtx = make_fdiv((unsigned)a.tx); a.dty = make_fdiv((unsigned)a.ty); a.dtz = make_fdiv((unsigned)a.tz);
  a.accumulate = accumulate;
  a.stats_acc = nullptr;
  a.ncib = Cin / 48; a.ncob = Cout / 48; a.ldx = Cin; a.ldy = Cout;
  a.cosplit = (a.total < 256 && a.ncob > 1) ? 1 : 0;
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv48_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  const long units = a.cosplit ? a.total * a.ncob : a.total;
  const long nb = units < 256 ? ((units + 7) / 8 * 8) : 256;
  hipLaunchKernelGGL((conv48_kernel<0, true>), dim3((unsigned)nb), dim3(512), LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}

// ================================================================================================================
// Weight gradient of the same layer: dW[co][ci][tap] += sum_v dY[v][co] * X[v + tap][ci]      (Cin = Cout = 48, bf16)
//
// Persistent 1024-thread workgroups (16 waves, 4 per SIMD, <=128 VGPRs) walk 4x4x16 voxel tiles.  Per tile the X halo
// (6x6x18 voxels, 62 KB) and the dY tile (256 voxels, 24 KB, double-buffered by LDS-DMA) sit in LDS exactly as they lie in
// memory ([voxel][channel], 96-B rows), and BOTH MFMA operands are contraction-major, i.e. read with ds_read_b64_tr_b16
// (rows = 8 consecutive x, conflict-free at a 96-B stride).  A k-step is 32 voxels = two adjacent x-lines.  The 27 taps x 3
// ci-tiles = 81 output column blocks are dealt round-robin to the 16 waves (5-6 blocks x 3 co-tiles = <=18 accumulator tiles
// per wave); a tap shift is just a different LDS base address.  Accumulators persist across all tiles of the workgroup and
// are flushed once to a per-workgroup fp32 partial, summed into the PyTorch-layout gradient by a second tiny kernel.
// ================================================================================================================
namespace w48 {
constexpr int TZ = 4, TY = 4, TX = 16, HY = TY + 2, HX = TX + 2;
constexpr int LINE = HX * 96, PLANE = HY * LINE, HALO = (TZ + 2) * PLANE;  // 1728, 10368, 62208
constexpr int DYT = TZ * TY * TX * 96;                                     // 24576
constexpr int LDS_BYTES = HALO + 2 * DYT;
constexpr int HCH = HALO / 16;                                             // 3888 chunks
constexpr int DCH = DYT / 16;                                              // 1536
constexpr int NUNIT = 81, PARTIAL = NUNIT * 3 * 256;                       // 62208 floats per workgroup
}  // namespace w48

__device__ uint4 g_zero16[4];  // zero source for out-of-range LDS-DMA lanes

struct W48Args {
  const bf16_t* X; const bf16_t* dY; float* ws;
  int B, D, H, W, tz, ty, tx;
  long total;
  // channel sub-problems: a Cin x Cout convolution is (Cin/48) x (Cout/48) independent 48 x 48 weight-gradient blocks on strided
  // channel slices of X / dY; blockIdx.y = sub-problem (ci slice fastest).  ldx/ldy = channel counts (row strides in elements)
  int ldx, ldy, nci;
  FDiv dtx, dty, dtz;
};

__device__ __forceinline__ void w48_tile_origin(const W48Args& a, long t, int& b, int& z0, int& y0, int& x0) {
  const unsigned tu = (unsigned)t;
  const unsigned r1 = c48_fdiv(tu, a.dtx), xt = tu - r1 * (unsigned)a.tx;
  const unsigned r2 = c48_fdiv(r1, a.dty), yt = r1 - r2 * (unsigned)a.ty;
  const unsigned r3 = c48_fdiv(r2, a.dtz), zt = r2 - r3 * (unsigned)a.tz;
  b = __builtin_amdgcn_readfirstlane((int)r3);
  z0 = __builtin_amdgcn_readfirstlane((int)zt * w48::TZ);
  y0 = __builtin_amdgcn_readfirstlane((int)yt * w48::TY);
  x0 = __builtin_amdgcn_readfirstlane((int)xt * w48::TX);
}

template <int N, int I = 0, class F> __device__ __forceinline__ void w48_static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    w48_static_for<N, I + 1>(f);
  }
}

template <int NW, int DBG = 0, bool RAW = false>  // waves per workgroup: 16 (<=128 VGPRs, 5 blocks/wave) or 8 (<=256 VGPRs, 10 blocks/wave); RAW: inline-asm transpose reads
__global__ __launch_bounds__(64 * NW) void conv48_wgrad_kernel(W48Args a) {
  using namespace w48;
  constexpr int NT = 64 * NW, HREG = (HCH + NT - 1) / NT, UPW = NUNIT / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* dyb = smem + HALO;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, p = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (gridDim.y == 1) __builtin_amdgcn_s_setprio(3);   // decoder-1's launch: issue priority over the background pass sharing its CUs (csrc/norm.hip: in_bwd_apply_bg_kernel; same-box -0.4 ms)

  // one problem (decoder1): the tile range is cut into 8 XCD-contiguous parts; channel sub-problems: plain striding inside the slice
  const bool multi = gridDim.y > 1;
  const int nx = multi ? 1 : 8, xcd = blockIdx.x % nx, jb = blockIdx.x / nx, jstride = gridDim.x / nx;
  const long per = (a.total + nx - 1) / nx;
  const long tbeg = (long)xcd * per, tend = (tbeg + per < a.total) ? tbeg + per : a.total;
  const int sub = blockIdx.y, cs = sub % a.nci, os = sub / a.nci;
  const bf16_t* Xs = a.X + cs * 48;
  const bf16_t* dYs = a.dY + os * 48;
  const long ldx = a.ldx, ldy = a.ldy;

  const unsigned sample_bytes_x = (unsigned)a.D * a.H * a.W * (unsigned)a.ldx * 2u;  // one sample of X, all channels (< 4 GiB: checked at launch)
  uint4 hreg[HREG];
  // halo request i / dY DMA instruction j of the tile at origin (b, z0, y0, x0).  `on` = false (no next tile) keeps the instruction in
  // the stream (no branches inside the k-loop) but touches no memory.  The requests of the NEXT tile are dealt over the k-steps of the
  // current one: issued as one burst they back up the CU's vector-memory path and every wave waits in front of its MFMAs (see conv48).
  auto halo_gload_one = [&](int i, int b, int z0, int y0, int x0, bool on) {
    int tv = tid;
    asm volatile("" : "+v"(tv));  // opaque: no hoisting of the tile-invariant index math out of the tile loop (VGPR budget)
    const int cid = tv + NT * i;
    const int line = (cid * 4855) >> 19, within = cid - line * (HX * 6);  // /108
    const int hz = (line * 43) >> 8, hy = line - hz * HY, hx = (within * 43) >> 8, c6 = within - hx * 6;  // /6, /6
    const int z = z0 - 1 + hz, y = y0 - 1 + hy, x = x0 - 1 + hx;
    // buffer resource over the channel slice of sample b: 32-bit offsets, and an offset >= num_records reads as zero (the zero padding)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Xs + (long)b * (sample_bytes_x / 2)), 0, on ? (int)sample_bytes_x : 0, 0x00020000);
    const bool ok = cid < HCH && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
    const unsigned off = ok ? (unsigned)(((z * a.H + y) * a.W + x) * (a.ldx * 2) + c6 * 16) : 0xFFFFFFF0u;
    hreg[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  auto halo_sstore = [&]() {
#pragma unroll
    for (int i = 0; i < HREG; ++i) {
      const int cid = tid + NT * i;
      if (cid < HCH) reinterpret_cast<uint4*>(halo)[cid] = hreg[i];  // halo image is dense: chunk id == LDS chunk index
    }
  };
  constexpr int NDMA = (DCH + NT - 1) / NT;  // DMA instructions per wave and tile: 3 (8 waves) or 2 (16 waves, the second one on waves 0-7)
  auto dy_dma_one = [&](int j, int b, int z0, int y0, int x0, int buf, bool on) {
    const int u0 = wave * 64 + NT * j;
    if (u0 < DCH) {
      int lv = lane;
      asm volatile("" : "+v"(lv));
      const int u = u0 + lv, v = (u * 10923) >> 16, c6 = u - v * 6;  // /6 (u < 4096)
      const int line = v >> 4, x = x0 + (v & 15), z = z0 + (line >> 2), y = y0 + (line & 3);
      const void* src = (on && z < a.D && y < a.H && x < a.W)
                            ? (const void*)(dYs + ((((long)b * a.D + z) * a.H + y) * a.W + x) * ldy + c6 * 8)
                            : (const void*)g_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dyb + buf * DYT + u0 * 16), 16, 0, 0);
    }
  };

  f32x4 acc[UPW][3];
#pragma unroll
  for (int i = 0; i < UPW; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's output blocks: unit u = wave + 16*idx -> (tap = u/3, ci-tile = u%3); LDS byte offset of the shifted window
  int uoff[UPW];
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + NW * i, tap = u / 3, cit = u - tap * 3;
    const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;  // already +1 biased
    uoff[i] = dz * PLANE + dy * LINE + dx * 96 + cit * 32;
  }
  // 81 = NW*UPW + 1: the last block (tap 26, ci-tile 2) is split by co-tile over waves 1..3 (one extra accumulator tile each)
  f32x4 accx = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xoff = 2 * PLANE + 2 * LINE + 2 * 96 + 2 * 32;
  const bool has_x = wave >= 1 && wave <= 3;
  const int lane_off = (4 * g + (p >> 2)) * 96 + (p & 3) * 8;  // row (x) and 8-byte column piece supplied by this lane
  const unsigned halo_u = lds_addr_u(halo), dyb_u = lds_addr_u(dyb), dy_lane_off = (unsigned)lane_off;   // (RAW: the dY tile has the same 96-byte rows)
  unsigned u_ad[UPW], x_ad = halo_u + (unsigned)(lane_off + xoff);
#pragma unroll
  for (int i = 0; i < UPW; ++i) u_ad[i] = halo_u + (unsigned)(lane_off + uoff[i]);

  long t = tbeg + jb;
  int cur = 0;
  if (t < tend) {
    int b, z0, y0, x0;
    w48_tile_origin(a, t, b, z0, y0, x0);
#pragma unroll
    for (int j = 0; j < NDMA; ++j) dy_dma_one(j, b, z0, y0, x0, 0, true);
#pragma unroll
    for (int i = 0; i < HREG; ++i) halo_gload_one(i, b, z0, y0, x0, true);
    halo_sstore();
  }
  __syncthreads();
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int slot) {
    if (DBG) { const long long now = (long long)__builtin_amdgcn_s_memtime(); ph[slot] += now - tlast; tlast = now; }
  };
  if (DBG) tlast = (long long)__builtin_amdgcn_s_memtime();
  for (; t < tend; t += jstride) {
    const long tn = t + jstride;
    const bool has_next = tn < tend;
    int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;
    if (has_next) w48_tile_origin(a, tn, nb, nz0, ny0, nx0);
    stamp(0);
    const char* dyc = dyb + cur * DYT;
    unsigned dy_ad[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dy_ad[c] = dyb_u + (unsigned)(cur * DYT + c * 32) + dy_lane_off;
    w48_static_for<8>([&](auto KS) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      // lines 2ks, 2ks+1 of the tile: (z_l, y_l) = (ks>>1, (ks&1)*2) and y_l+1
      const int lbase = (ks >> 1) * PLANE + ((ks & 1) * 2) * LINE + lane_off;
      if constexpr (!RAW) {
      Frag<bf16_t> af[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) af[c] = lds_frag_t(dyc, 96, ks * 32, c * 16, lane, (bf16_t*)nullptr);
      // VMEM schedule of the next tile: DMA instruction ks on the first NDMA steps; halo requests 1,1,1,2,2,1 (8 waves) or 1,1,1,1
      // (16 waves) on steps 0-5, so the last two steps (~2k cycles) give the newest requests time to land before the tile barrier
      if (ks < NDMA) dy_dma_one(ks, nb, nz0, ny0, nx0, cur ^ 1, has_next);
      {
        constexpr int h8[9] = {0, 1, 2, 3, 5, 7, 8, 8, 8}, h4[9] = {0, 1, 2, 3, 4, 4, 4, 4, 4};
        const int h0 = HREG == 8 ? h8[ks] : h4[ks], h1 = HREG == 8 ? h8[ks + 1] : h4[ks + 1];
#pragma unroll
        for (int i = 0; i < HREG; ++i)
          if (i >= h0 && i < h1) halo_gload_one(i, nb, nz0, ny0, nx0, has_next);
      }
#pragma unroll
      for (int i = 0; i < UPW; ++i) {
        const char* pb = halo + lbase + uoff[i];
        bf16x4 lo = ds_read_tr16(pb), hi = ds_read_tr16(pb + LINE);
        Frag<bf16_t> bfr;
        bfr.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int c = 0; c < 3; ++c) mma(acc[i][c], af[c], bfr);
      }
      if (has_x) {
        const char* pb = halo + lbase + xoff;
        bf16x4 lo = ds_read_tr16(pb), hi = ds_read_tr16(pb + LINE);
        Frag<bf16_t> bfr;
        bfr.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if (wave == 1) mma(accx, af[0], bfr);
        else if (wave == 2) mma(accx, af[1], bfr);
        else mma(accx, af[2], bfr);
      }
      } else {
      // RAW (round 5): the builtin transpose read is ordered behind EVERY pending vector-memory operation (`s_waitcnt vmcnt(0)`: hipcc cannot tell the dY
      // buffer / halo being read from the dY buffer being filled by LDS-DMA) -- i.e. behind the next tile's DMA and halo requests issued one k-step earlier,
      // the prefetch this loop exists to hide.  Raw reads (common.hpp), software-pipelined by hand: the operands of unit pair j + 1 are requested before the
      // MFMAs of pair j; lgkmcnt(0) only (scalar loads the compiler may have in flight return out of order, so a counted wait would not be safe).
      constexpr int GU = 2, NG = (UPW + GU - 1) / GU;
      // (the k-step's tile offsets go into the instructions' immediate field: the address registers -- one per unit, three for dY -- are the same in all 8 steps)
      constexpr int LB = (ks >> 1) * PLANE + ((ks & 1) * 2) * LINE, DB = ks * 32 * 96;
      TrFrag fa[3], fb[2][GU], fx;
#pragma unroll
      for (int c = 0; c < 3; ++c) tr_read_raw<16 * 96, DB>(fa[c], dy_ad[c]);
#pragma unroll
      for (int u = 0; u < GU; ++u) tr_read_raw<LINE, LB>(fb[0][u], u_ad[u]);
      if (ks < NDMA) dy_dma_one(ks, nb, nz0, ny0, nx0, cur ^ 1, has_next);
      {
        constexpr int h8[9] = {0, 1, 2, 3, 5, 7, 8, 8, 8}, h4[9] = {0, 1, 2, 3, 4, 4, 4, 4, 4};
        const int h0 = HREG == 8 ? h8[ks] : h4[ks], h1 = HREG == 8 ? h8[ks + 1] : h4[ks + 1];
#pragma unroll
        for (int i = 0; i < HREG; ++i)
          if (i >= h0 && i < h1) halo_gload_one(i, nb, nz0, ny0, nx0, has_next);
      }
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        tr_wait();
        if (j == 0) {
#pragma unroll
          for (int c = 0; c < 3; ++c) tr_pin(fa[c]);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (j * GU + u < UPW) tr_pin(fb[j & 1][u]);
        if (j + 1 < NG) {
#pragma unroll
          for (int u = 0; u < GU; ++u)
            if ((j + 1) * GU + u < UPW) tr_read_raw<LINE, LB>(fb[(j + 1) & 1][u], u_ad[(j + 1) * GU + u]);
        } else if (has_x) tr_read_raw<LINE, LB>(fx, x_ad);
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (j * GU + u < UPW) {
            const Frag<bf16_t> bfr = tr_frag(fb[j & 1][u]);
#pragma unroll
            for (int c = 0; c < 3; ++c) mma(acc[j * GU + u][c], tr_frag(fa[c]), bfr);
          }
      }
      if (has_x) {
        tr_wait();
        tr_pin(fx);
        const Frag<bf16_t> bfr = tr_frag(fx);
        if (wave == 1) mma(accx, tr_frag(fa[0]), bfr);
        else if (wave == 2) mma(accx, tr_frag(fa[1]), bfr);
        else mma(accx, tr_frag(fa[2]), bfr);
      }
      }
    });
    stamp(1);
    __syncthreads();  // everyone is done with halo / dY[cur]; the barrier also drains this wave's DMA + prefetch loads
    stamp(2);
    if (has_next) halo_sstore();
    stamp(3);
    __syncthreads();
    stamp(4);
    cur ^= 1;
  }
  if (DBG) {   // (stamps live behind the partial sums: the accumulators must still be flushed or the MFMAs are dead code)
    if (lane == 0) {
      long long* o = reinterpret_cast<long long*>(a.ws + 256L * PARTIAL) + ((long)blockIdx.x * NW + wave) * 8;
      for (int i = 0; i < 8; ++i) o[i] = ph[i];
    }
  }
  // flush: ws[block][u][ct][row 16][col 16]
  float* wsb = a.ws + ((long)blockIdx.y * gridDim.x + blockIdx.x) * PARTIAL;
#pragma unroll
  for (int i = 0; i < UPW; ++i) {
    const int u = wave + NW * i;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) wsb[(u * 3 + c) * 256 + (4 * g + r) * 16 + p] = acc[i][c][r];
  }
  if (has_x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) wsb[(80 * 3 + (wave - 1)) * 256 + (4 * g + r) * 16 + p] = accx[r];
  }
}

// dW[(co*Cin+ci)*27+tap] += sum_blocks ws[sub][block][u = tap*3+cit][ct][row][col], co = os*48+ct*16+row, ci = cs*48+cit*16+col
__global__ void conv48_wgrad_reduce_kernel(const float* ws, float* dW, int nblocks, int nci, int Cin, const float* stats = nullptr, int B = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w48::PARTIAL) return;
  const int sub = blockIdx.y, cs = sub % nci, os = sub / nci;
  // 8 independent partial sums: the loads of one thread are 249 KB apart, so memory-level parallelism has to come from unrolling
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* src = ws + (long)sub * nblocks * w48::PARTIAL + i;
  int b = stats ? nblocks : 0;
  for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s8[u] += src[(long)(b + u) * w48::PARTIAL];
  }
  for (; b < nblocks; ++b) s8[0] += src[(long)b * w48::PARTIAL];
  float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  const int col = i & 15, row = (i >> 4) & 15, uc = i >> 8, ct = uc % 3, u = uc / 3, tap = u / 3, cit = u - tap * 3;
  if (stats) {
    // scaled form (centered decoder1: the operand was z, the conv's input is rstd[sample][ci] * z): workgroup blk walked tiles of sample ((blk % 8) B) / 8 only
    // (launch_wgrad_halo checks the alignment), so the per-sample scale is applied to its partial -- slot u of the 8-way unrolled sum always sees the same sample
    float sc[8], t8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { sc[u] = stats[((long)((u * B) >> 3) * 48 + cit * 16 + col) * 2 + 1]; t8[u] = 0.f; }
    int bb = 0;
    for (; bb + 8 <= nblocks; bb += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t8[u] += src[(long)(bb + u) * w48::PARTIAL];
    }
    for (; bb < nblocks; ++bb) t8[bb & 7] += src[(long)bb * w48::PARTIAL];
    s = ((t8[0] * sc[0] + t8[1] * sc[1]) + (t8[2] * sc[2] + t8[3] * sc[3])) + ((t8[4] * sc[4] + t8[5] * sc[5]) + (t8[6] * sc[6] + t8[7] * sc[7]));
  }
  dW[((long)(os * 48 + ct * 16 + row) * Cin + cs * 48 + cit * 16 + col) * 27 + tap] += s;
}

long k_conv48_wgrad_ws_floats() { return 256L * w48::PARTIAL + 65536; }  // (+ room for the diagnostic phase counters)

static int launch_wgrad_halo(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st, const float* scale_stats = nullptr) {
  using namespace w48;
  W48Args a;
  a.X = (const bf16_t*)X; a.dY = (const bf16_t*)dY; a.ws = ws;
  a.B = B; a.D = D; a.H = H; a.W = W;
  a.tz = (D + TZ - 1) / TZ; a.ty = (H + TY - 1) / TY; a.tx = (W + TX - 1) / TX;
  a.total = (long)B * a.tz * a.ty * a.tx;
  if (a.total >= (1L << 31)) return -2;
  a.dtx = make_fdiv((unsigned)a.tx); a.dty = make_fdiv((unsigned)a.ty); a.dtz = make_fdiv((unsigned)a.tz);
  a.ldx = Cin; a.ldy = Cout; a.nci = Cin / 48;
  const int nsub = (Cin / 48) * (Cout / 48);
  static NmhPerDeviceOnce attr_set;
  if (attr_set.need()) {
    hipError_t e = hipFuncSetAttribute((const void*)conv48_wgrad_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv48_wgrad_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set.set();
  }
  if (scale_stats) {   // every one of the 8 tile ranges must lie inside ONE sample (the reduce scales a workgroup's partial by its sample's rstd)
    const long tps = (long)a.tz * a.ty * a.tx;
    if (nsub != 1 || B < 1 || 8 % B || tps % (8 / B)) return -2;   // (then B tps is a multiple of 8 and every range is tps B / 8 tiles of one sample)
  }
  int nb;
  if (nsub == 1) nb = a.total < 256 ? (int)((a.total + 7) / 8 * 8) : 256;   // 8 XCD-contiguous tile ranges
  else {
    // the sub-problem launches are the small decoder levels, which run on the weight-gradient queue under the encoder's backward chain: 256 persistent
    // workgroups hold every CU's LDS for the whole launch and stop the chain (a 30-us GEMM of the chain took 280-520 us); on HALF the CUs they take longer
    // but the chain keeps moving -- same-box step 45.0 (256) / 44.9 (224) / 44.7 (192) / 44.5 (128) / 44.4 (96) / 44.9 ms (64 workgroups)
    static const int side_wgs = [] { const char* e = getenv("NMH_W48_SIDE_WGS"); const int v = e ? atoi(e) : 128; return v < 1 ? 1 : (v > 256 ? 256 : v); }();
    nb = side_wgs / nsub;
    if (nb < 1) nb = 1;
    if (nb > a.total) nb = (int)a.total;
  }
  static const int nw = getenv("NMH_W48_WAVES") ? atoi(getenv("NMH_W48_WAVES")) : 8;  // 8 waves x 10 blocks measured 5 % faster than 16 x 5
  static const int wdbg = getenv("NMH_W48_DBG") ? atoi(getenv("NMH_W48_DBG")) : 0;
  if (wdbg) {   // phase cycle counters into ws (int64 [block][wave][8]); no gradient is produced
    hipFuncSetAttribute((const void*)conv48_wgrad_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL((conv48_wgrad_kernel<8, 1>), dim3(nb, nsub), dim3(512), LDS_BYTES, st, a);
    NMH_CHECK_LAUNCH();
    hipLaunchKernelGGL(conv48_wgrad_reduce_kernel, dim3((PARTIAL + 255) / 256, nsub), dim3(256), 0, st, ws, dW, nb, a.nci, Cin);
    NMH_CHECK_LAUNCH();
    return 0;
  }
  static const int raw = getenv("NMH_W48_RAW") ? atoi(getenv("NMH_W48_RAW")) : 1;
  if (nw == 8 && raw) {
    static NmhPerDeviceOnce raw_set;
    if (raw_set.need()) {
      hipError_t e = hipFuncSetAttribute((const void*)conv48_wgrad_kernel<8, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return (int)e;
      raw_set.set();
    }
    hipLaunchKernelGGL((conv48_wgrad_kernel<8, 0, true>), dim3(nb, nsub), dim3(512), LDS_BYTES, st, a);
  } else if (nw == 8) hipLaunchKernelGGL(conv48_wgrad_kernel<8>, dim3(nb, nsub), dim3(512), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(conv48_wgrad_kernel<16>, dim3(nb, nsub), dim3(1024), LDS_BYTES, st, a);
  NMH_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv48_wgrad_reduce_kernel, dim3((PARTIAL + 255) / 256, nsub), dim3(256), 0, st, ws, dW, nb, a.nci, Cin, scale_stats, B);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_conv48_wgrad(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, hipStream_t st, const float* scale_stats) {
  return launch_wgrad_halo(dY, X, dW, ws, B, D, H, W, 48, 48, st, scale_stats);
}

// conv2's forward weights of the centered decoder1, per sample: Wk[b] = fragment-ordered image (pack mode 6) of W[co][ci][tap] * rstd[b][ci] -- the conv's input
// is lrelu(InstanceNorm(y1)) = rstd * z (rstd > 0 commutes with the LeakyReLU), and z is what is stored
__global__ void conv48_pack_scaled_kernel(const float* __restrict__ W, const float* __restrict__ stats, bf16_t* __restrict__ out) {
  constexpr int IMG = c48::NSTEP * 3 * 512;
  const int b = blockIdx.y;
  for (int iw = blockIdx.x * 256 + threadIdx.x; iw < IMG; iw += gridDim.x * 256) {
    const int j = iw & 7, lane = (iw >> 3) & 63, sn = iw >> 9;
    const int nt = sn % 3, st = sn / 3, g = lane >> 4, li = lane & 15, n = 12 * (li >> 2) + 4 * nt + (li & 3);
    int r, c;
    if (st < 36) { r = 4 * (st / 18) + g; c = st % 18; } else { r = 8; c = 4 * (st - 36) + g; }
    float v = 0.f;
    if (c < 18) {
      const int tap = r * 3 + c / 6, k = (c % 6) * 8 + j;
      v = W[((long)n * 48 + k) * 27 + tap] * stats[((long)b * 48 + k) * 2 + 1];
    }
    out[(long)b * IMG + iw] = f2bf(v);
  }
}
int k_conv48_pack_scaled(const float* W, const float* stats, void* out, int B, hipStream_t st) {
  hipLaunchKernelGGL(conv48_pack_scaled_kernel, dim3(62, B), dim3(256), 0, st, W, stats, (bf16_t*)out);
  NMH_CHECK_LAUNCH();
  return 0;
}
// any Cin, Cout that are multiples of 48 (the decoder levels 96..768): (Cin/48)*(Cout/48) sub-problems, <= 256 workgroups in total
int k_conv3_wgrad_halo(const void* dY, const void* X, float* dW, float* ws, int B, int D, int H, int W, int Cin, int Cout, hipStream_t st) {
  if (Cin % 48 || Cout % 48 || (Cin / 48) * (Cout / 48) > 256 || (double)D * H * W * Cin * 2 >= 4294967296.0) return -2;
  return launch_wgrad_halo(dY, X, dW, ws, B, D, H, W, Cin, Cout, st);
}
