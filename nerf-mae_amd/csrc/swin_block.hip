// Fused Swin-block kernels for the encoder stages of width C = 96 * NW, NW in {1, 2, 4} (stage 0 / 1 / 2 of swin_t / swin_s), bf16, gfx950.
// SURVEY 2a "K1" + "K2" (reference: swin_mae3d.py:27-197 shifted_window_attention, :352-358 MLP, :366-369 the two residual branches).
//
// One workgroup owns 64 token rows -- a 4x4x4 window for the attention branch, 64 consecutive tokens for the MLP branch -- and keeps them in
// REGISTERS as MFMA operand fragments for the whole branch: NW waves (one per SIMD, the 512-register budget), every wave holds all 64 rows
// (4 row tiles x C / 32 k-steps x 4 registers = 192 registers at C = 384) and owns a slice of the OUTPUT features of every product, so that
//   * a weight fragment is read from LDS by exactly one wave, once, and used for 4 MFMAs (the 4 row tiles);
//   * the weights of a wave are a private, pre-packed, linear STREAM of lane-linear 1-KB fragment images in the order the wave consumes them
//     (swin_pack_kernel builds the streams from the fp32 masters once per step); a wave moves its stream through a private LDS ring with
//     LDS-DMA (global_load_lds) a few steps ahead of its MFMAs under counted vmcnt waits -- no workgroup barrier is involved in the weight
//     path, the waves drift freely (the 64-row unfused GEMMs re-read 120 KB of operands per 4.7 MFLOP tile and are L2 -> LDS bound);
//   * products are formed "transposed" (weights = MFMA A operand, tokens = B operand): a lane then holds, for token `lane & 15`, four
//     consecutive output features, and two such tiles pack into the operand fragment of the NEXT product without leaving the registers
//     (row map np_row below makes the packed fragment come out in natural feature order).
// A stream "step" is always 6 fragments = 24 MFMAs (6 weight tiles x 4 token tiles).
//
// Kernels:
//   swin_attn_fwd_kernel   per window:  gather x (pad -> roll -> partition folded into addressing) -> LN1 -> QKV (per head: the wave that owns
//                          the head computes its Q, K, V and the whole 64 x 64 attention in registers) -> O exchanged through LDS -> proj ->
//                          row scale -> + residual -> x1 scattered to token order.  Saves xnw, qkv, o, lse, mean1 / rstd1 in the layouts of the
//                          unfused path (the weight gradients stay on the grouped path and read them).
//   swin_mlp_fwd_kernel    per 64 tokens: LN2 -> fc1 -> GELU (hidden slices exchanged through LDS, one barrier per 32 NW hidden units) -> fc2
//                          -> row scale -> + residual.  Saves x1n, the fc1 pre-activation, mean2 / rstd2.
// (Round 4 also built the backward of both branches on this scheme -- one wave per SIMD, all 64 rows in registers; each of the three kernels made the
// step slower (+2.5 ... +6.5 ms, DESIGN_HISTORY.md section 11) and round 5 removed them: the backward is the unfused, token-ordered chain of model._BlockFn.)
#include "common.hpp"
#include "kernels.hpp"
#include <cstdlib>
#include <cstdio>
#include <type_traits>

namespace sw {

constexpr int G = 6;                 // fragments per stream step
constexpr int STEP_BYTES = G * 1024;
constexpr int MLP_ROUNDS = 12;       // 4 C hidden units / (32 per wave and round x NW waves) with C = 96 NW
// segments of the forward MLP stream: the rounds of a segment are packed in pipeline order on their own, so that a segment can be walked by its
// own workgroup.  C = 384 at 8 grids is 125 row tiles for 256 CUs: two workgroups per tile, each with half of the hidden units, then fill the chip
__host__ __device__ constexpr int mlp_segments(int NW) { return NW == 4 ? 2 : 1; }

// row map of a PAIR of weight tiles (32 output features): tile t in {0,1}, A-operand row li <-> feature 8 (li >> 2) + 4 t + (li & 3).
// In the C layout (row = 4 g + r) a lane then holds features 8 g + 4 t + r, and pack_tr(tile 0, tile 1) is the natural-order operand
// fragment (slot j of lane group g <-> feature 8 g + j) of the next product.
__host__ __device__ __forceinline__ int np_row(int t, int li) { return 8 * (li >> 2) + 4 * t + (li & 3); }

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
__device__ __forceinline__ Frag<bf16_t> pack_tr(const f32x4& lo, const f32x4& hi) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v u = {pk_bf16(lo[0], lo[1]), pk_bf16(lo[2], lo[3]), pk_bf16(hi[0], hi[1]), pk_bf16(hi[2], hi[3])};
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(bf16x8, u);
  return f;
}
__device__ __forceinline__ uint2 pack4(const f32x4& v) { return make_uint2(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])); }
__device__ __forceinline__ float quad_row_sum(float v) {   // sum over the 4 lanes (g = 0..3) that share row lane & 15
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ void unpack4(const uint2& u, float (&v)[4]) {
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u); v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
}

// raw LDS stores (a compiler-visible LDS store is preceded by `s_waitcnt vmcnt(0)` while an LDS-DMA is in flight: the backend cannot tell
// the regions apart -- that would drain the weight prefetch at every exchange).  The caller orders them with lgk0() before a barrier / a read.
__device__ __forceinline__ void lds_write16(unsigned addr, const bf16x8& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write8(unsigned addr, const uint2& v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write4(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lgk0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wg_barrier() {   // raw barrier: __syncthreads() would also wait for vmcnt(0), i.e. drain the weight prefetch
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

template <int OFF> __device__ __forceinline__ void dma16(const char* src, char* dst) {   // 64 lanes x 16 bytes -> 1 KB of LDS at dst + OFF (+ 16 lane)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, OFF, 0);
}

// ---- a wave's private weight stream: linear in global memory, A steps of it in flight / unread in the wave's LDS ring ----
template <int A, int DBG = 0> struct WStream {
  const char* gsrc;     // wave's stream (global)
  char* ring;           // wave's ring (LDS), A * STEP_BYTES
  int total;            // steps in the stream
  int is_step, is_slot; // next step to request, its slot
  int rd_slot;          // slot of the next step to read
  __device__ __forceinline__ void init(const char* src, char* r, int tot) { gsrc = src; ring = r; total = tot; is_step = 0; is_slot = 0; rd_slot = 0; }
  __device__ __forceinline__ void issue(int lane) {
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const char* s = gsrc + (long)is_step * STEP_BYTES + lv * 16;
    char* d = ring + is_slot * STEP_BYTES;
    if (!(DBG & 1)) {
      // two address / M0 set-ups per step: the instruction's immediate offset advances the global AND the LDS address (13 bits: four pieces per base)
      dma16<0>(s, d); dma16<1024>(s, d); dma16<2048>(s, d); dma16<3072>(s, d);
      dma16<0>(s + 4096, d + 4096); dma16<1024>(s + 4096, d + 4096);
    }
    is_step = is_step + 1 == total ? 0 : is_step + 1;   // behind the end the stream wraps: the in-flight count stays constant (the data is never used)
    is_slot = is_slot + 1 == A ? 0 : is_slot + 1;
  }
  // the oldest requested-and-unread step has landed once at most (A - 1) steps' worth of younger vector-memory operations are outstanding (they
  // retire in order; stores and loads issued in between only make the wait stricter)
  __device__ __forceinline__ void wait() const { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A - 1) * G) : "memory"); }
  // raw ds_reads (volatile asm keeps them where they are written: right behind the vmcnt wait, in front of the step's MFMAs -- left to the
  // scheduler they end up BEHIND the MFMAs with their latency exposed); the data counts as landed only after settle(), which names the six
  // destinations as read-write operands so that no consumer can be scheduled above it
  __device__ __forceinline__ void read(Frag<bf16_t> (&wf)[G], int lane) {
    const unsigned ad = lds_addr(ring) + (unsigned)(rd_slot * STEP_BYTES + lane * 16);
#pragma unroll
    for (int i = 0; i < G; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i].v) : "v"(ad), "n"(i * 1024));
    rd_slot = rd_slot + 1 == A ? 0 : rd_slot + 1;
  }
  static __device__ __forceinline__ void settle(Frag<bf16_t> (&f)[G]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0].v), "+v"(f[1].v), "+v"(f[2].v), "+v"(f[3].v), "+v"(f[4].v), "+v"(f[5].v)::"memory");
  }
  __device__ __forceinline__ void start(Frag<bf16_t> (&wf)[G], int lane) {   // requests steps 0 .. A, leaves step 0 in wf
#pragma unroll
    for (int k = 0; k < A; ++k) issue(lane);
    wait();
    read(wf, lane);
    settle(wf);
    issue(lane);
  }
  // one pipeline step: fetch the fragments of the NEXT step from the ring, run `f` (24 MFMAs) on the current ones, refill the slot just read
  template <class F> __device__ __forceinline__ void step(Frag<bf16_t> (&wf)[G], int lane, F&& f) {
    Frag<bf16_t> nx[G];
    wait();
    read(nx, lane);
    f(wf);
    settle(nx);
    issue(lane);
#pragma unroll
    for (int i = 0; i < G; ++i) wf[i] = nx[i];
  }
  __device__ __forceinline__ void drain() const { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// ------------------------------------------------------------------------------------------------
// weight streams from the fp32 masters
// ------------------------------------------------------------------------------------------------
enum { ST_ATTN_FWD = 0, ST_MLP_FWD = 1 };

struct PackDesc { const float* w0; const float* w1; bf16_t* dst; int type, NW; };
constexpr int MAX_PACK = 96;
struct PackArgs { PackDesc d[MAX_PACK]; };

__host__ __device__ inline int stream_steps(int type, int NW) {
  const int KS = 3 * NW;
  switch (type) {
    case ST_ATTN_FWD: return 4 * KS;
    case ST_MLP_FWD: return 2 * MLP_ROUNDS * NW;
  }
  return 0;
}

// one thread = one lane's 16 bytes of one fragment image
__global__ __launch_bounds__(256) void swin_pack_kernel(PackArgs pa) {
  const PackDesc& d = pa.d[blockIdx.y];
  const int NW = d.NW, KS = 3 * NW, C = 32 * KS;
  const int steps = stream_steps(d.type, NW);
  const long nthr = (long)NW * steps * G * 64;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < nthr; id += (long)gridDim.x * 256) {
    const int lane = (int)(id & 63);
    long q = id >> 6;
    const int i = (int)(q % G); q /= G;
    const int s = (int)(q % steps);
    const int w = (int)(q / steps);
    const int li = lane & 15, g = lane >> 4, t = i & 1, p = i >> 1;
    const float* base = nullptr; long idx0 = 0;
    if (d.type == ST_ATTN_FWD) {
      const int seg = s / KS, ks = s - seg * KS;
      if (seg < 3) { const int h = 3 * w + seg; base = d.w0; idx0 = (long)(p * C + 32 * h + np_row(t, li)) * C + 32 * ks + 8 * g; }          // qkv.weight [3C][C]: part p (Q, K, V)
      else { base = d.w1; idx0 = (long)(96 * w + 32 * p + np_row(t, li)) * C + 32 * ks + 8 * g; }                                          // proj.weight [C][C]
    } else {
      // step order within a segment of R rounds: fc1(0) | { fc1(c + 1), fc2(c) } for c = 0 .. R - 2 | fc2(R - 1); every part NW steps
      // (one segment of 12 rounds, or mlp_segments(NW) of them for the forward stream)
      const int NSEG = mlp_segments(NW), R = MLP_ROUNDS / NSEG, per = 2 * R * NW;
      const int sg = s / per, sl = s - sg * per;
      int c, kind, u;   // kind 0: first product of round c (rows = hidden), 1: second product of round c (rows = channels)
      if (sl < NW) { c = 0; kind = 0; u = sl; }
      else {
        const int s2 = sl - NW, blk = s2 / (2 * NW), r = s2 - blk * 2 * NW;
        if (blk < R - 1) { if (r < NW) { c = blk + 1; kind = 0; u = r; } else { c = blk; kind = 1; u = r - NW; } }
        else { c = R - 1; kind = 1; u = r; }
      }
      c += sg * R;
      const int H = 4 * C;
      if (kind == 0) {      // fragment i: k-step 3 u + (i >> 1), tile t = i & 1 of the wave's hidden pair of the round
        const int hid = 32 * (NW * c + w) + np_row(t, li), k0 = 32 * (3 * u + p) + 8 * g;
        base = d.w0; idx0 = (long)hid * C + k0;                                              // fc1.weight [4C][C]
      } else {              // fragment i: channel pair p, tile t; k-step = the 32 hidden units of wave u in round c
        const int ch = 96 * w + 32 * p + np_row(t, li), h0 = 32 * (NW * c + u) + 8 * g;
        base = d.w1; idx0 = (long)ch * H + h0;                                               // fc2.weight [C][4C]
      }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = base[idx0 + j];
    *reinterpret_cast<bf16x8*>(d.dst + id * 8) = pack8(v);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm of the workgroup's 64 rows into operand fragments held by EVERY wave: af[m][k] <- LN(row 16 m + li)[32 k + 8 g ..+7].
// Row tile m is normalised by wave m % NW only (all waves doing all rows cost 4x the VALU work of the whole prologue), written to the LDS
// tile xt as lane-linear fragment images [k][m] and read back by everyone after a barrier.  rowfn(m) -> {load row, save row, stat row} of this
// lane's row of tile m: load row < 0 = a pad token of the window -> zero fragment (the reference pads AFTER the norm, swin_mae3d.py:62-66);
// the normalised row is also stored to xsave[save row] (the operand of a weight gradient) and its statistics to mean / rstd[stat row].
// Ends with a barrier: xt may be reused at once.
// ------------------------------------------------------------------------------------------------
struct RowIdx { long load, save, stat; };
template <int KS, int NW, class RowFn>
__device__ __forceinline__ void ln_exchange(const bf16_t* __restrict__ x, RowFn&& rowfn, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                            bf16_t* __restrict__ xsave, float* __restrict__ mean_out, float* __restrict__ rstd_out, char* xt,
                                            Frag<bf16_t> (&af)[4][KS], int wave, int lane) {
  constexpr int C = 32 * KS;
  const int g = lane >> 4;
  const unsigned xt_a = lds_addr(xt);
  // the affine parameters of the lane's channels, requested before anything waits (per k-step inside the loop they were KS serial load -> use
  // round trips: 12 x ~1000 cycles at C = 384)
  float4 gmv[KS][2], btv[KS][2];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    gmv[k][0] = *reinterpret_cast<const float4*>(gamma + 32 * k + 8 * g); gmv[k][1] = *reinterpret_cast<const float4*>(gamma + 32 * k + 8 * g + 4);
    btv[k][0] = *reinterpret_cast<const float4*>(beta + 32 * k + 8 * g); btv[k][1] = *reinterpret_cast<const float4*>(beta + 32 * k + 8 * g + 4);
  }
#pragma unroll 1
  for (int m = wave; m < 4; m += NW) {
    const RowIdx ri = rowfn(m);
    const bf16_t* xr = x + (ri.load < 0 ? 0 : ri.load) * C + 8 * g;
    uint4 raw[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) raw[k] = *reinterpret_cast<const uint4*>(xr + 32 * k);
    float xv[KS][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      unpack8(raw[k], xv[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[k][j];
    }
    const float mean = quad_row_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[k][j] - mean; q += d * d; }
    const float rstd = rsqrtf(quad_row_sum(q) * (1.0f / C) + eps);
    if (g == 0 && ri.stat >= 0) { mean_out[ri.stat] = mean; rstd_out[ri.stat] = rstd; }
    bf16_t* const xs = xsave + (ri.save < 0 ? 0 : ri.save) * C + 8 * g;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const float gm[8] = {gmv[k][0].x, gmv[k][0].y, gmv[k][0].z, gmv[k][0].w, gmv[k][1].x, gmv[k][1].y, gmv[k][1].z, gmv[k][1].w};
      const float bt[8] = {btv[k][0].x, btv[k][0].y, btv[k][0].z, btv[k][0].w, btv[k][1].x, btv[k][1].y, btv[k][1].z, btv[k][1].w};
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ri.load < 0 ? 0.f : (xv[k][j] - mean) * rstd * gm[j] + bt[j];
      const bf16x8 f = pack8(o);
      lds_write16(xt_a + (k * 4 + m) * 1024 + lane * 16, f);
      if (ri.save >= 0) *reinterpret_cast<bf16x8*>(xs + 32 * k) = f;
    }
  }
  wg_barrier();
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < KS; ++k) af[m][k].v = *reinterpret_cast<const bf16x8*>(xt + (k * 4 + m) * 1024 + lane * 16);
  wg_barrier();
}

// ================================================================================================
// MLP branch, forward:  x2 = x1 + s_row (gelu(LN2(x1) W1^T + b1) W2^T + b2)
// ================================================================================================
struct MlpFwdArgs {
  const bf16_t* x1; const float* gamma; const float* beta; const char* wstream; const float* b1; const float* b2;
  const float* rowscale; int rows_per_scale;
  bf16_t* x2; bf16_t* x1n; bf16_t* hp; float* mean; float* rstd; long M; float eps; long long* ts; bf16_t* hact;
  float* part; int* cnt;   // SPLIT: fp32 partial outputs [tile][segment][wave][24 tiles][64 lanes][4] and one arrival counter per row tile (zero between launches)
};

template <int NW, int A> constexpr int mlp_lds() { return NW * A * STEP_BYTES + (12 * NW * 1024 > 2 * NW * 4096 ? 12 * NW * 1024 : 2 * NW * 4096) + 4 * 96 * NW * 4; }

// DBG & 8: wave 0 of workgroup 0 records the shader clock at phase boundaries (launchers print the differences)
template <int DBG> __device__ __forceinline__ void stamp(long long* ts, int i) {
  if ((DBG & 8) && ts && blockIdx.x == 0 && threadIdx.x == 0) ts[i] = (long long)__builtin_readcyclecounter();
}
template <int DBG> __device__ __forceinline__ void mmad(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {   // DBG & 2: timing-only builds without the MFMAs
  if (DBG & 2) asm volatile("" ::"v"(a.v), "v"(b.v)); else mma(acc, a, b);
}

// SPLIT: one workgroup per (row tile, segment of the hidden units); the workgroups of a tile leave their fp32 partial products in a.part, and the
// one that arrives last (a.cnt) adds the others' to its own and runs the epilogue.  Block b -> tile 8 (b / 16) + b % 8, segment (b / 8) % 2: the
// two workgroups of a tile are 8 blocks apart (same XCD under the round-robin block placement; correctness does not depend on it: the partials
// and the counter are accessed device-coherently).
template <int NW, int A, int DBG = 0, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW) void swin_mlp_fwd_kernel(MlpFwdArgs a) {
  constexpr int KS = 3 * NW, C = 32 * KS, H = 4 * C, TPS = 4 / NW;   // TPS: row tiles whose activation a step of the first product carries
  constexpr int NSEG = mlp_segments(NW), R = MLP_ROUNDS / NSEG, SEG_STEPS = 2 * R * NW, MYSEG = SPLIT ? 1 : NSEG;
  static_assert(!SPLIT || NSEG == 2, "the block -> (tile, segment) map is written for two segments");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  char* exch = smem + NW * A * STEP_BYTES;     // LayerNorm tile [KS][4] fragment images, then [2][NW k-steps][4 row tiles] images of gelu(h)
  const unsigned exch_a = lds_addr(exch);
  const long tile = SPLIT ? 8 * (long)(blockIdx.x >> 4) + (blockIdx.x & 7) : (long)blockIdx.x;
  const int seg0 = SPLIT ? (blockIdx.x >> 3) & 1 : 0;
  const long rbase = tile * 64;
  if (SPLIT && rbase >= a.M) return;

  stamp<DBG>(a.ts, 0);
  WStream<A, DBG> ws;
  ws.init(a.wstream + ((long)wave * (NSEG * SEG_STEPS) + (long)seg0 * SEG_STEPS) * STEP_BYTES, smem + wave * A * STEP_BYTES, MYSEG * SEG_STEPS);
  Frag<bf16_t> wf[G];
  ws.start(wf, lane);
  stamp<DBG>(a.ts, 1);

  Frag<bf16_t> af[4][KS];
  long rows[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { const long row = rbase + 16 * m + li; rows[m] = row < a.M ? row : -1; }
  // the fc1 bias goes to LDS once: a global load inside the rounds is consumed behind the whole weight prefetch (vector-memory operations retire
  // in order) -- two bias loads per round drained the ring every round (3 k of a round's 8 k cycles)
  float* const sB1 = reinterpret_cast<float*>(exch + (12 * NW * 1024 > 2 * NW * 4096 ? 12 * NW * 1024 : 2 * NW * 4096));
  {
    constexpr int NB = (H / 4 + 64 * NW - 1) / (64 * NW);
    float4 bv[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { const int i = tid + 64 * NW * q; bv[q] = i < H / 4 ? *reinterpret_cast<const float4*>(a.b1 + 4 * i) : float4{0.f, 0.f, 0.f, 0.f}; }
    const unsigned b_a = lds_addr(sB1);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int i = tid + 64 * NW * q;
      if (i < H / 4) { lds_write4(b_a + 16 * i, bv[q].x); lds_write4(b_a + 16 * i + 4, bv[q].y); lds_write4(b_a + 16 * i + 8, bv[q].z); lds_write4(b_a + 16 * i + 12, bv[q].w); }
    }
  }
  ln_exchange<KS, NW>(a.x1, [&](int m) { const long row = rbase + 16 * m + li; const long v = row < a.M && seg0 == 0 ? row : -1; return RowIdx{row < a.M ? row : a.M - 1, v, v}; },
                      a.gamma, a.beta, a.eps, a.x1n, a.mean, a.rstd, exch, af, wave, lane);
  bf16_t* hprow[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) hprow[m] = a.hp + (rows[m] < 0 ? 0 : rows[m]) * H + 32 * wave + 8 * g;

  f32x4 out[6][4];
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) out[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hcur[2][4], hnxt[2][4];

  // bias + store of the pre-activation + GELU of row tile m + hand-over of the wave's 32 hidden units of round c as an operand fragment
  auto act_tile = [&](int c, f32x4 (&h)[2][4], int m, const float4 (&bb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) { h[t][m][0] += bb[t].x; h[t][m][1] += bb[t].y; h[t][m][2] += bb[t].z; h[t][m][3] += bb[t].w; }
    // the two tiles of the pair are 8 consecutive hidden units of the lane's token: ONE 16-byte store (a store instruction costs the same issue
    // time whatever its width -- it touches 16 rows -- and 8-byte stores of the pre-activation were 3 k of a round's 8 k cycles)
    if (rows[m] >= 0) *reinterpret_cast<bf16x8*>(hprow[m] + 32 * NW * c) = pack_tr(h[0][m], h[1][m]).v;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][m][r] = (DBG & 4) ? h[t][m][r] : gelu_fast_f(h[t][m][r]);
    const Frag<bf16_t> hf = pack_tr(h[0][m], h[1][m]);
    lds_write16(exch_a + ((c & 1) * NW + wave) * 4096 + m * 1024 + lane * 16, hf.v);
    if (a.hact && rows[m] >= 0) *reinterpret_cast<bf16x8*>(a.hact + (hprow[m] - a.hp) + 32 * NW * c) = hf.v;   // (only for the unfused backward)
  };
  // first product of a round (into h); with ACT, step u also carries the activation of row tiles u TPS .. of the PREVIOUS round (hp): its
  // ~160 VALU instructions per tile then issue in the shadow of the step's 24 MFMAs instead of behind them
  auto fc1 = [&](f32x4 (&h)[2][4], auto with_act, int c_act, f32x4 (&hp)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < 4; ++m) h[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bb[2];
    if (decltype(with_act)::value) {
#pragma unroll
      for (int t = 0; t < 2; ++t) bb[t] = lds_read_f4_raw(sB1 + 32 * (NW * c_act + wave) + 8 * g + 4 * t);   // (a compiler-visible LDS load here is preceded by vmcnt(0): it drained the weight stream every round)
    }
#pragma unroll
    for (int u = 0; u < NW; ++u)
      ws.step(wf, lane, [&](const Frag<bf16_t> (&w)[G]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m) mmad<DBG>(h[t][m], w[2 * q + t], af[m][3 * u + q]);
        if (decltype(with_act)::value) {
#pragma unroll
          for (int mm = 0; mm < TPS; ++mm) act_tile(c_act, hp, u * TPS + mm, bb);
        }
      });
  };
  auto fc2 = [&](int c) __attribute__((always_inline)) {
    Frag<bf16_t> hf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) hf[m].v = *reinterpret_cast<const bf16x8*>(exch + ((c & 1) * NW) * 4096 + m * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < NW; ++u)
      ws.step(wf, lane, [&](const Frag<bf16_t> (&w)[G]) __attribute__((always_inline)) {
        Frag<bf16_t> hn[4];
        if (u + 1 < NW) {
#pragma unroll
          for (int m = 0; m < 4; ++m) hn[m].v = *reinterpret_cast<const bf16x8*>(exch + ((c & 1) * NW + u + 1) * 4096 + m * 1024 + lane * 16);
        }
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m) mmad<DBG>(out[n][m], w[n], hf[m]);
        if (u + 1 < NW) {
#pragma unroll
          for (int m = 0; m < 4; ++m) hf[m] = hn[m];
        }
      });
  };

  stamp<DBG>(a.ts, 2);
#pragma unroll 1
  for (int sg = 0; sg < MYSEG; ++sg) {
    const int c0 = (seg0 + sg) * R, cl = c0 + R - 1;
    fc1(hcur, std::false_type{}, 0, hcur);
    if (sg == 0) stamp<DBG>(a.ts, 3);
#pragma unroll 1
    for (int c = c0; c < cl; ++c) {
      fc1(hnxt, std::true_type{}, c, hcur);
      if (c == 2) stamp<DBG>(a.ts, 4);
      wg_barrier();
      if (c == 2) stamp<DBG>(a.ts, 5);
      fc2(c);
      if (c == 2) stamp<DBG>(a.ts, 6);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) hcur[t][m] = hnxt[t][m];
    }
    {
      float4 bb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bb[t] = lds_read_f4_raw(sB1 + 32 * (NW * cl + wave) + 8 * g + 4 * t);
#pragma unroll
      for (int m = 0; m < 4; ++m) act_tile(cl, hcur, m, bb);
    }
    wg_barrier();
    fc2(cl);
  }
  stamp<DBG>(a.ts, 7);

  if (SPLIT) {
    // partial products of this segment -> a.part (1 KB per store instruction, lane-linear), then the arrival counter of the tile
    // Device-coherent accesses (sc1: performed at the memory side, past the per-XCD L2s) instead of fences: a release / acquire fence at agent
    // scope is a write-back / invalidate walk of the whole L2 per wave -- 1000 of them at the end of a launch cost more than the split saves
    // (78 vs 52 us).  Order: every wave's stores acknowledged (vmcnt 0) -> workgroup barrier -> counter -> barrier -> loads.
    const char* const pbase = reinterpret_cast<const char*>(a.part) + ((((tile * NSEG) * NW + wave) * 24) * 64 + lane) * 16;
    const long seg_bytes = (long)NW * 24 * 64 * 16;
    {
      // (one asm statement per group of stores, closed by wait states: the compiler recycles the data registers -- copies out of the accumulation
      // registers -- for the next group, and its hazard handling does not look inside asm)
      const char* mine = pbase + seg0 * seg_bytes;
#pragma unroll
      for (int n = 0; n < 6; ++n)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:1024 sc1\n\t"
                     "global_store_dwordx4 %0, %3, off offset:2048 sc1\n\tglobal_store_dwordx4 %0, %4, off offset:3072 sc1\n\ts_nop 3"
                     ::"v"(mine + n * 4096), "v"(out[n][0]), "v"(out[n][1]), "v"(out[n][2]), "v"(out[n][3]) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    volatile int* const flag = reinterpret_cast<volatile int*>(exch);
    if (tid == 0) {
      const int old = atomicAdd(a.cnt + tile, 1);
      if (old == NSEG - 1) atomicExch(a.cnt + tile, 0);   // everyone has arrived: ready for the next launch
      *flag = old == NSEG - 1;
    }
    __syncthreads();
    if (!*flag) { ws.drain(); return; }
    {
      // all 24 loads and their wait in ONE asm statement: between separate statements the compiler may copy a destination register (to make room:
      // 96 + 96 live values here) before the data has arrived -- asm loads are not counted by its waitcnt insertion
      const char* oth = pbase + (seg0 ^ 1) * seg_bytes;
      f32x4 pv[24];
      asm volatile(
          "global_load_dwordx4 %0, %24, off sc1\n\tglobal_load_dwordx4 %1, %24, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %24, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %24, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %4, %25, off sc1\n\tglobal_load_dwordx4 %5, %25, off offset:1024 sc1\n\tglobal_load_dwordx4 %6, %25, off offset:2048 sc1\n\tglobal_load_dwordx4 %7, %25, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %8, %26, off sc1\n\tglobal_load_dwordx4 %9, %26, off offset:1024 sc1\n\tglobal_load_dwordx4 %10, %26, off offset:2048 sc1\n\tglobal_load_dwordx4 %11, %26, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %12, %27, off sc1\n\tglobal_load_dwordx4 %13, %27, off offset:1024 sc1\n\tglobal_load_dwordx4 %14, %27, off offset:2048 sc1\n\tglobal_load_dwordx4 %15, %27, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %16, %28, off sc1\n\tglobal_load_dwordx4 %17, %28, off offset:1024 sc1\n\tglobal_load_dwordx4 %18, %28, off offset:2048 sc1\n\tglobal_load_dwordx4 %19, %28, off offset:3072 sc1\n\t"
          "global_load_dwordx4 %20, %29, off sc1\n\tglobal_load_dwordx4 %21, %29, off offset:1024 sc1\n\tglobal_load_dwordx4 %22, %29, off offset:2048 sc1\n\tglobal_load_dwordx4 %23, %29, off offset:3072 sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(pv[0]), "=&v"(pv[1]), "=&v"(pv[2]), "=&v"(pv[3]), "=&v"(pv[4]), "=&v"(pv[5]), "=&v"(pv[6]), "=&v"(pv[7]), "=&v"(pv[8]), "=&v"(pv[9]), "=&v"(pv[10]), "=&v"(pv[11]),
            "=&v"(pv[12]), "=&v"(pv[13]), "=&v"(pv[14]), "=&v"(pv[15]), "=&v"(pv[16]), "=&v"(pv[17]), "=&v"(pv[18]), "=&v"(pv[19]), "=&v"(pv[20]), "=&v"(pv[21]), "=&v"(pv[22]), "=&v"(pv[23])
          : "v"(oth), "v"(oth + 4096), "v"(oth + 2 * 4096), "v"(oth + 3 * 4096), "v"(oth + 4 * 4096), "v"(oth + 5 * 4096)
          : "memory");
#pragma unroll
      for (int n = 0; n < 6; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) out[n][m] += pv[n * 4 + m];
    }
  }

  // epilogue: lane (li, g) owns channels 96 w + 32 p + 8 g + 4 t + r of token li.  Every load first (24 + 6 + 4 requests in flight), then the stores:
  // written load -> use per piece, the epilogue was 24 serial L2 round trips (15 k of the kernel's 130 k cycles)
  {
    uint4 xres[4][3];
    float4 b2v[6];
    float scv[4];
#pragma unroll
    for (int n = 0; n < 6; ++n) b2v[n] = *reinterpret_cast<const float4*>(a.b2 + 96 * wave + 32 * (n >> 1) + 8 * g + 4 * (n & 1));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long row = rows[m] < 0 ? 0 : rows[m];
      scv[m] = a.rowscale ? a.rowscale[row / a.rows_per_scale] : 1.0f;
#pragma unroll
      for (int p = 0; p < 3; ++p) xres[m][p] = *reinterpret_cast<const uint4*>(a.x1 + row * C + 96 * wave + 32 * p + 8 * g);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (rows[m] < 0) continue;
      bf16_t* const orow = a.x2 + rows[m] * C + 96 * wave + 8 * g;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float xr[8];
        unpack8(xres[m][p], xr);
        f32x4 v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b0 = r == 0 ? b2v[2 * p].x : r == 1 ? b2v[2 * p].y : r == 2 ? b2v[2 * p].z : b2v[2 * p].w;
          const float b1 = r == 0 ? b2v[2 * p + 1].x : r == 1 ? b2v[2 * p + 1].y : r == 2 ? b2v[2 * p + 1].z : b2v[2 * p + 1].w;
          v0[r] = (out[2 * p][m][r] + b0) * scv[m] + xr[r];
          v1[r] = (out[2 * p + 1][m][r] + b1) * scv[m] + xr[4 + r];
        }
        *reinterpret_cast<bf16x8*>(orow + 32 * p) = pack_tr(v0, v1).v;
      }
    }
  }
  stamp<DBG>(a.ts, 8);
  ws.drain();
  stamp<DBG>(a.ts, 9);
}

// ================================================================================================
// attention branch, forward:  x1 = x + s_row * window_reverse(proj(attn(window_partition(LN1(x)))))
// ================================================================================================
struct AttnFwdArgs {
  const bf16_t* x; const float* gamma; const float* beta; const char* wstream; const float* bqkv; const float* table; const float* bproj;
  const float* rowscale; int rows_per_scale;
  bf16_t* xnw; float* mean; float* rstd; bf16_t* qkv; bf16_t* o; float* lse; bf16_t* x1;
  WinMap wm; float eps; long long* ts;
  int tok_saves;   // xnw and o are written in TOKEN order ([T][C], pad rows dropped) -- the operands of token-ordered weight gradients (nmh_window_attn_bwd_tokens)
};

__device__ __forceinline__ int relidx(int i, int j) {
  return ((i >> 4) - (j >> 4) + 3) * 49 + (((i >> 2) & 3) - ((j >> 2) & 3) + 3) * 7 + ((i & 3) - (j & 3) + 3);
}
__device__ __forceinline__ int axis_region(int p, int P, int s) { return s == 0 ? 0 : (p < P - 4 ? 0 : (p < P - s ? 1 : 2)); }
__device__ __forceinline__ int token_region(const WinMap& w, int winl, int t) {
  const int nwy = w.PW >> 2, nwx = w.PD >> 2;
  int wx = winl % nwx, wy = (winl / nwx) % nwy, wz = winl / (nwx * nwy);
  return axis_region(wz * 4 + (t >> 4), w.PH, w.s0) * 9 + axis_region(wy * 4 + ((t >> 2) & 3), w.PW, w.s1) * 3 + axis_region(wx * 4 + (t & 3), w.PD, w.s2);
}

constexpr int VRS = 96;                                  // row stride of the wave's V tile [64 tokens][32 features] (conflict-free transpose reads)
template <int NW, int A> constexpr int attn_fwd_lds() {
  constexpr int scratch = NW * 64 * VRS + 3 * NW * 344 * 4 + 256;   // V tiles, bias table [heads][344], region ids
  constexpr int exch = 3 * NW * 4096;                                 // O fragments [head][row tile]: aliases the scratch after the attention
  return NW * A * STEP_BYTES + (scratch > exch ? scratch : exch);
}

template <int NW, int A, int DBG = 0>
__global__ __launch_bounds__(64 * NW) void swin_attn_fwd_kernel(AttnFwdArgs a) {
  constexpr int KS = 3 * NW, C = 32 * KS, HEADS = KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, li = lane & 15;
  char* scratch = smem + NW * A * STEP_BYTES;
  char* sV = scratch + wave * 64 * VRS;
  float* sTab = reinterpret_cast<float*>(scratch + NW * 64 * VRS);        // [HEADS][344]
  unsigned char* sRb = reinterpret_cast<unsigned char*>(scratch + NW * 64 * VRS + HEADS * 344 * 4);   // region id of the 64 tokens, one byte each
  const long win = blockIdx.x;
  const int nW = (a.wm.PH >> 2) * (a.wm.PW >> 2) * (a.wm.PD >> 2);
  const bool shifted = (a.wm.s0 + a.wm.s1 + a.wm.s2) > 0;

  stamp<DBG>(a.ts, 0);
  WStream<A, DBG> ws;
  ws.init(a.wstream + (long)wave * (4 * KS) * STEP_BYTES, smem + wave * A * STEP_BYTES, 4 * KS);
  Frag<bf16_t> wf[G];
  ws.start(wf, lane);
  stamp<DBG>(a.ts, 1);

  // ---- gather + LN1 (the LayerNorm tile uses the scratch area before the bias table is staged there)
  Frag<bf16_t> xn[4][KS];
  long tok[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) tok[m] = win_to_tok(a.wm, win * 64 + 16 * m + li);
  ln_exchange<KS, NW>(a.x, [&](int m) { const long t = win_to_tok(a.wm, win * 64 + 16 * m + li); return RowIdx{t, a.tok_saves ? t : win * 64 + 16 * m + li, t}; },
                      a.gamma, a.beta, a.eps, a.xnw, a.mean, a.rstd, scratch, xn, wave, lane);
  {
    const unsigned tab_a = lds_addr(sTab), sr_a = lds_addr(sRb);
    // all requests first, then the LDS stores (volatile asm: a load cannot pass one -- load -> store per element was 17 serial round trips)
    constexpr int NT = (HEADS * 343 + 64 * NW - 1) / (64 * NW);
    float tv[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int i = tid + 64 * NW * q; tv[q] = i < HEADS * 343 ? a.table[i] : 0.f; }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int i = tid + 64 * NW * q;
      const int t = i / HEADS, h = i - t * HEADS;
      if (i < HEADS * 343) lds_write4(tab_a + (h * 344 + t) * 4, tv[q]);
    }
    if (tid < 16) {
      unsigned pk = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) pk |= (unsigned)(shifted ? token_region(a.wm, (int)(win % nW), 4 * tid + q) : 0) << (8 * q);
      lds_write4(sr_a + 4 * tid, __uint_as_float(pk));
    }
  }
  wg_barrier();
  stamp<DBG>(a.ts, 2);

  Frag<bf16_t> of[3][4];      // this wave's heads: O fragments (token tile m, k = the head's 32 features)
  const float scale = 0.17677669529663689f;  // 32^-0.5
  const unsigned sV_a = lds_addr(sV);
#pragma unroll 1
  for (int j = 0; j < 3; ++j) {
    const int h = 3 * wave + j;
    Frag<bf16_t> ofj[4];
    f32x4 acc[6][4];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bq[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) bq[i] = *reinterpret_cast<const float4*>(a.bqkv + (i >> 1) * C + 32 * h + 8 * g + 4 * (i & 1));
#pragma unroll
    for (int k = 0; k < KS; ++k)
      ws.step(wf, lane, [&](const Frag<bf16_t> (&w)[G]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int m = 0; m < 4; ++m) mmad<DBG>(acc[i][m], w[i], xn[m][k]);
      });
    if (j == 1) stamp<DBG>(a.ts, 3);
    // bias, stores in the unfused layout qkv[row][3C] (the packed pair of tiles = 8 consecutive features of the lane's token: 16-byte stores), V tile to LDS
    Frag<bf16_t> qf[4], kf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      bf16_t* qrow = a.qkv + (win * 64 + 16 * m + li) * (3L * C) + 32 * h + 8 * g;
#pragma unroll
      for (int i = 0; i < 6; ++i) { acc[i][m][0] += bq[i].x; acc[i][m][1] += bq[i].y; acc[i][m][2] += bq[i].z; acc[i][m][3] += bq[i].w; }
      qf[m] = pack_tr(acc[0][m], acc[1][m]);
      kf[m] = pack_tr(acc[2][m], acc[3][m]);
      const Frag<bf16_t> vfm = pack_tr(acc[4][m], acc[5][m]);
      *reinterpret_cast<bf16x8*>(qrow) = qf[m].v;
      *reinterpret_cast<bf16x8*>(qrow + C) = kf[m].v;
      *reinterpret_cast<bf16x8*>(qrow + 2 * C) = vfm.v;
#pragma unroll
      for (int t = 0; t < 2; ++t) lds_write8(sV_a + (16 * m + li) * VRS + (16 * t + 4 * g) * 2, pack4(acc[4 + t][m]));   // feature 8g+4t+r at column 16t+4g+r
    }
    lgk0();
    if (j == 1) stamp<DBG>(a.ts, 4);
    // S^T = K Q^T: key j = 16 jt + 4 g + r, query i = 16 it + li
    f32x4 s[4][4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int it = 0; it < 4; ++it) { s[jt][it] = f32x4{0.f, 0.f, 0.f, 0.f}; mma(s[jt][it], kf[jt], qf[it]); }
    // bias bin of (query i = 16 it + li, key j = 16 jt + 4 g + r): binA + (it - jt + 3) * 49 - r -- one LDS read with an immediate offset each
    const float* tb = sTab + h * 344 + ((li >> 2) - g + 3) * 7 + (li & 3);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = 16 * it + li;
      const unsigned ri = shifted ? sRb[i] : 0u;
      float mx = -3.0e38f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const unsigned rw = shifted ? *reinterpret_cast<const unsigned*>(sRb + 16 * jt + 4 * g) : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[jt][it][r] * scale + tb[(it - jt + 3) * 49 + 3 - r];
          if (shifted && ((rw >> (8 * r)) & 255u) != ri) v += -100.0f;
          s[jt][it][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(s[jt][it][r] - mx); s[jt][it][r] = e; sum += e; }
      sum = quad_row_sum(sum);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[jt][it][r] *= inv;
      if (g == 0) a.lse[(win * HEADS + h) * 64 + i] = mx + __logf(sum);
    }
    if (j == 1) stamp<DBG>(a.ts, 5);
    // O^T = V^T P^T: feature 8 g + 4 dt + r of query li
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      f32x4 oT[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        const Frag<bf16_t> pa = pack_tr(s[2 * ks2][it], s[2 * ks2 + 1][it]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const Frag<bf16_t> vfr = lds_frag_t(sV, VRS, ks2 * 32, dt * 16, lane, (bf16_t*)nullptr);
          mma(oT[dt], vfr, pa);
        }
      }
      ofj[it] = pack_tr(oT[0], oT[1]);
      if (!a.tok_saves) *reinterpret_cast<bf16x8*>(a.o + (win * 64 + 16 * it + li) * C + 32 * h + 8 * g) = ofj[it].v;
      else if (tok[it] >= 0) *reinterpret_cast<bf16x8*>(a.o + tok[it] * C + 32 * h + 8 * g) = ofj[it].v;
    }
    // (no dynamic register indexing: `of[j]` with a run-time j would put the array into scratch memory)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (j == 0) of[0][it] = ofj[it];
      else if (j == 1) of[1][it] = ofj[it];
      else of[2][it] = ofj[it];
    }
  }

  // ---- O of all heads to every wave (the exchange area aliases the attention scratch: everyone is done with it first)
  stamp<DBG>(a.ts, 6);
  wg_barrier();
  const unsigned ex_a = lds_addr(scratch);
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int m = 0; m < 4; ++m) lds_write16(ex_a + ((3 * wave + j) * 4 + m) * 1024 + lane * 16, of[j][m].v);
  wg_barrier();
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int k = 0; k < KS; ++k) xn[m][k].v = *reinterpret_cast<const bf16x8*>(scratch + (k * 4 + m) * 1024 + lane * 16);

  stamp<DBG>(a.ts, 7);
  // ---- proj: this wave's 96 output channels
  f32x4 pacc[6][4];
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) pacc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KS; ++k)
    ws.step(wf, lane, [&](const Frag<bf16_t> (&w)[G]) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < 6; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) mmad<DBG>(pacc[n][m], w[n], xn[m][k]);
    });
  stamp<DBG>(a.ts, 8);
  {
    uint4 xres[4][3];
    float4 bpv[6];
    float scv[4];
#pragma unroll
    for (int n = 0; n < 6; ++n) bpv[n] = *reinterpret_cast<const float4*>(a.bproj + 96 * wave + 32 * (n >> 1) + 8 * g + 4 * (n & 1));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long t = tok[m] < 0 ? 0 : tok[m];
      scv[m] = a.rowscale ? a.rowscale[t / a.rows_per_scale] : 1.0f;
#pragma unroll
      for (int p = 0; p < 3; ++p) xres[m][p] = *reinterpret_cast<const uint4*>(a.x + t * C + 96 * wave + 32 * p + 8 * g);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (tok[m] < 0) continue;
      bf16_t* const orow = a.x1 + tok[m] * C + 96 * wave + 8 * g;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        float xr[8];
        unpack8(xres[m][p], xr);
        f32x4 v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float b0 = r == 0 ? bpv[2 * p].x : r == 1 ? bpv[2 * p].y : r == 2 ? bpv[2 * p].z : bpv[2 * p].w;
          const float b1 = r == 0 ? bpv[2 * p + 1].x : r == 1 ? bpv[2 * p + 1].y : r == 2 ? bpv[2 * p + 1].z : bpv[2 * p + 1].w;
          v0[r] = (pacc[2 * p][m][r] + b0) * scv[m] + xr[r];
          v1[r] = (pacc[2 * p + 1][m][r] + b1) * scv[m] + xr[4 + r];
        }
        *reinterpret_cast<bf16x8*>(orow + 32 * p) = pack_tr(v0, v1).v;
      }
    }
  }
  stamp<DBG>(a.ts, 9);
  ws.drain();
  stamp<DBG>(a.ts, 10);
}


}  // namespace sw

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
namespace {
// The "attribute set" flag is keyed on (device, kernel ADDRESS): K is only the function-pointer type, which every instantiation of a kernel family
// shares, so a flag per K would skip the opt-in of all instantiations but the first one launched.
template <class K> int set_lds(K kernel, int bytes) {   // per device: a second GPU driven from the same process needs its own attribute
  constexpr int SLOTS = 64;
  static const void* seen[64][SLOTS] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const void* key = (const void*)kernel;
  int free_slot = -1;
  for (int i = 0; i < SLOTS; ++i) {
    if (seen[dev][i] == key) return 0;
    if (!seen[dev][i]) { free_slot = i; break; }
  }
  hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  if (free_slot >= 0) seen[dev][free_slot] = key;   // (table full: the idempotent call is simply repeated)
  return 0;
}
constexpr int RING_A = 4;
long long* ts_buf() { static long long* p = nullptr; if (!p) { hipMalloc(&p, 64 * sizeof(long long)); hipMemset(p, 0, 64 * sizeof(long long)); } return p; }
void ts_print(const char* what, int n, hipStream_t st) {
  long long h[64];
  hipStreamSynchronize(st);
  hipMemcpy(h, ts_buf(), sizeof(h), hipMemcpyDeviceToHost);
  fprintf(stderr, "[swin ts] %s:", what);
  for (int i = 1; i < n; ++i) fprintf(stderr, " %d-%d: %lld", i - 1, i, h[i] - h[i - 1]);
  fprintf(stderr, "  total %lld\n", h[n - 1] - h[0]);
}
}  // namespace

int k_swin_supported(int C) { return C == 96 || C == 192 || C == 384; }
long k_swin_stream_numel(int type, int C) {   // bf16 elements of a weight stream
  if (!k_swin_supported(C)) return -1;
  const int NW = C / 96;
  return (long)NW * sw::stream_steps(type, NW) * sw::STEP_BYTES / 2;
}

int k_swin_pack(const SwinPackItem* items, int n, hipStream_t st) {
  for (int base = 0; base < n; base += sw::MAX_PACK) {
    const int cnt = n - base < sw::MAX_PACK ? n - base : sw::MAX_PACK;
    sw::PackArgs pa;
    long maxthr = 0;
    for (int i = 0; i < cnt; ++i) {
      const SwinPackItem& it = items[base + i];
      if (!k_swin_supported(it.C) || it.type < 0 || it.type > 1 || !it.w0 || !it.dst) return -4;
      const int NW = it.C / 96;
      pa.d[i] = sw::PackDesc{it.w0, it.w1, (bf16_t*)it.dst, it.type, NW};
      const long thr = (long)NW * sw::stream_steps(it.type, NW) * sw::G * 64;
      if (thr > maxthr) maxthr = thr;
    }
    long gx = (maxthr + 255) / 256;
    if (gx > 1152) gx = 1152;
    hipLaunchKernelGGL(sw::swin_pack_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, st, pa);
    NMH_CHECK_LAUNCH();
  }
  return 0;
}

constexpr long SPLIT_MAX_TILES = 128, SPLIT_CNT_BYTES = 1024;
// bytes of the zero-initialised workspace the split MLP forward needs for M rows of width C (0: this shape is not split): arrival counters, fp32 partials
long k_swin_mlp_split_ws_bytes(long M, int C) {
  const long tiles = (M + 63) / 64;
  if (C != 384 || tiles > SPLIT_MAX_TILES || M <= 0) return 0;
  return SPLIT_CNT_BYTES + tiles * sw::mlp_segments(4) * 64 * C * 4;
}
template <int NW> static int launch_mlp_fwd(const sw::MlpFwdArgs& a, hipStream_t st) {
  constexpr int lds = sw::mlp_lds<NW, RING_A>();
  const int dbg = getenv("NMH_SWIN_DBG") ? atoi(getenv("NMH_SWIN_DBG")) : 0;   // timing-only variants (wrong results): 1 no weight DMA, 2 no MFMAs, 4 no GELU / softmax
  if (NW == 4 && dbg) {
#define DBG_CASE(D) case D: if (int e = set_lds(sw::swin_mlp_fwd_kernel<4, RING_A, D>, lds)) return e; \
    hipLaunchKernelGGL((sw::swin_mlp_fwd_kernel<4, RING_A, D>), dim3((unsigned)((a.M + 63) / 64)), dim3(256), lds, st, a); break;
    if (dbg == 8) {
      sw::MlpFwdArgs b = a; b.ts = ts_buf();
      if (int e = set_lds(sw::swin_mlp_fwd_kernel<4, RING_A, 8>, lds)) return e;
      hipLaunchKernelGGL((sw::swin_mlp_fwd_kernel<4, RING_A, 8>), dim3((unsigned)((a.M + 63) / 64)), dim3(256), lds, st, b);
      ts_print("mlp_fwd  start|ln|fc1(0)|.. round2: fc1+act|barrier|fc2 ..|tail|epilogue|drain", 10, st);
      return 0;
    }
    switch (dbg) { DBG_CASE(1) DBG_CASE(2) DBG_CASE(3) DBG_CASE(4) DBG_CASE(6) DBG_CASE(7) default: return -4; }
#undef DBG_CASE
    NMH_CHECK_LAUNCH();
    return 0;
  }
  if constexpr (NW == 4) {
    if (a.part) {   // (k_swin_mlp_fwd: a workspace was passed and the row tiles alone would leave half of the CUs idle)
      if (int e = set_lds(sw::swin_mlp_fwd_kernel<NW, RING_A, 0, true>, lds)) return e;
      const long tiles = (a.M + 63) / 64;
      hipLaunchKernelGGL((sw::swin_mlp_fwd_kernel<NW, RING_A, 0, true>), dim3((unsigned)((tiles + 7) / 8 * 16)), dim3(64 * NW), lds, st, a);
      NMH_CHECK_LAUNCH();
      return 0;
    }
  }
  if (int e = set_lds(sw::swin_mlp_fwd_kernel<NW, RING_A>, lds)) return e;
  hipLaunchKernelGGL((sw::swin_mlp_fwd_kernel<NW, RING_A>), dim3((unsigned)((a.M + 63) / 64)), dim3(64 * NW), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_swin_mlp_fwd(const void* x1, const float* gamma, const float* beta, const void* wstream, const float* b1, const float* b2, const float* rowscale, int rows_per_scale,
                   void* x2, void* x1n, void* hp, void* hact, float* mean, float* rstd, long M, int C, float eps, void* split_ws, long split_ws_bytes, hipStream_t st) {
  sw::MlpFwdArgs a{(const bf16_t*)x1, gamma, beta, (const char*)wstream, b1, b2, rowscale, rows_per_scale > 0 ? rows_per_scale : 1,
                   (bf16_t*)x2, (bf16_t*)x1n, (bf16_t*)hp, mean, rstd, M, eps, nullptr, (bf16_t*)hact, nullptr, nullptr};
  // two workgroups per row tile when the tiles alone cannot fill the chip (<= 128 tiles) and the caller lent the workspace (k_swin_mlp_split_ws_bytes)
  if (split_ws && C == 384 && k_swin_mlp_split_ws_bytes(M, C) > 0 && split_ws_bytes >= k_swin_mlp_split_ws_bytes(M, C)) {
    a.cnt = (int*)split_ws;
    a.part = (float*)((char*)split_ws + SPLIT_CNT_BYTES);
  }
  switch (C) {
    case 96: return launch_mlp_fwd<1>(a, st);
    case 192: return launch_mlp_fwd<2>(a, st);
    case 384: return launch_mlp_fwd<4>(a, st);
  }
  return -1;
}

template <int NW> static int launch_attn_fwd(const sw::AttnFwdArgs& a, long nwin, hipStream_t st) {
  constexpr int lds = sw::attn_fwd_lds<NW, RING_A>();
  const int dbg_all = getenv("NMH_SWIN_DBG") ? atoi(getenv("NMH_SWIN_DBG")) : 0;
  const int dbg = dbg_all & 3;
  if (NW == 4 && dbg_all == 8) {
    sw::AttnFwdArgs b = a; b.ts = ts_buf();
    if (int e = set_lds(sw::swin_attn_fwd_kernel<4, RING_A, 8>, lds)) return e;
    hipLaunchKernelGGL((sw::swin_attn_fwd_kernel<4, RING_A, 8>), dim3((unsigned)nwin), dim3(256), lds, st, b);
    ts_print("attn_fwd  start|ln+table|heads0..: qkv(1)|pack+stores(1)|S+softmax(1)|PV(1)+head2|exchange|proj|epilogue|drain", 11, st);
    return 0;
  }
  if (NW == 4 && dbg) {
#define DBG_CASE(D) case D: if (int e = set_lds(sw::swin_attn_fwd_kernel<4, RING_A, D>, lds)) return e; \
    hipLaunchKernelGGL((sw::swin_attn_fwd_kernel<4, RING_A, D>), dim3((unsigned)nwin), dim3(256), lds, st, a); break;
    switch (dbg) { DBG_CASE(1) DBG_CASE(2) DBG_CASE(3) default: return -4; }
#undef DBG_CASE
    NMH_CHECK_LAUNCH();
    return 0;
  }
  if (int e = set_lds(sw::swin_attn_fwd_kernel<NW, RING_A>, lds)) return e;
  hipLaunchKernelGGL((sw::swin_attn_fwd_kernel<NW, RING_A>), dim3((unsigned)nwin), dim3(64 * NW), lds, st, a);
  NMH_CHECK_LAUNCH();
  return 0;
}
int k_swin_attn_fwd(const void* x, const float* gamma, const float* beta, const void* wstream, const float* bqkv, const float* table, const float* bproj,
                    const float* rowscale, int rows_per_scale, void* xnw, float* mean, float* rstd, void* qkv, void* o, float* lse, void* x1,
                    const WinMap& wm, int C, float eps, int token_saves, hipStream_t st) {
  sw::AttnFwdArgs a{(const bf16_t*)x, gamma, beta, (const char*)wstream, bqkv, table, bproj, rowscale, rows_per_scale > 0 ? rows_per_scale : 1,
                    (bf16_t*)xnw, mean, rstd, (bf16_t*)qkv, (bf16_t*)o, lse, (bf16_t*)x1, wm, eps, nullptr, token_saves};
  const long nwin = (long)wm.B * (wm.PH / 4) * (wm.PW / 4) * (wm.PD / 4);
  if (nwin <= 0) return 0;
  if (nwin * 64 >= (1L << 31)) return -2;
  switch (C) {
    case 96: return launch_attn_fwd<1>(a, nwin, st);
    case 192: return launch_attn_fwd<2>(a, nwin, st);
    case 384: return launch_attn_fwd<4>(a, nwin, st);
  }
  return -1;
}
