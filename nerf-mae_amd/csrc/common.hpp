// Shared device primitives for the gfx950 (CDNA4, wave64) kernels of the NeRF-MAE hot path.
// Everything here is written for gfx950 only: 16x16 MFMA fragments (layouts pinned on hardware by
// tools/probe/probe_layouts.hip), 128-byte swizzled LDS rows read with ds_read_b128, and
// ds_read_b64_tr_b16 transpose reads for the weight-gradient (contraction-major) operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define NMH_DT_F32 0
#define NMH_DT_BF16 1

// fp32 -> bf16, round-to-nearest-even: gfx950 has v_cvt_pk_bf16_f32 (two values per instruction; the integer emulation is 4-5 VALU
// instructions per value, which showed up in every epilogue)
typedef float nmh_f2v __attribute__((ext_vector_type(2)));
typedef __bf16 nmh_b2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {   // {lo, hi} -> one dword, lo in bits 0-15
  nmh_f2v v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, nmh_b2v));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); pointers must be 16-B aligned.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  // streaming variants (non-temporal: the lines are not kept in L2 / MALL -- for one-pass tensors far larger than the caches)
  static __device__ __forceinline__ void load_nt(const float* p, float (&v)[8]) {
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  static __device__ __forceinline__ void store_nt(float* p, const float (&v)[8]) {
    __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
    __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(p + 4));
  }
};
template <> struct Vec8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ void load_nt(const bf16_t* p, float (&v)[8]) {
    const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void store_nt(bf16_t* p, const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
  }
};

// ---- MFMA fragments: one "k-step" = 32 contraction elements for both dtypes -------------------
// slot (g = lane>>4, j = 0..7) of lane (i = lane&15) holds A[i][slot] / B[slot][i]; C: col = lane&15,
// row = 4*(lane>>4)+r.  bf16: one v_mfma_f32_16x16x32_bf16; f32: eight exact v_mfma_f32_16x16x4_f32.
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma(f32x4& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// ---- LDS tiles with 128-byte rows, 16-byte chunks XOR-swizzled by (row>>1)&7 ---------------------
// bf16 row = 64 elements (two k-steps), f32 row = 32 elements (one k-step).
template <typename T> struct Row128 { static constexpr int KT = 128 / sizeof(T); static constexpr int KSTEPS = KT / 32; };

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// read the k-step-s fragment of `row` (this lane's group g) from a swizzled tile at byte base `lds`
__device__ __forceinline__ Frag<bf16_t> lds_frag(const char* lds, int row, int s, int g, bf16_t*) {
  Frag<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(lds + swz_off(row, s * 4 + g));
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag(const char* lds, int row, int /*s*/, int g, float*) {
  Frag<float> f;
  float4 a = *reinterpret_cast<const float4*>(lds + swz_off(row, 2 * g));
  float4 b = *reinterpret_cast<const float4*>(lds + swz_off(row, 2 * g + 1));
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}

// ---- contraction-major ("transposed") fragments for the TN / wgrad kernels ------------------------
// The LDS tile is stored as loaded from memory: [m (contraction)][cols], row stride RS bytes.
// k-step covers 32 contraction rows m0..m0+31; slot (g,j): m = m0 + 4g + j (j<4), m0 + 16 + 4g + (j-4).
// Both operands use the same slot->m map, so the MFMA sums matching m.  (probe: tr-read semantics)
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x4 ds_read_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(__attribute__((address_space(3))) char*)p);
}
__device__ __forceinline__ Frag<bf16_t> lds_frag_t(const char* tile, int RS, int m0, int col0, int lane, bf16_t*) {
  const int g = lane >> 4, p = lane & 15;
  const char* a = tile + (m0 + 4 * g + (p >> 2)) * RS + (col0 + (p & 3) * 4) * 2;
  bf16x4 lo = ds_read_tr16(a);
  bf16x4 hi = ds_read_tr16(a + 16 * RS);
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_t(const char* tile, int RS, int m0, int col0, int lane, float*) {
  const int g = lane >> 4, i = lane & 15;
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int m = m0 + 4 * g + (j & 3) + (j >> 2) * 16;
    f.v[j] = *reinterpret_cast<const float*>(tile + m * RS + (col0 + i) * 4);
  }
  return f;
}

// ---- raw transpose reads for LDS-DMA pipelines ----------------------------------------------------
// hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of a `__builtin_amdgcn_ds_read_tr16_b64` whenever an LDS-DMA (`global_load_lds`) may be in flight: its
// wait-count pass cannot tell the ring slot being read from the slot being filled.  In a ring with counted `vmcnt(N)` waits that turns every step into a
// synchronous load (round 5: every contraction-major LDS-DMA kernel of rounds 2-4 ran that way -- tools/scan_vmcnt0.py).  Inline-asm reads are invisible to
// that pass; the caller orders them: tr_read_raw ... (all reads of a step), tr_wait(), tr_pin(f) on every fragment (an empty volatile asm that redefines the
// registers behind the wait, so that no consumer can be scheduled above it), then tr_frag(f) as the MFMA operand.
struct TrFrag { bf16x4 lo, hi; };
__device__ __forceinline__ unsigned lds_addr_u(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
template <int HI_OFF, int OFF0 = 0> __device__ __forceinline__ void tr_read_raw(TrFrag& f, unsigned addr) {   // HI_OFF: byte distance of contraction rows m + 16 (16 x row stride); OFF0: immediate added to both
  static_assert(OFF0 >= 0 && OFF0 + HI_OFF < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(addr), "n"(OFF0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(OFF0 + HI_OFF));
}
// a 16-byte LDS read the compiler does not see (it orders every LDS read it does see behind ALL LDS-DMAs in flight: s_waitcnt vmcnt(0)); own lgkmcnt wait
__device__ __forceinline__ float4 lds_read_f4_raw(const float* p) {
  float4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr_u(p)) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ void tr_read1(bf16x4& d, unsigned addr) {   // one half of a fragment at its own address (rows of the two halves not a fixed distance apart)
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tr_pin(TrFrag& f) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }
__device__ __forceinline__ Frag<bf16_t> tr_frag(const TrFrag& f) {
  Frag<bf16_t> r;
  r.v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return r;
}

// ---- misc ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
// bf16 epilogues: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute, three orders below the bf16 rounding of the result) --
// one v_rcp_f32, one v_exp_f32 and six FMAs instead of the two-branch libm erff.  The Linear(C -> 4C) + GELU epilogue at 40^3 tokens is
// VALU-bound on erff (K = 96: three k-steps of MFMAs against 24 GELUs per lane); the fp32 parity mode keeps erff.
__device__ __forceinline__ void erf_as_parts(float z, float& erfz, float& e) {   // erf(z) and exp(-z^2)
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
  e = __expf(-az * az);
  float p = 1.061405429f;
  p = p * t - 1.453152027f; p = p * t + 1.421413741f; p = p * t - 0.284496736f; p = p * t + 0.254829592f;
  erfz = copysignf(1.0f - p * t * e, z);
}
__device__ __forceinline__ float gelu_fast_f(float x) {
  float er, e;
  erf_as_parts(x * 0.70710678118654752f, er, e);
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {   // Phi(x) + x phi(x); exp(-x^2/2) is the erf formula's own exponential
  float er, e;
  erf_as_parts(x * 0.70710678118654752f, er, e);
  return 0.5f * (1.0f + er) + x * 0.39894228040143268f * e;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: a launcher's "done once" flag is kept per device, so that a process driving several
// GPUs (or a second device selected later) opts every one of them in.  (Idempotent, so a benign race between two host threads only repeats the call.)
struct NmhPerDeviceOnce {
  bool done[64] = {};
  static int dev() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0; return d; }
  bool need() const { return !done[dev()]; }
  void set() { done[dev()] = true; }
};
// Small accumulators are cleared by a kernel instead of hipMemsetAsync: inside a captured graph a memset node costs 20-30 us on the
// dependent chain (rocprofv3: __amd_rocclr_fillBufferAligned, 2-3 workgroups), a kernel node ~5 us.  bytes must be a multiple of 4.
#ifdef __HIPCC__
static __global__ void nmh_zero_kernel(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
// nmh_set_prezeroed_arena (capi.hip): accumulators inside the caller's arena are zero on entry by contract (the caller clears the arena
// once per step with ONE launch), so the per-call clearing launch is skipped for them -- 21 of the 24 clearing launches of a training step
extern char* g_nmh_arena_base;
extern size_t g_nmh_arena_bytes;
static inline hipError_t nmh_zero_async(void* p, size_t bytes, hipStream_t st) {
  const long n = (long)(bytes / 4);
  if (n <= 0) return hipSuccess;
  if (g_nmh_arena_bytes && (char*)p >= g_nmh_arena_base && (char*)p + bytes <= g_nmh_arena_base + g_nmh_arena_bytes) return hipSuccess;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(nmh_zero_kernel, dim3((unsigned)nb), dim3(256), 0, st, (unsigned*)p, n);
  return hipGetLastError();
}
#endif
#define NMH_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
