// Probe: pins gfx950 MFMA fragment layouts and ds_read_b64_tr_b16 semantics
// before any production kernel relies on them. Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); u += 0x7fff + ((u>>16)&1); return (uint16_t)(u>>16);}
static inline float bf2f(uint16_t h){ uint32_t u=((uint32_t)h)<<16; float f; memcpy(&f,&u,4); return f;}

// C[16x16] = A[16x32] * B[32x16], A row-major [16][32], Bt row-major [16 n][32 k]
__global__ void k_mfma_bf16(const uint16_t* A, const uint16_t* Bt, float* C){
  int l = threadIdx.x; int i = l & 15, g = l >> 4;
  bf16x8_t a, b;
  for (int j=0;j<8;++j){ a[j] = (short)A[i*32 + g*8 + j]; b[j] = (short)Bt[i*32 + g*8 + j]; }
  f32x4_t acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0,0,0);
  for (int r=0;r<4;++r) C[(g*4+r)*16 + i] = acc[r];
}
// f32: C[16x16] = A[16x4]*B[4x16]; A [16][4], Bt [16][4]
__global__ void k_mfma_f32(const float* A, const float* Bt, float* C){
  int l = threadIdx.x; int i = l & 15, g = l >> 4;
  f32x4_t acc = {0,0,0,0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i*4+g], Bt[i*4+g], acc, 0,0,0);
  for (int r=0;r<4;++r) C[(g*4+r)*16 + i] = acc[r];
}
// tr-read probe: LDS holds ushort values = index; every lane supplies its own address.
__global__ void k_trread(const int* addr_elems, uint16_t* out){
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int t=threadIdx.x;t<4096;t+=64) lds[t]=(uint16_t)t;
  __syncthreads();
  int l = threadIdx.x;
  unsigned a = (unsigned)(size_t)(&lds[0]) ; // LDS address (low 32 bits of generic? use builtin below)
  (void)a;
  unsigned ldsaddr = (unsigned)(__builtin_amdgcn_readfirstlane(0)) + (unsigned)(addr_elems[l]*2);
  // address of lds[0] in LDS space:
  unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  ldsaddr += base;
  bf16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ldsaddr) : "memory");
  for (int j=0;j<4;++j) out[l*4+j] = (uint16_t)v[j];
}

int main(){
  int fails=0;
  { // bf16 mfma
    std::vector<uint16_t> A(16*32), Bt(16*32); std::vector<float> C(256), R(256);
    for(int i=0;i<16;++i)for(int k=0;k<32;++k){ A[i*32+k]=f2bf((float)((i*7+k*3)%11-5)); Bt[i*32+k]=f2bf((float)((i*5+k*2+i*k)%13-6)); }
    for(int i=0;i<16;++i)for(int n=0;n<16;++n){ float s=0; for(int k=0;k<32;++k) s+=bf2f(A[i*32+k])*bf2f(Bt[n*32+k]); R[i*16+n]=s; }
    uint16_t *dA,*dB; float* dC; hipMalloc(&dA,1024); hipMalloc(&dB,1024); hipMalloc(&dC,1024);
    hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice); hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice);
    k_mfma_bf16<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,1024,hipMemcpyDeviceToHost);
    double e=0; for(int t=0;t<256;++t) e=fmax(e,fabs(C[t]-R[t]));
    printf("mfma_f32_16x16x32_bf16 layout (A[i=l&15][k=8g+j], B[k=8g+j][n=l&15], C row=4g+r col=l&15): maxerr=%g %s\n", e, e<1e-3?"PASS":"FAIL"); fails += !(e<1e-3);
  }
  { // f32 mfma
    std::vector<float> A(64), Bt(64), C(256), R(256);
    for(int i=0;i<16;++i)for(int k=0;k<4;++k){ A[i*4+k]=(float)((i*7+k*3)%11-5)+0.25f; Bt[i*4+k]=(float)((i*5+k*2+i*k)%13-6)+0.5f; }
    for(int i=0;i<16;++i)for(int n=0;n<16;++n){ float s=0; for(int k=0;k<4;++k) s+=A[i*4+k]*Bt[n*4+k]; R[i*16+n]=s; }
    float *dA,*dB,*dC; hipMalloc(&dA,256); hipMalloc(&dB,256); hipMalloc(&dC,1024);
    hipMemcpy(dA,A.data(),256,hipMemcpyHostToDevice); hipMemcpy(dB,Bt.data(),256,hipMemcpyHostToDevice);
    k_mfma_f32<<<1,64>>>(dA,dB,dC); hipMemcpy(C.data(),dC,1024,hipMemcpyDeviceToHost);
    double e=0; for(int t=0;t<256;++t) e=fmax(e,fabs(C[t]-R[t]));
    printf("mfma_f32_16x16x4f32 layout: maxerr=%g %s\n", e, e<1e-4?"PASS":"FAIL"); fails += !(e<1e-4);
  }
  { // tr read: lane l in 16-group p=l&15, group q=l>>4: address = row (p>>2) of a [4][16] block with row stride S elems, cols (p&3)*4; block q at q*4 rows
    const int S=48; std::vector<int> ad(64); std::vector<uint16_t> out(256);
    for(int l=0;l<64;++l){ int p=l&15,q=l>>4; ad[l] = (q*4 + (p>>2))*S + (p&3)*4; }
    int* dad; uint16_t* dout; hipMalloc(&dad,256); hipMalloc(&dout,512);
    hipMemcpy(dad,ad.data(),256,hipMemcpyHostToDevice);
    k_trread<<<1,64>>>(dad,dout); hipMemcpy(out.data(),dout,512,hipMemcpyDeviceToHost);
    // hypothesis: lane (q,c=p) elem j == element at row q*4+j, col c  => value (q*4+j)*S + c
    int bad=0; for(int l=0;l<64;++l){int c=l&15,q=l>>4; for(int j=0;j<4;++j){ int exp=(q*4+j)*S+c; if(out[l*4+j]!=exp) bad++; }}
    printf("ds_read_b64_tr_b16 hypothesis (lane c gets rows q*4+0..3 at col c; row fed by lanes 4k..4k+3): mismatches=%d %s\n", bad, bad==0?"PASS":"FAIL"); fails += bad!=0;
    if(bad){ for(int l=0;l<64;++l){ printf("lane %2d:",l); for(int j=0;j<4;++j) printf(" %5d(r%d,c%d)", out[l*4+j], out[l*4+j]/S, out[l*4+j]%S); printf("\n"); } }
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p,0);
  printf("device %s CUs=%d clock=%d kHz lds/blk=%zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  printf("PROBE %s\n", fails?"FAILED":"OK");
  return fails;
}
