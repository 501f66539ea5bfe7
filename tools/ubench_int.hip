// Integer-multiply issue-rate microbenchmark for gfx950 (design input for the modmul choice).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_int tools/ubench_int.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

typedef unsigned long long u64; typedef unsigned int u32;
#define ITER 4096

template<int OP> __global__ void __launch_bounds__(256) k_raw(u32* out, u32 seed) {
  u32 a0=threadIdx.x+seed, a1=a0*3+1, a2=a0*5+7, a3=a0*7+3, a4=a0*11+1, a5=a0*13+5, a6=a0*17+9, a7=a0*19+2;
  u32 c = seed|1;
  u64 b0=a0,b1=a1,b2=a2,b3=a3,b4=a4,b5=a5,b6=a6,b7=a7;
  double d0=a0,d1=a1,d2=a2,d3=a3,d4=a4,d5=a5,d6=a6,d7=a7; double dc = 1.0000001;
  for (int i=0;i<ITER;i++) {
    if (OP==0) { // v_mul_lo_u32
#define R(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==1) {
#define R(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==2) { // v_mad_u64_u32
#define R(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(c), "v"(a0) : "vcc");
      R(b0)R(b1)R(b2)R(b3)R(b4)R(b5)R(b6)R(b7)
#undef R
    } else if (OP==3) { // v_mul_u32_u24
#define R(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==4) { // v_add_u32
#define R(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==5) { // v_lshl_add_u64
#define R(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(b7));
      R(b0)R(b1)R(b2)R(b3)R(b4)R(b5)R(b6)R(b0)
#undef R
    } else if (OP==6) { // v_fma_f64
#define R(x) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(dc));
      R(d0)R(d1)R(d2)R(d3)R(d4)R(d5)R(d6)R(d7)
#undef R
    } else if (OP==7) { // v_add_co_u32 + v_addc_co_u32 pair
#define R(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(x), "+v"(c), "+v"(a7) : : "vcc");
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a0)
#undef R
    } else if (OP==8) { // v_mul_hi_u32_u24
#define R(x) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==9) { // v_mad_u32_u24
#define R(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==10) { // v_add3_u32
#define R(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==12) { // v_mul_f64
#define R(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(dc));
      R(d0)R(d1)R(d2)R(d3)R(d4)R(d5)R(d6)R(d7)
#undef R
    } else if (OP==13) { // v_dot4_u32_u8: four 8x8-bit products + accumulate per lane
#define R(x) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==14) { // v_mad_u64_u32 with the product's operands independent of the accumulator (dot-product shape)
#define R(x, y) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(c), "v"(y) : "vcc");
      R(b0,a1)R(b1,a2)R(b2,a3)R(b3,a4)R(b4,a5)R(b5,a6)R(b6,a7)R(b7,a0)
#undef R
    } else if (OP==15) { // v_pk_mul_lo_u16: two 16x16 -> low 16 products per lane
#define R(x) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==16) { // v_fma_f32
#define R(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==17) { // v_cvt_f64_u32 + v_cvt_u32_f64 pair
#define R(x, y) asm volatile("v_cvt_f64_u32 %1, %0\n v_cvt_u32_f64 %0, %1" : "+v"(x), "+v"(y));
      R(a0,d0)R(a1,d1)R(a2,d2)R(a3,d3)R(a4,d4)R(a5,d5)R(a6,d6)R(a7,d7)
#undef R
    } else if (OP==18) { // v_min_u32 (conditional subtract idiom: sub + min)
#define R(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(c));
      R(a0)R(a1)R(a2)R(a3)R(a4)R(a5)R(a6)R(a7)
#undef R
    } else if (OP==11) { // v_cndmask + v_cmp (cmp_u64)
#define R(x) asm volatile("v_cmp_ge_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(x), "+v"(b7), "+v"(a7), "+v"(c) : : "vcc");
      R(b0)R(b1)R(b2)R(b3)R(b4)R(b5)R(b6)R(b0)
#undef R
    }
  }
  u32 r = a0^a1^a2^a3^a4^a5^a6^a7 ^ (u32)(b0^b1^b2^b3^b4^b5^b6^b7) ^ (u32)(d0+d1+d2+d3+d4+d5+d6+d7);
  if (r==0x12345) out[threadIdx.x]=r;
}

// ---------------- butterfly-level benchmarks (C level, compiler-generated code) ----------------
__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a,b); }

// Shoup lazy modmul: returns y*w mod q in [0,2q)
__device__ __forceinline__ u64 shoup(u64 y, u64 w, u64 wp, u64 q) { u64 Q = mulhi64(y, wp); return y*w - Q*q; }

// pseudo-Mersenne q = 2^60 - c, c<2^32: y<2^64?, w<2^60 -> result in [0,2q)
__device__ __forceinline__ u64 pmers(u64 y, u64 w, u32 c) {
  unsigned __int128 x = (unsigned __int128)y * w;           // < 2^124
  u64 lo = (u64)x & ((1ull<<60)-1); u64 H = (u64)(x >> 60); // H < 2^64
  unsigned __int128 x2 = (unsigned __int128)H * c + lo;     // < 2^96
  u64 lo2 = (u64)x2 & ((1ull<<60)-1); u64 H2 = (u64)(x2 >> 60); // < 2^36
  return H2 * c + lo2;   // < 2^68?? H2<2^36,c<2^27 in practice -> < 2^63+2^60
}

template<int VAR> __global__ void __launch_bounds__(256) k_bfly(u64* out, u64 q, u64 w, u64 wp, u32 c, u64 seed) {
  u64 x[8], y[8];
  for (int i=0;i<8;i++){ x[i]=(threadIdx.x*8+i+seed)%q; y[i]=(x[i]*7+3)%q; }
  u64 twoq = 2*q;
  for (int it=0; it<ITER/4; it++) {
#pragma unroll
    for (int i=0;i<8;i++) {
      u64 X=x[i], Y=y[i];
      if (VAR==0) { // Harvey lazy butterfly w/ correction
        if (X>=twoq) X-=twoq;
        u64 T = shoup(Y,w,wp,q);
        x[i]=X+T; y[i]=X-T+twoq;
      } else if (VAR==1) { // no correction of X (lazy-lazy), correction amortised elsewhere
        u64 T = shoup(Y,w,wp,q);
        x[i]=X+T; y[i]=X-T+twoq;
      } else if (VAR==2) { // pseudo-Mersenne
        if (X>=twoq) X-=twoq;
        u64 T = pmers(Y,w,c);
        x[i]=X+T; y[i]=X-T+twoq;
      } else if (VAR==3) {
        u64 T = pmers(Y,w,c);
        x[i]=X+T; y[i]=X-T+twoq;
      }
      w += 2; // defeat hoisting a bit
    }
  }
  u64 r=0; for (int i=0;i<8;i++) r^=x[i]^y[i];
  if (r==0x1234567) out[threadIdx.x]=r;
}

int main() {
  int dev=0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,dev));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  u32* d; CK(hipMalloc(&d, 1<<20)); u64* d64=(u64*)d;
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int blocks = p.multiProcessorCount*8; // 8 blocks of 256 = 32 waves/CU
  const char* names[]={"v_mul_lo_u32","v_mul_hi_u32","v_mad_u64_u32","v_mul_u32_u24","v_add_u32","v_lshl_add_u64","v_fma_f64","add_co+addc(2 instr)","v_mul_hi_u32_u24","v_mad_u32_u24","v_add3_u32","cmp_ge_u64+cndmask(2 instr)","v_mul_f64","v_dot4_u32_u8","v_mad_u64_u32 (indep. operands)","v_pk_mul_lo_u16","v_fma_f32","cvt_f64_u32+cvt_u32_f64(2 instr)","v_min_u32"};
#define RUN(OP) { k_raw<OP><<<blocks,256>>>(d,1); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for(int r=0;r<5;r++) k_raw<OP><<<blocks,256>>>(d,r+2); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double ops=5.0*blocks*256.0*ITER*8; printf("%-28s %8.2f Glane-instr/s  (%.3f of 78.6T full rate)\n", names[OP], ops/ms/1e6, ops/ms/1e6/78643.2*1.0); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18)
  u64 q = (1ull<<60) - 33*65536ull + 1; // shape only
  const char* bn[]={"harvey_shoup","shoup_nocorr","pmers_corr","pmers_nocorr"};
#define RUNB(V) { k_bfly<V><<<blocks,256>>>(d64,q,12345,6789,(u32)(33*65536-1),1); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for(int r=0;r<5;r++) k_bfly<V><<<blocks,256>>>(d64,q,12345+r,6789,(u32)(33*65536-1),r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double ops=5.0*blocks*256.0*(ITER/4)*8; printf("%-16s %8.2f Gbutterfly/s -> n=2^14 rows/s = %.2f M (HBM 8TB/s = 30.5M)\n", bn[V], ops/ms/1e6, ops/ms/1e6*1e9/114688/1e6); }
  RUNB(0) RUNB(1) RUNB(2) RUNB(3)
  return 0;
}
