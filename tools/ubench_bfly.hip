// Butterfly-core microbenchmark: compares modmul formulations (design input). hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32; typedef unsigned __int128 u128;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
#define ITER 1024

__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
struct Tw { u32 w0, w1, p0, p1; };
struct QC { u32 nq0, nq1; u32 twoq_hi; u64 twoq, m2q, threeq; };

// T = y*w mod q in [0, 1.5q) for y < 2^63-2^33; tw.p = floor(w*2^63/q)
__device__ __forceinline__ u64 shoup63(u64 y, const Tw& t, u32 nq0, u32 nq1) {
  const u32 y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 A = mad32(y0, t.p0, 0) >> 32;
  u64 M = mad32(y1, t.p0, A);
  M = mad32(y0, t.p1, M);
  u64 Q = mad32(y1 << 1, t.p1, M >> 31);
  const u32 q0 = (u32)Q, q1 = (u32)(Q >> 32);
  u64 R = mad32(y0, t.w0, 0);
  R = mad32(q0, nq0, R);
  u32 hi = (u32)(R >> 32) + y0 * t.w1 + y1 * t.w0 + q0 * nq1 + q1 * nq0;
  return (R & 0xffffffffull) | ((u64)hi << 32);
}
__device__ __forceinline__ void bfly63(u64& X, u64& Y, const Tw& t, const QC& c) {
  const bool big = (u32)(X >> 32) > c.twoq_hi;
  const u64 xc = X + (big ? c.m2q : 0ull);
  const u64 xc2 = X + (big ? 0ull : c.twoq);
  const u64 T = shoup63(Y, t, c.nq0, c.nq1);
  X = xc + T;
  Y = xc2 - T;
}
// reference formulation (current kernel)
__device__ __forceinline__ void bfly_ref(u64& X, u64& Y, u64 w, u64 wp, u64 q, u64 two_q) {
  u64 x = X >= two_q ? X - two_q : X;
  u64 Q = __umul64hi(Y, wp); u64 t = Y * w - Q * q;
  X = x + t; Y = x - t + two_q;
}

template<int VAR> __global__ void __launch_bounds__(256) k(u64* out, u64 q, u64 w, u64 wp, Tw t, QC c, u64 seed) {
  u64 x[8], y[8];
  for (int i=0;i<8;i++){ x[i]=(threadIdx.x*8+i+seed)&((1ull<<59)-1); y[i]=(x[i]*7+3)&((1ull<<59)-1); }
  for (int it=0; it<ITER; it++) {
#pragma unroll
    for (int i=0;i<8;i++) {
      if (VAR==0) bfly_ref(x[i], y[i], w, wp, q, 2*q);
      else bfly63(x[i], y[i], t, c);
    }
    t.w0 += 2; w += 2;
  }
  u64 r=0; for (int i=0;i<8;i++) r^=x[i]^y[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}
// correctness: both variants must agree mod q
__global__ void check(u64* bad, u64 q, u64 w) {
  const u64 wp = (u64)(((u128)w << 64) / q), wpp = (u64)(((u128)w << 63) / q);
  Tw t = {(u32)w, (u32)(w>>32), (u32)wpp, (u32)(wpp>>32)};
  QC c; c.nq0=(u32)(0-q); c.nq1=(u32)((0-q)>>32); c.twoq=2*q; c.m2q=0-2*q; c.twoq_hi=(u32)((2*q)>>32);
  u64 X = (threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x * 0xBF58476D1CE4E5B9ull) % (4*q);
  u64 Y = (X * 0x94D049BB133111EBull + 12345) % (4*q);
  u64 a=X,b=Y,cx=X,cy=Y;
  for (int s=0;s<20;s++) { bfly_ref(a,b,w,wp,q,2*q); bfly63(cx,cy,t,c); if (a%q!=cx%q || b%q!=cy%q || cx>=4*q+(1ull<<32) || cy>=4*q+(1ull<<32)) atomicAdd((unsigned long long*)bad,1ull); u64 tmp=a; a=b; b=tmp; tmp=cx; cx=cy; cy=tmp; }
}
int main() {
  CK(hipSetDevice(0)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  int blocks = p.multiProcessorCount*8; u64* d; CK(hipMalloc(&d, (size_t)blocks*256*8 + 64));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  u64 q = 1152921504606830593ull; // 2^60 - 2^14*k + 1 shape
  u64 w = 0x0123456789abcdefull % q;
  CK(hipMemset(d,0,64)); check<<<1024,256>>>(d,q,w); CK(hipDeviceSynchronize()); u64 bad; CK(hipMemcpy(&bad,d,8,hipMemcpyDeviceToHost)); printf("mismatches: %llu\n",(unsigned long long)bad);
  const char* nm[]={"ref_harvey_shoup64","shoup63_nq_hicmp"};
  const u64 wp = (u64)(((u128)w << 64) / q), wpp = (u64)(((u128)w << 63) / q);
  Tw t = {(u32)w, (u32)(w>>32), (u32)wpp, (u32)(wpp>>32)};
  QC c; c.nq0=(u32)(0-q); c.nq1=(u32)((0-q)>>32); c.twoq=2*q; c.m2q=0-2*q; c.twoq_hi=(u32)((2*q)>>32); c.threeq=3*q;
#define RUN(V) { k<V><<<blocks,256>>>(d,q,w,wp,t,c,1); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for(int r=0;r<5;r++) k<V><<<blocks,256>>>(d,q,w,wp,t,c,r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); double ops=5.0*blocks*256.0*ITER*8; printf("%-22s %8.2f Gbutterfly/s -> n=2^14 rows/s = %.2f M\n", nm[V], ops/ms/1e6, ops/ms/1e6*1e9/114688/1e6); }
  RUN(0) RUN(1)
  return 0;
}
