// micro-benchmark: sustained FP64 MFMA (v_mfma_f64_16x16x4_f64) rate on this GPU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template<int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-3, b=b0-threadIdx.x*1e-3;
  long long c0=clock64(), w0=wall_clock64();
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  long long c1=clock64(), w1=wall_clock64();
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*256+threadIdx.x]=s;
  if(threadIdx.x==0 && blockIdx.x==0){ ((long long*)out)[1<<20]=c1-c0; ((long long*)out)[(1<<20)+1]=w1-w0; }
}
int main(){
  double* out; hipMalloc(&out,(1<<23)+64); hipStream_t st; hipStreamCreate(&st); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int wgs : {256, 512, 1024}) for(int rep=0;rep<2;rep++){
    int iters=20000;
    hipEventRecord(e0,st); hipLaunchKernelGGL(k<16>,dim3(wgs),dim3(256),0,st,out,iters,1.0001,0.9999); hipEventRecord(e1,st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms,e0,e1); long long h[2]; hipMemcpy(h,((long long*)out)+(1<<20),16,hipMemcpyDeviceToHost);
    double flops=(double)wgs*4*iters*16*2048.0;
    printf("NACC=16 wgs=%d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles/MFMA(per wave)=%.1f\n", wgs, ms, flops/ms/1e9, h[0]/(h[1]/100.0), (double)h[0]/(iters*16.0));
  }
  { int iters=20000, wgs=256;
    hipEventRecord(e0,st); hipLaunchKernelGGL(k<4>,dim3(wgs),dim3(256),0,st,out,iters,1.0001,0.9999); hipEventRecord(e1,st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms,e0,e1); long long h[2]; hipMemcpy(h,((long long*)out)+(1<<20),16,hipMemcpyDeviceToHost);
    printf("NACC=4 wgs=%d: %.3f ms  %.1f TFLOP/s clock %.0f MHz cycles/MFMA=%.1f\n", wgs, ms, (double)wgs*4*iters*4*2048.0/ms/1e9, h[0]/(h[1]/100.0), (double)h[0]/(iters*4.0)); }
  return 0;
}
