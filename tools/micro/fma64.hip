#include <hip/hip_runtime.h>
#include <cstdio>
template<int N>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0){
  double acc[N];
  for(int i=0;i<N;i++) acc[i]=threadIdx.x*1e-9+i;
  double a=a0+threadIdx.x*1e-12, b=b0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<N;i++) acc[i]=__builtin_fma(acc[i],a,b);
  }
  double s=0; for(int i=0;i<N;i++) s+=acc[i];
  out[blockIdx.x*256+threadIdx.x]=s;
}
int main(){
  double* out; (void)hipMalloc(&out,1<<24); hipStream_t st; (void)hipStreamCreate(&st); hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for(int wgs : {1024, 2048}) for(int rep=0;rep<2;rep++){
    int iters=20000;
    (void)hipEventRecord(e0,st); hipLaunchKernelGGL(k<16>,dim3(wgs),dim3(256),0,st,out,iters,0.999999,1e-7); (void)hipEventRecord(e1,st); (void)hipStreamSynchronize(st);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1);
    printf("v_fma_f64 wgs=%d: %.3f ms  %.1f TFLOP/s\n", wgs, ms, (double)wgs*256*iters*16*2.0/ms/1e9);
  }
  return 0;
}
