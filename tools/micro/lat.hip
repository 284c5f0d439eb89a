// micro-benchmark: what does a latency-bound single-workgroup kernel cost on this GPU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void k_empty(long long* out){ if(threadIdx.x==0 && out==nullptr) out[0]=1; }

// potrf-like: 64 steps of {lds write, barrier, lds reads, fp64 division, 32 fma}
__global__ __launch_bounds__(256) void k_steps(double* g, long long* out, int steps, int do_div){
  __shared__ double buf[2][64];
  long long c0 = clock64(), w0 = wall_clock64();
  double e[16], m[16];
  for(int i=0;i<16;i++){ e[i]=g[threadIdx.x*16+i]; m[i]=0.5*e[i]; }
  int tx = threadIdx.x&15, ty=threadIdx.x>>4;
  for(int j=0;j<steps;j++){
    int b=j&1;
    if(ty==(j&15)) for(int a=0;a<4;a++) buf[b][tx+16*a]=e[a*4+(j>>4)%4];
    __syncthreads();
    double d = buf[b][j&63];
    double inv = do_div ? 1.0/d : d*0.5;
    double ck[4]; for(int q=0;q<4;q++) ck[q]=buf[b][ty+16*q];
    for(int a=0;a<4;a++){ double li = buf[b][tx+16*a]*inv; for(int q=0;q<4;q++){ e[a*4+q]-=li*ck[q]; m[a*4+q]-=li*ck[3-q]; } }
  }
  double s=0; for(int i=0;i<16;i++) s+=e[i]+m[i];
  g[threadIdx.x]=s;
  long long c1 = clock64(), w1 = wall_clock64();
  if(threadIdx.x==0){ out[0]=c1-c0; out[1]=w1-w0; }
}
// dependent global load chain
__global__ void k_chain(const int* idx, long long* out, int n){
  long long c0 = clock64(), w0 = wall_clock64();
  int p=0; for(int i=0;i<n;i++) p=idx[p];
  long long c1 = clock64(), w1 = wall_clock64();
  if(threadIdx.x==0){ out[0]=c1-c0; out[1]=w1-w0; out[2]=p; }
}
int main(){
  double* g; long long* out; int* idx;
  HC(hipMalloc(&g, 1<<20)); HC(hipMalloc(&out, 64)); HC(hipMalloc(&idx, 64<<20));
  std::vector<double> h(1<<17, 1.25); HC(hipMemcpy(g,h.data(),1<<20,hipMemcpyHostToDevice));
  std::vector<int> hi(16<<20); for(size_t i=0;i<hi.size();i++) hi[i]=(int)((i*1048583ull+12345)%hi.size());
  HC(hipMemcpy(idx,hi.data(),64<<20,hipMemcpyHostToDevice));
  hipStream_t st; HC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0,e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  long long ho[8]; float ms;
  for(int rep=0;rep<2;rep++){
    HC(hipEventRecord(e0,st)); for(int i=0;i<1000;i++) hipLaunchKernelGGL(k_empty,dim3(1),dim3(256),0,st,out); HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st));
    HC(hipEventElapsedTime(&ms,e0,e1)); printf("empty kernel x1000: %.2f us each (back-to-back on one stream)\n", ms);
    for(int dv=0;dv<2;dv++){
      HC(hipEventRecord(e0,st)); for(int i=0;i<200;i++) hipLaunchKernelGGL(k_steps,dim3(1),dim3(256),0,st,g,out,64,dv); HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st));
      HC(hipEventElapsedTime(&ms,e0,e1)); HC(hipMemcpy(ho,out,64,hipMemcpyDeviceToHost));
      printf("k_steps(64 steps, div=%d) 1 WG: %.2f us per launch; in-kernel cycles=%lld wall(100MHz ticks)=%lld => clock %.0f MHz, %.0f cycles/step\n", dv, ms*1000/200, ho[0], ho[1], ho[0]/(ho[1]/100.0), ho[0]/64.0);
    }
    HC(hipEventRecord(e0,st)); for(int i=0;i<200;i++) hipLaunchKernelGGL(k_steps,dim3(256),dim3(256),0,st,g,out,64,1); HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st));
    HC(hipEventElapsedTime(&ms,e0,e1)); HC(hipMemcpy(ho,out,64,hipMemcpyDeviceToHost));
    printf("k_steps 256 WGs: %.2f us per launch; cycles=%lld wall=%lld => clock %.0f MHz\n", ms*1000/200, ho[0], ho[1], ho[0]/(ho[1]/100.0));
    hipLaunchKernelGGL(k_chain,dim3(1),dim3(64),0,st,idx,out,1000); HC(hipStreamSynchronize(st)); HC(hipMemcpy(ho,out,64,hipMemcpyDeviceToHost));
    printf("dependent HBM load chain: %.0f cycles/load, %.1f ns/load\n", ho[0]/1000.0, ho[1]*10.0/1000.0);
  }
  return 0;
}
