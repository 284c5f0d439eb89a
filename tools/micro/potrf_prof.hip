// micro-benchmark: the 64x64 diagonal-block Cholesky + inverse (k_potrf64) on one workgroup: time per launch,
// phase cycles (build with -DGMRFX_CYC) and a correctness check of L and X = L^-1.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DGMRFX_CYC] -I../../gaussianmarkovrandomfields.jl_amd/csrc potrf_prof.hip -o potrf_prof
#include "../../gaussianmarkovrandomfields.jl_amd/csrc/potrf64.hip"
#include <cstdio>
#include <chrono>
#include <vector>
#include <cmath>
#include <climits>
#define HC(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
using namespace gmrfx;
int main(){
  const int n=64;
  std::vector<double> A(n*n,0.0);
  for(int j=0;j<n;j++) for(int i=j;i<n;i++) A[i+j*n]= (i==j)? 9.0+0.01*i : -1.0/(1+abs(i-j));   // strictly diagonally dominant: SPD (4.0 was indefinite: NaNs that fmax() hid)
  double *dL,*dA; int *dlist,*dsf,*dld,*dinfo; long long *dpp;
  HC(hipMalloc(&dL,n*n*8)); HC(hipMalloc(&dA,n*n*8)); HC(hipMemcpy(dA,A.data(),n*n*8,hipMemcpyHostToDevice));
  int list0=0, sf[2]={0,n}, ldv=n, info=INT_MAX; long long pp[2]={0,(long long)n*n};
  HC(hipMalloc(&dlist,4)); HC(hipMalloc(&dsf,8)); HC(hipMalloc(&dld,4)); HC(hipMalloc(&dinfo,4)); HC(hipMalloc(&dpp,16));
  HC(hipMemcpy(dlist,&list0,4,hipMemcpyHostToDevice)); HC(hipMemcpy(dsf,sf,8,hipMemcpyHostToDevice));
  HC(hipMemcpy(dld,&ldv,4,hipMemcpyHostToDevice)); HC(hipMemcpy(dinfo,&info,4,hipMemcpyHostToDevice)); HC(hipMemcpy(dpp,pp,16,hipMemcpyHostToDevice));
  long long rp[2]={0,n}; long long *drp; HC(hipMalloc(&drp,16)); HC(hipMemcpy(drp,rp,16,hipMemcpyHostToDevice));
  DevSym S{}; S.n=n; S.nsuper=1; S.sfirst=dsf; S.ld=dld; S.panelptr=dpp; S.rowptr=drp;
  hipStream_t st; HC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0,e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  const int reps=200; float ms;
  for(int form : {3}) {      // (the register-patch form of rounds 1-3 is gone from the library)
  auto launch = [&](hipStream_t q, double *dst, int wv) {
    const FrontArg fa{1,0,wv,wv,n,0,0};
    if (form == 3) { if (wv <= 16) hipLaunchKernelGGL((k_potrf64_b<1,1>),dim3(1),dim3(64),0,q,S,(const FrontView*)nullptr,0,dst,dinfo,fa);
      else if (wv <= 32) hipLaunchKernelGGL((k_potrf64_b<2,2>),dim3(1),dim3(128),0,q,S,(const FrontView*)nullptr,0,dst,dinfo,fa);
      else if (wv <= 48) hipLaunchKernelGGL((k_potrf64_b<4,3>),dim3(1),dim3(256),0,q,S,(const FrontView*)nullptr,0,dst,dinfo,fa);
      else hipLaunchKernelGGL((k_potrf64_b<8,4>),dim3(1),dim3(512),0,q,S,(const FrontView*)nullptr,0,dst,dinfo,fa); }
  };
  printf("==== %s\n", form == 3 ? "k_potrf64_b (16-column steps, 512 threads)" : "k_potrf64 (register patches, 256 threads)");
  for(int wv : {64, 37, 16, 3, 61, 48}) {
    int sf2[2]={0,wv}; HC(hipMemcpy(dsf,sf2,8,hipMemcpyHostToDevice));
    HC(hipEventRecord(e0,st));
    for(int r=0;r<reps;r++){ hipMemcpyAsync(dL,dA,n*n*8,hipMemcpyDeviceToDevice,st); launch(st,dL,wv); }
    HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st)); HC(hipEventElapsedTime(&ms,e0,e1));
    printf("k_potrf64 (w=%d) + d2d copy: %.2f us per launch\n", wv, ms*1000/reps);
#ifdef GMRFX_CYC
    if (form == 3) { long long c[64]; HC(hipMemcpyFromSymbol(c, HIP_SYMBOL(pb::g_pb_cyc), sizeof(c)));
      printf("  diagonal wave, cycles since kernel start: loaded %lld;", c[1]-c[0]);
      for (int d = 0; d < (wv+15)/16; d++) printf(" [d=%d: top %lld, diag done %lld, B1 out %lld, window A done %lld, B2 out %lld]", d, c[2+8*d]-c[0], c[3+8*d]-c[0], c[4+8*d]-c[0], c[5+8*d]-c[0], c[6+8*d]-c[0]);
      printf(" loop end %lld, stored %lld\n", c[40]-c[0], c[41]-c[0]);
      long long cw[8][64]; HC(hipMemcpyFromSymbol(cw, HIP_SYMBOL(pb::g_pb_cycw), sizeof(cw)));
      for (int v = 1; v < 8; v++) { printf("    wave %d:", v); for (int d = 0; d < (wv+15)/16; d++) printf(" [d=%d: B1 out %lld, A done %lld, B2 out %lld, ops done %lld, stores issued %lld]", d, cw[v][8*d]-c[0], cw[v][8*d+1]-c[0], cw[v][8*d+2]-c[0], cw[v][8*d+3]-c[0], cw[v][8*d+4]-c[0]); printf(" end %lld\n", cw[v][40]-c[0]); } }
#endif
    std::vector<double> Lh(n*n); HC(hipMemcpy(Lh.data(),dL,n*n*8,hipMemcpyDeviceToHost));
    double err=0, errx=0, errpad=0;
    for(int j=0;j<wv;j++) for(int i=j;i<wv;i++){ double s=0; for(int k=0;k<=j;k++) s+=Lh[i+k*n]*Lh[j+k*n]; { double e_=fabs(s-A[i+j*n]); if(!(e_<=err)) err=e_; } }
    // X = L^-1 lower, stored transposed in the strict upper part: X[i][b] at (b, i); diag(X) = 1/diag(L)
    for(int i=0;i<wv;i++) for(int b=0;b<=i;b++){ double s=0; for(int k=b;k<=i;k++){ double x = (k==b)? 1.0/Lh[b+b*n] : Lh[b+k*n]; s+=Lh[i+k*n]*x; } { double e_=fabs(s-(i==b?1.0:0.0)); if(!(e_<=errx)) errx=e_; } }
    for(int j=0;j<n;j++) for(int i=0;i<n;i++) if(i>=wv||j>=wv) errpad=fmax(errpad,fabs(Lh[i+j*n]-A[i+j*n]));
    printf("  max |LL'-A| = %.3e, max |L X - I| = %.3e, untouched outside w: %.1e\n", err, errx, errpad);
    HC(hipEventRecord(e0,st));
    for(int r=0;r<reps;r++){ hipMemcpyAsync(dL,dA,n*n*8,hipMemcpyDeviceToDevice,st); }
    HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st)); HC(hipEventElapsedTime(&ms,e0,e1));
    printf("  d2d copy alone: %.2f us\n", ms*1000/reps);
  }
  { HC(hipMemcpy(dL,dA,n*n*8,hipMemcpyDeviceToDevice));
    HC(hipEventRecord(e0,st));
    for(int r=0;r<reps;r++) launch(st,dL,64);
    HC(hipEventRecord(e1,st)); HC(hipStreamSynchronize(st)); HC(hipEventElapsedTime(&ms,e0,e1));
    printf("  back to back (no copy; refactoring its own output: timing only): %.2f us per launch\n", ms*1000/reps); }
  { // an indefinite block: the first non-positive pivot must be reported
    std::vector<double> B(A); B[10+10*n] = -1.0;
    int big = INT_MAX, got = 0;
    HC(hipMemcpy(dL,B.data(),n*n*8,hipMemcpyHostToDevice)); HC(hipMemcpy(dinfo,&big,4,hipMemcpyHostToDevice));
    launch(st,dL,64); HC(hipStreamSynchronize(st));
    HC(hipMemcpy(&got,dinfo,4,hipMemcpyDeviceToHost)); HC(hipMemcpy(dinfo,&big,4,hipMemcpyHostToDevice));
    printf("  bad pivot at column 10 -> info = %d\n", got); }
  }
  // do two chains of dependent one-workgroup launches on two streams run side by side? (the two panel chains of a level)
  { double *dL2; HC(hipMalloc(&dL2,n*n*8)); HC(hipMemcpy(dL2,dA,n*n*8,hipMemcpyDeviceToDevice)); HC(hipMemcpy(dL,dA,n*n*8,hipMemcpyDeviceToDevice));
    hipStream_t st2; HC(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    hipEvent_t f0,f1; HC(hipEventCreate(&f0)); HC(hipEventCreate(&f1));
    const FrontArg fa{1,0,64,64,n,0,0};
    for (int two = 0; two < 2; two++) {
      HC(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for(int r=0;r<reps;r++){
        hipLaunchKernelGGL((k_potrf64_b<8,4>),dim3(1),dim3(512),0,st,S,(const FrontView*)nullptr,0,dL,dinfo,fa);
        if (two) hipLaunchKernelGGL((k_potrf64_b<8,4>),dim3(1),dim3(512),0,st2,S,(const FrontView*)nullptr,0,dL2,dinfo,fa);
      }
      HC(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%d chain(s) of %d dependent k_potrf64 launches on %d stream(s): %.2f us per chain step (wall)\n", two + 1, reps, two + 1, us / reps);
    } }
  return 0;
}
