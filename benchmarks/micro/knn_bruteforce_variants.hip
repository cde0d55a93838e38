// Scratch microbenchmark: brute-force exact 1-NN variants on gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <vector>
#include <cmath>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float f2 __attribute__((ext_vector_type(2)));

// Variant A: straightforward, uniform scalar loads, idx tracking every pair.
__global__ __launch_bounds__(256) void knnA(const float* __restrict__ sx,const float* __restrict__ sy,const float* __restrict__ sz,int n,
    const float* __restrict__ tx,const float* __restrict__ ty,const float* __restrict__ tz,int m, int* __restrict__ oi, float* __restrict__ od){
  int q = blockIdx.x*256+threadIdx.x; if(q>=n) return;
  float px=sx[q],py=sy[q],pz=sz[q];
  float best=FLT_MAX; int bi=-1;
  for(int j=0;j<m;j++){
    float dx=px-tx[j],dy=py-ty[j],dz=pz-tz[j];
    float d=(dx*dx+dy*dy)+dz*dz;
    if(d<best){best=d;bi=j;}
  }
  oi[q]=bi; od[q]=best;
}

// Variant B: packed-f32 math, min3 chunk filter, rare rescan; 4 waves split the target range.
template<int CH>
__global__ __launch_bounds__(256) void knnB(const float* __restrict__ sx,const float* __restrict__ sy,const float* __restrict__ sz,int n,
    const float* __restrict__ tx,const float* __restrict__ ty,const float* __restrict__ tz,int mpad, int* __restrict__ oi, float* __restrict__ od){
  __shared__ float sd[4][64]; __shared__ int si[4][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int q = blockIdx.x*64+lane; int qq = q<n?q:n-1;
  float px=sx[qq],py=sy[qq],pz=sz[qq];
  f2 px2={px,px},py2={py,py},pz2={pz,pz};
  const int nch = mpad/CH;             // chunks total
  const int c0 = (nch*w)/4, c1=(nch*(w+1))/4;
  float best=FLT_MAX; int bi=-1;
  for(int c=c0;c<c1;c++){
    const int j0=c*CH;
    float mm=FLT_MAX;
    #pragma unroll
    for(int k=0;k<CH;k+=2){
      f2 qx=*(const f2*)(tx+j0+k), qy=*(const f2*)(ty+j0+k), qz=*(const f2*)(tz+j0+k);
      f2 dx=px2-qx, dy=py2-qy, dz=pz2-qz;
      f2 s=(dx*dx+dy*dy)+dz*dz;
      mm=fminf(mm,fminf(s.x,s.y));
    }
    if(mm<best){
      for(int k=0;k<CH;k++){
        float dx=px-tx[j0+k],dy=py-ty[j0+k],dz=pz-tz[j0+k];
        float d=(dx*dx+dy*dy)+dz*dz;
        if(d<best){best=d;bi=j0+k;}
      }
    }
  }
  sd[w][lane]=best; si[w][lane]=bi;
  __syncthreads();
  if(w==0 && q<n){
    #pragma unroll
    for(int k=1;k<4;k++){ float d=sd[k][lane]; int i=si[k][lane]; if(d<best){best=d;bi=i;} }
    oi[q]=bi; od[q]=best;
  }
}

// Variant C: Q queries per lane, targets staged in LDS tiles, broadcast ds_read; packed math; chunk filter.
template<int Q,int TILE,int CH>
__global__ __launch_bounds__(256) void knnC(const float* __restrict__ sx,const float* __restrict__ sy,const float* __restrict__ sz,int n,
    const float* __restrict__ tx,const float* __restrict__ ty,const float* __restrict__ tz,int mpad, int* __restrict__ oi, float* __restrict__ od){
  __shared__ float lx[TILE],ly[TILE],lz[TILE];
  float px[Q],py[Q],pz[Q],best[Q]; int bi[Q];
  #pragma unroll
  for(int u=0;u<Q;u++){ int q=(blockIdx.x*Q+u)*256+threadIdx.x; int qq=q<n?q:n-1; px[u]=sx[qq];py[u]=sy[qq];pz[u]=sz[qq];best[u]=FLT_MAX;bi[u]=-1; }
  for(int t0=0;t0<mpad;t0+=TILE){
    __syncthreads();
    for(int k=threadIdx.x;k<TILE;k+=256){lx[k]=tx[t0+k];ly[k]=ty[t0+k];lz[k]=tz[t0+k];}
    __syncthreads();
    for(int c=0;c<TILE;c+=CH){
      float mm[Q];
      #pragma unroll
      for(int u=0;u<Q;u++) mm[u]=FLT_MAX;
      #pragma unroll
      for(int k=0;k<CH;k+=2){
        f2 qx=*(const f2*)(lx+c+k), qy=*(const f2*)(ly+c+k), qz=*(const f2*)(lz+c+k);
        #pragma unroll
        for(int u=0;u<Q;u++){
          f2 p2x={px[u],px[u]},p2y={py[u],py[u]},p2z={pz[u],pz[u]};
          f2 dx=p2x-qx, dy=p2y-qy, dz=p2z-qz;
          f2 s=(dx*dx+dy*dy)+dz*dz;
          mm[u]=fminf(mm[u],fminf(s.x,s.y));
        }
      }
      bool any=false;
      #pragma unroll
      for(int u=0;u<Q;u++) any|=(mm[u]<best[u]);
      if(any){
        for(int k=0;k<CH;k++){
          float qx=lx[c+k],qy=ly[c+k],qz=lz[c+k];
          #pragma unroll
          for(int u=0;u<Q;u++){
            float dx=px[u]-qx,dy=py[u]-qy,dz=pz[u]-qz;
            float d=(dx*dx+dy*dy)+dz*dz;
            if(d<best[u]){best[u]=d;bi[u]=t0+c+k;}
          }
        }
      }
    }
  }
  #pragma unroll
  for(int u=0;u<Q;u++){ int q=(blockIdx.x*Q+u)*256+threadIdx.x; if(q<n){oi[q]=bi[u];od[q]=best[u];} }
}

int main(int argc,char**argv){
  int n = argc>1?atoi(argv[1]):370488; int m=n; int coherent = argc>2?atoi(argv[2]):0;
  int mpad=((m+1023)/1024)*1024;
  std::vector<float> hs(3*(size_t)n), ht(3*(size_t)mpad);
  srand(1);
  auto rnd=[&](){return (float)rand()/RAND_MAX*16.f-8.f;};
  for(int i=0;i<n;i++){
    if(coherent){ float a=(float)i/n*6.28f*40, r=2+3.f*i/n; hs[i]=r*cosf(a)+0.01f*rnd(); hs[n+i]=r*sinf(a)+0.01f*rnd(); hs[2*n+i]=(float)i/n*2.6f; }
    else { hs[i]=rnd();hs[n+i]=rnd();hs[2*n+i]=rnd()*0.16f; }
  }
  for(int i=0;i<mpad;i++){
    if(i>=m){ht[i]=INFINITY;ht[mpad+i]=INFINITY;ht[2*mpad+i]=INFINITY;continue;}
    if(coherent){ float a=(float)i/m*6.28f*40+0.01f, r=2+3.f*i/m; ht[i]=r*cosf(a)+0.01f*rnd(); ht[mpad+i]=r*sinf(a)+0.01f*rnd(); ht[2*mpad+i]=(float)i/m*2.6f+0.02f; }
    else { ht[i]=rnd();ht[mpad+i]=rnd();ht[2*mpad+i]=rnd()*0.16f; }
  }
  float *ds,*dt,*od0,*od1; int *oi0,*oi1;
  CK(hipMalloc(&ds,hs.size()*4)); CK(hipMalloc(&dt,ht.size()*4));
  CK(hipMalloc(&od0,n*4));CK(hipMalloc(&od1,n*4));CK(hipMalloc(&oi0,n*4));CK(hipMalloc(&oi1,n*4));
  CK(hipMemcpy(ds,hs.data(),hs.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dt,ht.data(),ht.size()*4,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0));CK(hipEventCreate(&e1));
  std::vector<int> r0(n),r1(n); std::vector<float> f0(n),f1(n);
  auto timeit=[&](const char*name,auto launch,int reps,bool check){
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=reps;
    double pairs=(double)n*m;
    printf("%-28s %9.3f ms  %8.2f Gpair/s", name, ms, pairs/ms*1e-6);
    if(check){ CK(hipMemcpy(r1.data(),oi1,n*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(f1.data(),od1,n*4,hipMemcpyDeviceToHost));
      long bad=0; for(int i=0;i<n;i++) if(r1[i]!=r0[i]||memcmp(&f1[i],&f0[i],4)) bad++; printf("  mismatches=%ld",bad); }
    printf("\n"); fflush(stdout);
  };
  timeit("A scalar idx-track", [&]{ hipLaunchKernelGGL(knnA,dim3((n+255)/256),dim3(256),0,0,ds,ds+n,ds+2*n,n,dt,dt+mpad,dt+2*mpad,m,oi0,od0); },1,false);
  CK(hipMemcpy(r0.data(),oi0,n*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(f0.data(),od0,n*4,hipMemcpyDeviceToHost));
  #define RUNB(CH) timeit("B pk s_load CH=" #CH, [&]{ CK(hipMemset(oi1,0xff,n*4)); hipLaunchKernelGGL(knnB<CH>,dim3((n+63)/64),dim3(256),0,0,ds,ds+n,ds+2*n,n,dt,dt+mpad,dt+2*mpad,mpad,oi1,od1); },2,true);
  RUNB(16) RUNB(32) RUNB(64)
  #define RUNC(Q,TILE,CH) timeit("C lds Q=" #Q " T=" #TILE " CH=" #CH, [&]{ CK(hipMemset(oi1,0xff,n*4)); hipLaunchKernelGGL((knnC<Q,TILE,CH>),dim3((n+256*Q-1)/(256*Q)),dim3(256),0,0,ds,ds+n,ds+2*n,n,dt,dt+mpad,dt+2*mpad,mpad,oi1,od1); },2,true);
  RUNC(1,1024,32) RUNC(2,1024,32) RUNC(4,1024,32) RUNC(2,1024,64) RUNC(4,1024,16) RUNC(4,2048,32)
  return 0;
}
