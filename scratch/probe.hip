#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void k_axpy(const float* x, float* y, float a, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) y[i]=a*x[i]+y[i];}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_mfma(const float* A,const float* B,float* C){ // 32x32x2: A[32][2], B[2][32]
  int l=threadIdx.x; f32x16 acc={0};
  float a=A[(l&31)*2+(l>>5)], b=B[(l>>5)*32+(l&31)];
  acc=__builtin_amdgcn_mfma_f32_32x32x2f32(a,b,acc,0,0,0);
  for(int r=0;r<16;r++){int row=(r&3)+8*(r>>2)+4*(l>>5); C[row*32+(l&31)]=acc[r];}
}
extern "C" int probe_axpy(const float* x,float* y,float a,int n,void* stream){
  hipLaunchKernelGGL(k_axpy,dim3((n+255)/256),dim3(256),0,(hipStream_t)stream,x,y,a,n); return (int)hipGetLastError();}
extern "C" int probe_mfma(const float* A,const float* B,float* C,void* stream){
  hipLaunchKernelGGL(k_mfma,dim3(1),dim3(64),0,(hipStream_t)stream,A,B,C); return (int)hipGetLastError();}
extern "C" int probe_rtver(){int v=0; hipRuntimeGetVersion(&v); return v;}
