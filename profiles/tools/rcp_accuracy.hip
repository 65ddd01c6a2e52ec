#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x*blockDim.x+threadIdx.x; if (i>=n) return;
  double d = x[i];
  double a = __builtin_amdgcn_rcp(d);
  r0[i] = a;
  double e = fma(-d, a, 1.0); a = fma(a, e, a); r1[i] = a;
  e = fma(-d, a, 1.0); a = fma(a, e, a); r2[i] = a;
}
int main() {
  const int n = 1<<20; double *x,*r0,*r1,*r2;
  hipMallocManaged(&x,n*8); hipMallocManaged(&r0,n*8); hipMallocManaged(&r1,n*8); hipMallocManaged(&r2,n*8);
  for (int i=0;i<n;i++) x[i] = std::ldexp(1.0 + (double)rand()/RAND_MAX, (rand()%200)-100) * ((rand()&1)?1:-1);
  k<<<n/256,256>>>(x,r0,r1,r2,n); hipDeviceSynchronize();
  double m0=0,m1=0,m2=0;
  for (int i=0;i<n;i++){ double t=1.0/x[i]; m0=fmax(m0,fabs(r0[i]/t-1)); m1=fmax(m1,fabs(r1[i]/t-1)); m2=fmax(m2,fabs(r2[i]/t-1)); }
  printf("max rel err: rcp %.3e  +1NR %.3e  +2NR %.3e\n", m0,m1,m2);
}
