// Issue cost of f64 transcendental instructions on gfx950, one wave: cycles per instruction of independent
// v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 / v_rcp_f32 / v_rsq_f32 streams against v_fma_f64 (s_memtime around 8 x 64 ops).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(double* out, unsigned long long* cyc) {
  double a[8];
  float f[8];
  for (int i = 0; i < 8; i++) { a[i] = 1.0 + threadIdx.x * 0.001 + i; f[i] = (float)a[i]; }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);
      if (OP == 1) a[i] = __builtin_amdgcn_rcp(a[i]);
      if (OP == 2) a[i] = __builtin_amdgcn_rsq(a[i]);
      if (OP == 3) a[i] = __builtin_amdgcn_sqrt(a[i]);
      if (OP == 4) f[i] = __builtin_amdgcn_rcpf(f[i]);
      if (OP == 5) f[i] = __builtin_amdgcn_rsqf(f[i]);
      if (OP == 6) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(((int*)f)[i]) : "v"(((int*)f)[i]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + f[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* o; unsigned long long *c, h;
  hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
  const char* names[] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f32", "v_rsq_f32"};
#define RUN(OP) hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, o, c); hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, o, c); hipDeviceSynchronize(); \
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-12s %.1f cycles / instruction (one wave, 8 independent chains)\n", names[OP], (double)h / 512.0);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
