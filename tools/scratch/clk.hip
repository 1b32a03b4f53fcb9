#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void k_dep(unsigned long long* out, int n, T b) {
  unsigned long long t0 = __builtin_readcyclecounter();
  T a = threadIdx.x * (T)1e-9;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 32; j++) a = fma(a, b, (T)1e-9);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = (unsigned long long)(a * 1e6); }
}
template <typename T>
__global__ void k_indep(unsigned long long* out, int n, T b) {
  unsigned long long t0 = __builtin_readcyclecounter();
  T a[8];
  for (int j = 0; j < 8; j++) a[j] = threadIdx.x * (T)1e-9 + j;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < 8; j++) a[j] = fma(a[j], b, (T)1e-9);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  T s = 0; for (int j = 0; j < 8; j++) s += a[j];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = (unsigned long long)(s * 1e6); }
}
__global__ void k_lds(unsigned long long* out, int n) {
  __shared__ double buf[256];
  buf[threadIdx.x] = threadIdx.x; buf[threadIdx.x + 64] = 1; buf[threadIdx.x+128] = 2; buf[threadIdx.x+192]=3;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  int idx = threadIdx.x;
  double acc = 0;
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) { double v = buf[idx]; idx = ((int)v + j) & 255; acc += v; }   // dependent LDS chain
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = (unsigned long long)acc + idx; }
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 64);
  unsigned long long h[3];
  const int n = 20000;
  for (int blocks : {1, 3072}) {
    hipLaunchKernelGGL(k_dep<double>, dim3(blocks), dim3(64), 0, 0, d, n, 1.0000001); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %d: dependent fma_f64: %.2f cycles each\n", blocks, (double)h[0] / (n * 32.0));
    hipLaunchKernelGGL(k_dep<float>, dim3(blocks), dim3(64), 0, 0, d, n, 1.0000001f); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %d: dependent fma_f32: %.2f cycles each\n", blocks, (double)h[0] / (n * 32.0));
    hipLaunchKernelGGL(k_indep<double>, dim3(blocks), dim3(64), 0, 0, d, n, 1.0000001); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %d: independent fma_f64 (8 chains): %.2f cycles each\n", blocks, (double)h[0] / (n * 32.0));
    hipLaunchKernelGGL(k_indep<float>, dim3(blocks), dim3(64), 0, 0, d, n, 1.0000001f); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %d: independent fma_f32 (8 chains): %.2f cycles each\n", blocks, (double)h[0] / (n * 32.0));
    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, d, n); (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %d: dependent LDS read_b64 round trip (incl. cvt+and+add): %.2f cycles each\n", blocks, (double)h[0] / (n * 16.0));
  }
  return 0;
}
