// The f64 matrix instruction against the f64 vector FMA on gfx950 ("one bounded probe").
// Question: could v_mfma_f64_16x16x4_f64 take the Z'VZ half of the condensed system and the Vxx recursion of a backward knot
// (ddp_optimizer.cpp:517-520, 622-628) off the VALU?  The answer needs two numbers only: what one such MFMA costs, alone and
// next to VALU work of the same wave and of co-resident waves, against what the same multiply-adds cost as v_fma_f64.
//   (1) issue cost: cycles per instruction of independent streams, one wave (s_memtime around 64 x 8 instructions);
//   (2) dependent chain (latency);
//   (3) co-issue: a wave of MFMAs next to a wave of FMAs on the same SIMD - do the pipes overlap?
// usage: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe && tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: 8 independent MFMA accumulators; 1: one dependent MFMA chain; 2: 8 independent v_fma_f64; 3: one dependent v_fma_f64 chain
// MODE 4: waves with even id run MODE 0, odd ones MODE 2 (blockDim 128..256: several waves per SIMD)
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
  d4 acc[8];
  double f[8];
  for (int i = 0; i < 8; i++) { acc[i] = (d4){0.0, 0.0, 0.0, 0.0}; f[i] = a + i; }
  const int mode = MODE == 4 ? ((wave & 1) ? 2 : 0) : MODE;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < 64; r++) {
    if (mode == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else if (mode == 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
    } else if (mode == 2) {
#pragma unroll
      for (int i = 0; i < 8; i++) f[i] = __builtin_fma(f[i], 1.0000001, 0.5);
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) f[0] = __builtin_fma(f[0], 1.0000001, 0.5);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w + f[i];
  out[threadIdx.x] = s;
  if (lane == 0) cyc[wave] = t1 - t0;
}

int main() {
  double* o;
  unsigned long long *c, h[8];
  hipMalloc(&o, 1024 * 8);
  hipMalloc(&c, 64);
  const char* names[] = {"v_mfma_f64_16x16x4_f64, 8 independent accumulators", "v_mfma_f64_16x16x4_f64, dependent chain",
                         "v_fma_f64, 8 independent chains", "v_fma_f64, dependent chain"};
#define RUN(M, T)                                                                              \
  hipLaunchKernelGGL(k<M>, dim3(1), dim3(T), 0, 0, o, c);                                      \
  hipLaunchKernelGGL(k<M>, dim3(1), dim3(T), 0, 0, o, c);                                      \
  hipDeviceSynchronize();                                                                      \
  hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
  for (int m = 0; m < 4; m++) {
    switch (m) {
      case 0: RUN(0, 64) break;
      case 1: RUN(1, 64) break;
      case 2: RUN(2, 64) break;
      default: RUN(3, 64)
    }
    printf("{\"what\": \"%s\", \"waves\": 1, \"cycles_per_instruction\": %.2f}\n", names[m], (double)h[0] / 512.0);
  }
  // 1024 multiply-adds per MFMA against 64 per FMA: cycles per 1024 multiply-adds, one wave
  // co-issue on one SIMD: one workgroup of 8 waves = two per SIMD (waves w and w + 4 share a SIMD): MFMA-only, FMA-only, mixed
  RUN(0, 512)
  printf("{\"what\": \"8 waves (2 per SIMD), all MFMA\", \"cycles_per_instruction_per_wave\": %.2f}\n", (double)h[0] / 512.0);
  RUN(2, 512)
  printf("{\"what\": \"8 waves (2 per SIMD), all v_fma_f64\", \"cycles_per_instruction_per_wave\": %.2f}\n", (double)h[0] / 512.0);
  RUN(4, 512)
  printf("{\"what\": \"8 waves (2 per SIMD), even waves MFMA / odd waves v_fma_f64\", \"mfma_wave\": %.2f, \"fma_wave\": %.2f}\n",
         (double)h[0] / 512.0, (double)h[1] / 512.0);
  return 0;
}
