// Calibration microbenchmark for the HBM traffic counters and the measured HBM peak.
// Streams a buffer of known size through four kernels - dword-per-lane read / write (the access width of the
// DDP sweeps) and dwordx4-per-lane read / write (the width MI355X_MICROARCH.md calibrates FETCH_SIZE for) - and a
// device-to-device hipMemcpy.  Run plainly it prints achieved GB/s per kernel; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// the per-kernel counter value divided by the known byte count is the correction factor for that width.
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib   (tools/profile_round.sh does it)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__global__ void calib_read_dword(const float* __restrict__ src, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 123.456f) *sink = acc;  // never true: keeps the loads alive
}
__global__ void calib_write_dword(float* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 1.0f;
}
__global__ void calib_read_dwordx4(const float4* __restrict__ src, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *sink = acc;
}
__global__ void calib_write_dwordx4(float4* __restrict__ dst, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// The output-sampling kernel's store pattern (direct_amd/csrc/traj_sample.h, float storage): one wavefront per trajectory
// walks chunks of 64 samples and writes, per chunk, 64 x 12 bytes to each of THREE arrays (positions, velocities,
// accelerations: one global_store_dwordx3 per lane and array).  Pure writes, no evaluation: the ceiling of that pattern.
// `mode` 0: dwordx3 per lane as the kernel does; 1: the same bytes as three dword stores per lane and array, lane-contiguous
// (what a transpose through LDS would issue); 2: as four-dword stores (48 lanes x 16 bytes per 768-byte chunk).
__global__ __launch_bounds__(64) void calib_write_sampler(float* __restrict__ p0, float* __restrict__ p1, float* __restrict__ p2,
                                                           size_t floats_per_traj, int mode) {
  const size_t base = (size_t)blockIdx.x * floats_per_traj;
  const int lane = threadIdx.x;
  float* arr[3] = {p0 + base, p1 + base, p2 + base};
  for (size_t off = 0; off + 192 <= floats_per_traj; off += 192) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float* d = arr[a] + off;
      if (mode == 0) {
        float3 v = make_float3(1.f + a, 2.f, 3.f);
        *reinterpret_cast<float3*>(d + 3 * lane) = v;
      } else if (mode == 1) {
        d[lane] = 1.f + a;
        d[64 + lane] = 2.f;
        d[128 + lane] = 3.f;
      } else {
        if (lane < 48) *reinterpret_cast<float4*>(d + 4 * lane) = make_float4(1.f + a, 2.f, 3.f, 4.f);
      }
    }
  }
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20;  // MiB, default 2 GiB
  const size_t n = bytes / 4;
  float *a, *b, *sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * 32, block = 256, reps = 5;
  auto time = [&](const char* name, auto launch, double moved) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"%s\", \"bytes_per_launch\": %.0f, \"ms\": %.4f, \"GBs\": %.1f}\n", name, moved, ms / reps,
           moved / (ms / reps * 1e-3) / 1e9);
  };
  time("calib_read_dword", [&] { hipLaunchKernelGGL(calib_read_dword, dim3(grid), dim3(block), 0, 0, a, n, sink); }, (double)bytes);
  time("calib_write_dword", [&] { hipLaunchKernelGGL(calib_write_dword, dim3(grid), dim3(block), 0, 0, b, n); }, (double)bytes);
  time("calib_read_dwordx4", [&] { hipLaunchKernelGGL(calib_read_dwordx4, dim3(grid), dim3(block), 0, 0, (const float4*)a, n / 4, sink); }, (double)bytes);
  time("calib_write_dwordx4", [&] { hipLaunchKernelGGL(calib_write_dwordx4, dim3(grid), dim3(block), 0, 0, (float4*)b, n / 4); }, (double)bytes);
  time("hipMemcpy_d2d(read+write)", [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 2.0 * bytes);
  {  // the sampler's pattern: 32768 trajectories x 20544 samples (tools/sample_bench.py: B = 32768, N = 100, dt = 0.02: 673 M points)
    const int traj = 32768;
    const size_t fpt = 20544 * 3;  // floats per trajectory and array
    float *q0, *q1, *q2;
    const size_t ab = (size_t)traj * fpt * 4;
    CK(hipMalloc(&q0, ab)); CK(hipMalloc(&q1, ab)); CK(hipMalloc(&q2, ab));
    const char* names[3] = {"sampler_pattern_dwordx3_x3arrays", "sampler_pattern_dword_lane_contiguous_x3arrays", "sampler_pattern_dwordx4_x3arrays"};
    for (int mode = 0; mode < 3; mode++)
      time(names[mode], [&] { hipLaunchKernelGGL(calib_write_sampler, dim3(traj), dim3(64), 0, 0, q0, q1, q2, fpt, mode); },
           3.0 * (double)traj * (double)(fpt / 192 * 192) * 4.0);
    CK(hipFree(q0)); CK(hipFree(q1)); CK(hipFree(q2));
  }
  return 0;
}
