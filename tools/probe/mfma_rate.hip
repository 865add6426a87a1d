// Calibration: what does one wave per SIMD (and two) sustain on v_mfma_f32_32x32x16_bf16 -- alone, with one ds_read_b128 per MFMA
// (the direct convolution's inner loop shape), and with accumulator rotation over 4 independent tiles?  Prints TFLOP/s and
// shader-clock cycles per MFMA per wave.     hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int LDSREAD, int NACC>
__global__ __launch_bounds__(512, 1) void k(float* out, long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
  __syncthreads();
  bf16x8_t a, b[NACC];
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(0.001f * (lane + e));
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.002f * (lane + e + j));
  f32x16_t acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // conflict-free b128 pattern: lane-linear 16-B chunks
  unsigned off = (unsigned)((t & 63) * 16 + (t >> 6) * 4096);
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (LDSREAD) {
          b[j] = *reinterpret_cast<const bf16x8_t*>(smem + ((off + (unsigned)(u * NACC + j) * 1024u) & 32767u));
        }
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * blockDim.x + t] = s;
  if (t == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int LDSREAD, int NACC>
void run(const char* name, int threads, int iters) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  const int grid = prop.multiProcessorCount;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * grid * threads); hipMalloc(&cyc, sizeof(long long) * grid);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<LDSREAD, NACC>), dim3(grid), dim3(threads), 65536, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<LDSREAD, NACC>), dim3(grid), dim3(threads), 65536, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(grid); hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)iters * 8 * NACC;
  const double flops = mfma_per_wave * (threads / 64) * grid * 32768.0;
  printf("%-44s %2d waves/CU  %7.3f ms  %7.1f TFLOP/s  wall/MFMA/wave %.1f ns  counter ticks/MFMA %.1f\n", name, threads / 64, ms,
         flops / ms / 1e9, ms * 1e6 / mfma_per_wave, (double)h[0] / mfma_per_wave);
  hipFree(out); hipFree(cyc);
}

// sustained mode (any argument): the MFMA-only and the MFMA + LDS-read loops for ~4 s each, rate printed per launch batch -- long
// enough for the package power controller to settle and for tools/probe/power_watch.py-style sampling beside it
template <int LDSREAD>
void sustained(const char* name, double secs) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  const int grid = prop.multiProcessorCount, threads = 256, iters = 40000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * grid * threads); hipMalloc(&cyc, sizeof(long long) * grid);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<LDSREAD, 4>), dim3(grid), dim3(threads), 65536, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms;
    const double flops = 10.0 * iters * 8 * 4 * (threads / 64) * grid * 32768.0;
    printf("%-40s t=%5.2f s  %7.1f TFLOP/s\n", name, total_ms / 1e3, flops / ms / 1e9);
    fflush(stdout);
  }
  hipFree(out); hipFree(cyc);
}

int main(int argc, char**) {
  if (argc > 1) {
    sustained<0>("sustained: mfma only, 4 acc, 4 waves/CU", 4.0);
    sustained<1>("sustained: mfma + ds_read_b128 per MFMA", 4.0);
    return 0;
  }
  const int it = 4000;
  run<0, 4>("mfma only, 4 accumulators", 256, it);
  run<0, 4>("mfma only, 4 accumulators", 512, it);
  run<0, 2>("mfma only, 2 accumulators", 256, it);
  run<0, 1>("mfma only, 1 accumulator (dependent chain)", 256, it);
  run<1, 4>("mfma + ds_read_b128 per MFMA, 4 acc", 256, it);
  run<1, 4>("mfma + ds_read_b128 per MFMA, 4 acc", 512, it);
  return 0;
}
