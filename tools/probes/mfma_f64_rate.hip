// Measurement aid: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- wall time and shader cycles (s_memtime) per MFMA for 1, 2, 4, 6
// independent accumulators, one or two wavefronts per SIMD, every CU busy.   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(int iters, double *out, unsigned long long *cyc)
{
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
static void run(int wg_threads, int iters)
{
  double *out;
  unsigned long long *cyc, h = 0;
  hipMalloc(&out, sizeof(double) * 256 * 512);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(wg_threads), 0, 0, 16, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(wg_threads), 0, 0, iters, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * NACC; // MFMAs per wave
  const double waves_per_simd = wg_threads / 256.0;
  printf("acc %d  waves/SIMD %.0f  wall %.3f ms  -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s;  s_memtime: %.1f ticks per MFMA of one wave (100 MHz ticks x 24 = shader cycles at 2.4 GHz: %.0f)\n", NACC,
         waves_per_simd, ms, ms * 1e6 / (nm * waves_per_simd), 2048.0 * nm * (wg_threads / 64) * 256 / (ms * 1e-3) / 1e12, (double)h / nm, (double)h / nm * 24);
  hipFree(out);
  hipFree(cyc);
}
int main()
{
  const int iters = 20000;
  run<1>(256, iters);
  run<2>(256, iters);
  run<4>(256, iters);
  run<6>(256, iters);
  run<1>(512, iters);
  run<4>(512, iters);
  return 0;
}
