// Measurement aid: v_mfma_f64_16x16x4_f64 with the accumulator in AGPRs ("a") against ArchVGPRs ("v"), issue-bound loop, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int FORM, int NACC>
__global__ __launch_bounds__(256) void k(int iters, double *out, unsigned long long *cyc)
{
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (FORM == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      else if (FORM == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int FORM, int NACC>
static void run(const char *name, int iters)
{
  double *out;
  unsigned long long *cyc, h = 0;
  hipMalloc(&out, sizeof(double) * 256 * 256);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FORM, NACC>), dim3(256), dim3(256), 0, 0, 16, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<FORM, NACC>), dim3(256), dim3(256), 0, 0, iters, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double o = 0;
  hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * NACC;
  printf("%-22s acc %d: %.1f ns per MFMA per SIMD, %.1f TFLOP/s, %.1f cycles per MFMA (wave 0), out[0] = %.6e\n", name, NACC, ms * 1e6 / nm, 2048.0 * nm * 4 * 256 / (ms * 1e-3) / 1e12, (double)h / nm, o);
}
int main()
{
  const int iters = 20000;
  run<0, 3>("asm, AGPR accumulator", iters);
  run<1, 3>("asm, VGPR accumulator", iters);
  run<2, 3>("builtin", iters);
  run<0, 1>("asm, AGPR accumulator", iters);
  run<1, 1>("asm, VGPR accumulator", iters);
  run<2, 1>("builtin", iters);
  return 0;
}
