// Profiling aid (GPU box): sustained v_mfma_f32_16x16x32_bf16 rate and the shader clock under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_clock.hip -o build_variants/mfma_clock && build_variants/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((vector_size(16)));
typedef __bf16 bf16x8 __attribute__((vector_size(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(blockIdx.x + i); }
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int NACC>
void run(int wgs, int iters) {
  float* out; long long* clk;
  hipMalloc(&out, wgs * 256 * 4); hipMalloc(&clk, wgs * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<wgs, 256>>>(out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<wgs, 256>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mfmas = (double)wgs * 4 * iters * NACC;
  const double flops = mfmas * 16 * 16 * 32 * 2;
  printf("NACC=%d wgs=%d: %.3f ms  %.0f TFLOP/s bf16 dense | wg0: %lld clock64 ticks, %lld wall ticks (100 MHz) -> clock64 runs at %.0f MHz | "
         "%.1f clock64 ticks per MFMA per wave\n", NACC, wgs, ms, flops / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 100.0,
         (double)h[0] / ((double)iters * NACC));
  hipFree(out); hipFree(clk);
}
int main() {
  run<8>(256, 20000);
  run<8>(1024, 5000);
  run<2>(1024, 20000);
  run<1>(1024, 40000);
  run<8>(64, 20000);
  return 0;
}
