// Which clock does the GPU run at under a light and under a heavy arithmetic load?  argv[1] = wavefronts per CU (1..16),
// argv[2] = seconds.  Sample `rocm-smi --showclocks` meanwhile (profiles/tools/clocks2.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  v2 a = {1.0f + threadIdx.x, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f}, e = {2.0f, 5.0f}, f = {7.0f, 1.5f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      a = __builtin_elementwise_fma(a, b, c); d = __builtin_elementwise_fma(d, b, c);
      e = __builtin_elementwise_fma(e, c, b); f = __builtin_elementwise_fma(f, c, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + d.y + e.x + f.y;
}
int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 1;
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  const int iters = argc > 3 ? atoi(argv[3]) : 20000;  // 20000: launches of 11 .. 28 ms; 110: launches of about 150 us
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double total = 0; long long launches = 0; float ms = 0;
  const int per_sync = iters >= 20000 ? 1 : 200;  // short launches are queued back to back, as kbench does
  while (total < secs * 1e3) {
    (void)hipEventRecord(e0);
    for (int j = 0; j < per_sync; ++j) hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    total += ms; launches += per_sync;
    ms /= per_sync;
  }
  // 20000 x 256 packed FMAs per wavefront and launch
  printf("waves/CU %d: last launch %.2f ms -> %.3f ns per v_pk_fma_f32 per SIMD (%d wavefront(s) per SIMD)\n", waves, ms,
         ms * 1e6 / ((double)iters * 256) / ((waves + 3) / 4), (waves + 3) / 4);
  return 0;
}
