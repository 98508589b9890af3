// Do LDS reads and VALU work of the waves of one SIMD overlap?  Each wave alternates a burst of ds_read_b64 and a
// burst of v_fma_f32 of about equal duration; 3 waves per SIMD.  Compared with the two bursts run alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters) {
  __shared__ __align__(16) float lds[8192];
  const int t = threadIdx.x;
  for (int i = t; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  v2 a0 = {1.f + t, 2.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  float f0 = t, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
  const float c = 0.999f;
  const unsigned lp = (unsigned)(size_t)(lds) + (t & 63) * 8 + ((t >> 6) & 3) * 4096;
  for (int it = 0; it < iters; ++it) {
    if (KIND & 1) {  // 64 ds_read_b64
      REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lp));)
    }
    if (KIND & 2) {  // 320 v_fma_f32
      REP4(REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));))
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int KIND>
void run(const char* name, float* d, int waves_per_simd) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %d waves/SIMD: %.1f ns per wave-iteration per SIMD\n", name, waves_per_simd, ms * 1e6 / ((double)iters * waves_per_simd));
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  run<2>("warm-up", d, 3);
  for (int w : {1, 2, 3}) {
    run<1>("64 ds_read_b64", d, w);
    run<2>("328 v_fma_f32", d, w);
    run<3>("64 ds_read_b64 + 328 v_fma_f32", d, w);
  }
  return 0;
}
