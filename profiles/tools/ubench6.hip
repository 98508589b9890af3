// What LDS-DMA span staging takes away from the other pipes of a CU (12 wavefronts, each staging an 8 KB span per
// round like the front-end kernel): rounds of {8 x global_load_lds_dwordx4, WORK, wait for the span at the top of the
// next round} against the same rounds without the DMA, for WORK = packed FMAs, LDS reads, LDS writes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
enum { W_FMA, W_LDSR, W_LDSW, W_MIX, W_FMA_AFTER_LDS_WAIT };
template <int WORK, bool DMA>
__global__ void __launch_bounds__(768) k(const float* src, float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* X = lds + wave * 3072;          // 2048 floats of span + 1024 floats the LDS work uses
  float* Y = X + 2048;
  const float* g = src + (size_t)blockIdx.x * 12 * 16384 + wave * 16384 + 4 * lane;
  v2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f};
  v4 acc = {0, 0, 0, 0};
  unsigned long long ticks = 0;
  for (int it = 0; it < iters; ++it) {
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the span ordered in the previous round
      const float* gi = g + (it & 7) * 1280;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const auto* gs = (const __attribute__((address_space(1))) void*)(gi + 1024 * q);
        auto* ld = (__attribute__((address_space(3))) void*)(X + 1024 * q);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 3072, 0);
      }
    }
    if (WORK == W_FMA_AFTER_LDS_WAIT) {  // one LDS read completed (lgkmcnt(0)) between the DMA issue and the arithmetic
      float r;
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(size_t)(Y + lane)) : "memory");
      ticks += __builtin_amdgcn_s_memtime() - t0;
      acc.x += r;
    }
    if (WORK == W_FMA || WORK == W_MIX || WORK == W_FMA_AFTER_LDS_WAIT) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        a = __builtin_elementwise_fma(a, b, c);
        d = __builtin_elementwise_fma(d, b, c);
        a = __builtin_elementwise_fma(a, c, b);
        d = __builtin_elementwise_fma(d, c, b);
      }
    }
    if (WORK == W_LDSR || WORK == W_MIX) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += *reinterpret_cast<const v4*>(Y + ((4 * lane + 4 * i) & 1023));
    }
    if (WORK == W_LDSW || WORK == W_MIX) {
#pragma unroll
      for (int i = 0; i < 32; ++i) *reinterpret_cast<v2*>(Y + ((2 * lane + 2 * i) & 1023)) = a + d;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 768 + t] = X[lane] + acc.x + acc.y + acc.z + acc.w + a.x + a.y + d.x + d.y;
  if (WORK == W_FMA_AFTER_LDS_WAIT && blockIdx.x == 0 && t == 64) printf("    (ds_read_b32 + lgkmcnt(0) right after the DMA issue: %.0f s_memtime ticks on average)\n", (double)ticks / iters);
}
template <int WORK, bool DMA>
double run(const float* s, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<WORK, DMA>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 147456, 0, s, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 147456, 0, s, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
template <int WORK>
void both(const char* name, const float* s, float* d) {
  const double off = run<WORK, false>(s, d), on = run<WORK, true>(s, d);
  printf("%-44s %8.1f ns per round without DMA  %8.1f ns with 12 x 8 KB of DMA per round  (+%.1f ns = %.2f ns per KB)\n", name, off, on,
         on - off, (on - off) / 96.0);
}
int main() {
  float *s, *d;
  const size_t n = (size_t)256 * 12 * 16384 + 65536;
  (void)hipMalloc(&s, n * 4);
  (void)hipMemset(s, 0, n * 4);
  (void)hipMalloc(&d, 256 * 768 * 4);
  run<W_FMA, false>(s, d);
  both<W_FMA>("256 pk_fma per wave", s, d);
  both<W_FMA_AFTER_LDS_WAIT>("ds_read_b32, lgkmcnt(0), 256 pk_fma", s, d);
  both<W_LDSR>("32 ds_read_b128 per wave", s, d);
  both<W_LDSW>("32 ds_write_b64 per wave", s, d);
  both<W_MIX>("256 pk_fma + 32 read_b128 + 32 write_b64", s, d);
  return 0;
}
