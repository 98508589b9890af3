// Cost of staging PCM into LDS on one CU (12 wavefronts): LDS-DMA (global_load_lds) in 16-byte and 4-byte forms against
// 16-byte global loads into registers followed by ds_write_b128, all from an L2-resident buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
enum { K_DMA16, K_DMA4, K_REG16, K_REG16_NOLDS, K_REG8_NOLDS };
template <int KIND>
__global__ void __launch_bounds__(768) k(const float* src, float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* X = lds + wave * 3072;
  const float* g = src + (size_t)blockIdx.x * 65536 + wave * 4096 + 4 * lane;
  v4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const float* gi = g + (it & 7) * 1280;
    if constexpr (KIND == K_DMA16) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const auto* gs = (const __attribute__((address_space(1))) void*)(gi + 1024 * q);
        auto* ld = (__attribute__((address_space(3))) void*)(X + 1024 * q);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 16, 3072, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_DMA4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const auto* gs = (const __attribute__((address_space(1))) void*)(gi - 3 * lane + 256 * q);
        auto* ld = (__attribute__((address_space(3))) void*)(X + 256 * q);
        __builtin_amdgcn_global_load_lds(gs, ld, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 4, 256, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 4, 512, 0);
        __builtin_amdgcn_global_load_lds(gs, ld, 4, 768, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if constexpr (KIND == K_REG16 || KIND == K_REG16_NOLDS) {
      v4 r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const v4*>(gi + 256 * q);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (KIND == K_REG16) *reinterpret_cast<v4*>(X + 256 * q + 4 * lane) = r[q];
        else acc += r[q];
      }
    } else {
      typedef float v2 __attribute__((ext_vector_type(2)));
      v2 r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = *reinterpret_cast<const v2*>(gi - 2 * lane + 128 * q);
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc.x += r[q].x; acc.y += r[q].y; }
    }
  }
  __syncthreads();
  out[blockIdx.x * 768 + t] = X[lane] + acc.x + acc.y + acc.z + acc.w;
}
template <int KIND>
void run(const char* name, const float* s, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<KIND>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 147456, 0, s, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 147456, 0, s, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %8.1f ns per 8 KB span per wave (12 waves/CU)  = %6.2f ns per KB per CU\n", name, ms * 1e6 / iters, ms * 1e6 / iters / (12 * 8.0));
}
int main() {
  float *s, *d;
  (void)hipMalloc(&s, 256 * 65536 * 4 + 65536);
  (void)hipMemset(s, 0, 256 * 65536 * 4 + 65536);
  (void)hipMalloc(&d, 256 * 768 * 4);
  run<K_REG16_NOLDS>("warm-up", s, d);
  run<K_DMA16>("global_load_lds_dwordx4 x 8", s, d);
  run<K_DMA4>("global_load_lds_dword x 32", s, d);
  run<K_REG16>("global_load_dwordx4 x 8 + ds_write_b128 x 8", s, d);
  run<K_REG16_NOLDS>("global_load_dwordx4 x 8 (registers only)", s, d);
  run<K_REG8_NOLDS>("global_load_dwordx2 x 16, 128-byte rows (4 KB)", s, d);
  return 0;
}
