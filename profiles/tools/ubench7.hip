// Do a VALU phase and an LDS phase of different wavefronts of a SIMD overlap?  Every wavefront loops over
// {V: NV packed FMAs on four independent chains; L: NR ds_read_b128 + NW ds_write_b64, one lgkmcnt(0) at the end},
// all LDS instructions hand-written so that the compiler adds no arithmetic and no waits of its own.
// 12 wavefronts per CU (3 per SIMD).  Times per round: V alone, L alone, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int NV, int NR, int NW, bool STAGGER>
__global__ void __launch_bounds__(768) k(float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* Y = lds + wave * 2048;
  const unsigned ya = (unsigned)(size_t)(Y + 4 * lane);
  v2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f}, e = {2.0f, 5.0f}, f = {7.0f, 1.5f};
  v4 r0, r1, r2, r3;
  r0 = r1 = r2 = r3 = v4{0, 0, 0, 0};
  if (STAGGER) {  // wavefronts of one SIMD start a third of a round apart
    const int slot = wave >> 2;  // waves w, w+4, w+8 share a SIMD
    for (int i = 0; i < slot * (NV / 3); ++i) a = __builtin_elementwise_fma(a, b, c);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
      a = __builtin_elementwise_fma(a, b, c);
      d = __builtin_elementwise_fma(d, b, c);
      e = __builtin_elementwise_fma(e, c, b);
      f = __builtin_elementwise_fma(f, c, b);
    }
    asm volatile("" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));
#pragma unroll
    for (int i = 0; i < NR / 4; ++i) {
      asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                   : "v"(ya), "n"(1024 * 0 + 64 * (i % 16)), "n"(1024 * 1 + 64 * (i % 16)), "n"(1024 * 2 + 64 * (i % 16)), "n"(1024 * 3 + 64 * (i % 16))
                   : "memory");
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ya), "v"(a), "n"(4096 + 8 * (i % 32)) : "memory");
    if (NR + NW > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)::"memory");
  }
  out[blockIdx.x * 768 + t] = a.x + a.y + d.x + d.y + e.x + f.y + r0.x + r1.y + r2.z + r3.w;
}
template <int NV, int NR, int NW, bool STAGGER>
double run(float* d, int waves) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<NV, NR, NW, STAGGER>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 12 * 2048 * 4, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 12 * 2048 * 4, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
template <int NV, int NR, int NW>
void trio(float* d) {
  const double v = run<NV, 0, 0, false>(d, 12), l = run<0, NR, NW, false>(d, 12), both = run<NV, NR, NW, false>(d, 12),
               st = run<NV, NR, NW, true>(d, 12);
  printf("%4d pk_fma | %3d read_b128 + %3d write_b64 :  V %7.1f ns  L %7.1f ns  both %7.1f ns (sum %7.1f, max %7.1f)  staggered start %7.1f ns\n",
         NV, NR, NW, v, l, both, v + l, v > l ? v : l, st);
}

// The same work per round cut into K alternations {NV/K FMAs; NR/K reads + NW/K writes; lgkmcnt(0)}: K wait points per round.
// DEP: the FMAs of a chunk use the values just read (as real code does), so the wait cannot be moved.
template <int NV, int NR, int NW, int K, bool PREFETCH>
__global__ void __launch_bounds__(768) kc(float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* Y = lds + wave * 2048;
  const unsigned ya = (unsigned)(size_t)(Y + 4 * lane);
  v2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f}, e = {2.0f, 5.0f}, f = {7.0f, 1.5f};
  v4 r0, r1, r2, r3;
  r0 = r1 = r2 = r3 = v4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      if (!PREFETCH) {
#pragma unroll
        for (int i = 0; i < NR / K / 4; ++i)
          asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                       : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                       : "v"(ya), "n"(0), "n"(1024), "n"(2048), "n"(3072) : "memory");
#pragma unroll
        for (int i = 0; i < NW / K; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ya), "v"(a), "n"(4096 + 8 * (i % 32)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)::"memory");
        b.x += r0.x * 0.0f;  // the arithmetic depends on what was read
      }
#pragma unroll
      for (int i = 0; i < NV / K / 4; ++i) {
        a = __builtin_elementwise_fma(a, b, c);
        d = __builtin_elementwise_fma(d, b, c);
        e = __builtin_elementwise_fma(e, c, b);
        f = __builtin_elementwise_fma(f, c, b);
      }
      asm volatile("" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));
    }
  }
  out[blockIdx.x * 768 + t] = a.x + a.y + d.x + d.y + e.x + f.y + r0.x + r1.y + r2.z + r3.w;
}
template <int NV, int NR, int NW, int K>
void chunked(float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = kc<NV, NR, NW, K, false>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 2048 * 4, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 2048 * 4, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%4d pk_fma | %3d read_b128 + %3d write_b64 in %2d alternations, FMAs depend on the reads: %7.1f ns per round\n", NV, NR, NW, K, ms * 1e6 / iters);
}

// The balanced loop with span staging added: per round every wavefront also orders NB x 8 KB by LDS-DMA (into its own
// span region, waited for at the top of the next round), from a buffer that streams through HBM / L2.
template <int NV, int NR, int NW, int NB, int SPANS = 32, bool WAIT = true, int MODE = 0>
__global__ void __launch_bounds__(768) kd(const float* src, float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* Y = lds + wave * 3200;      // 1152 floats of work area + 2048 floats of span
  float* X = Y + 1152;
  const unsigned ya = (unsigned)(size_t)(Y + 4 * lane);
  const float* g = src + ((size_t)blockIdx.x * 12 + wave) * 65536 + 4 * lane;
  v2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f}, e = {2.0f, 5.0f}, f = {7.0f, 1.5f};
  v4 r0, r1, r2, r3;
  r0 = r1 = r2 = r3 = v4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (NB > 0 && WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float* gi = g + ((it * NB + nb) & (SPANS - 1)) * 1280;
      v4 sp[8];
      if (MODE == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const auto* gs = (const __attribute__((address_space(1))) void*)(gi + 1024 * q);
          auto* ld = (__attribute__((address_space(3))) void*)(X + 1024 * q);
          __builtin_amdgcn_global_load_lds(gs, ld, 16, 0, 0);
          __builtin_amdgcn_global_load_lds(gs, ld, 16, 1024, 0);
          __builtin_amdgcn_global_load_lds(gs, ld, 16, 2048, 0);
          __builtin_amdgcn_global_load_lds(gs, ld, 16, 3072, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) sp[q] = *reinterpret_cast<const v4*>(gi + 256 * q);
      }
#pragma unroll
      for (int i = 0; i < NV / NB / 4; ++i) {
        a = __builtin_elementwise_fma(a, b, c);
        d = __builtin_elementwise_fma(d, b, c);
        e = __builtin_elementwise_fma(e, c, b);
        f = __builtin_elementwise_fma(f, c, b);
      }
      asm volatile("" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));
#pragma unroll
      for (int i = 0; i < NR / NB / 4; ++i)
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                     : "v"(ya), "n"(0), "n"(1024), "n"(2048), "n"(3072) : "memory");
#pragma unroll
      for (int i = 0; i < NW / NB; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ya), "v"(a), "n"(4096 + 8 * (i % 32)) : "memory");
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<v4*>(X + 256 * q + 4 * lane) = sp[q];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)::"memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 768 + t] = a.x + a.y + d.x + d.y + e.x + f.y + r0.x + r1.y + r2.z + r3.w + X[lane];
}
template <int NV, int NR, int NW, int NB, int SPANS = 32, bool WAIT = true, int MODE = 0>
double rund(const float* s, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = kd<NV, NR, NW, NB, SPANS, WAIT, MODE>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 3200 * 4, 0, s, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 3200 * 4, 0, s, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 768 * 4);
  run<384, 0, 0, false>(d, 12);
  trio<384, 48, 0>(d);
  trio<384, 0, 48>(d);
  trio<384, 48, 24>(d);
  trio<768, 96, 48>(d);
  trio<192, 24, 12>(d);
  trio<96, 12, 6>(d);
  chunked<768, 96, 48, 1>(d);
  chunked<768, 96, 48, 2>(d);
  chunked<768, 96, 48, 4>(d);
  chunked<768, 96, 48, 8>(d);
  chunked<768, 96, 48, 12>(d);
  chunked<768, 96, 48, 24>(d);
  {
    float* src;
    const size_t n = (size_t)256 * 12 * 65536 + 65536;   // 805 MB: streams through HBM
    (void)hipMalloc(&src, n * 4);
    (void)hipMemset(src, 0, n * 4);
    // NB = 3 in both: the same three sub-rounds, with and without the DMA instructions (NB = -3 is not expressible, so
    // the no-DMA case is built by a second kernel with the same loop structure: NB sub-rounds, zero blocks)
    const double with3 = rund<768, 96, 48, 3>(src, d);
    const double with1 = rund<768, 96, 48, 1>(src, d);
    printf(" 768 pk_fma | 96 read_b128 + 48 write_b64 + 3 x 8 KB LDS-DMA per wavefront and round (3 sub-rounds): %7.1f ns per round\n", with3);
    printf("   ... 3 x 8 KB, every wavefront re-reading one span (L2 / L1 resident): %7.1f ns per round\n", rund<768, 96, 48, 3, 1>(src, d));
    printf("   ... 3 x 8 KB from HBM, never waited for (vmcnt only at the end):        %7.1f ns per round\n", rund<768, 96, 48, 3, 32, false>(src, d));
    printf("   ... 3 x 8 KB re-read, never waited for:                                  %7.1f ns per round\n", rund<768, 96, 48, 3, 1, false>(src, d));
    printf("   ... 3 x 8 KB from HBM through registers (8 x global_load_dwordx4 early, 8 x ds_write_b128 late): %7.1f ns per round\n", rund<768, 96, 48, 3, 32, false, 1>(src, d));
    printf("   ... 3 x 8 KB re-read through registers:                                                           %7.1f ns per round\n", rund<768, 96, 48, 3, 1, false, 1>(src, d));
    printf(" 768 pk_fma | 96 read_b128 + 48 write_b64 + 1 x 8 KB LDS-DMA per wavefront and round               : %7.1f ns per round\n", with1);
  }
  return 0;
}
