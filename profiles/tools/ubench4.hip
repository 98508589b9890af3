// Costs of LDS instruction kinds and their overlap with packed VALU work on one CU (12 wavefronts per CU, all four
// SIMDs busy).  Each wave-iteration issues NL LDS instructions of one kind and NV v_pk_fma_f32, either with the wait
// for the LDS data before the VALU burst (DEP = 1: overlap only between wavefronts) or after it (DEP = 0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

enum { K_NONE, K_R128, K_R64, K_R2X64, K_W32, K_W2X32, K_W128, K_R32, K_W64, K_R2X64S, K_R64S, K_R2ADJ };

template <int KIND>
__device__ __forceinline__ void lds8(unsigned lp, v4& a0, v4& a1, v4& a2, v4& a3, v4& a4, v4& a5, v4& a6, v4& a7) {
  if constexpr (KIND == K_R128)
    asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                 "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lp));
  if constexpr (KIND == K_R64)
    asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:1024\n ds_read_b64 %2, %8 offset:2048\n ds_read_b64 %3, %8 offset:3072\n"
                 "ds_read_b64 %4, %8 offset:4096\n ds_read_b64 %5, %8 offset:5120\n ds_read_b64 %6, %8 offset:6144\n ds_read_b64 %7, %8 offset:7168\n"
                 : "=v"(*(v2*)&a0), "=v"(*(v2*)&a1), "=v"(*(v2*)&a2), "=v"(*(v2*)&a3), "=v"(*(v2*)&a4), "=v"(*(v2*)&a5), "=v"(*(v2*)&a6), "=v"(*(v2*)&a7) : "v"(lp));
  if constexpr (KIND == K_R32)
    asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                 "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n"
                 : "=v"(a0.x), "=v"(a1.x), "=v"(a2.x), "=v"(a3.x), "=v"(a4.x), "=v"(a5.x), "=v"(a6.x), "=v"(a7.x) : "v"(lp));
  if constexpr (KIND == K_R2X64)
    asm volatile("ds_read2_b64 %0, %8 offset1:16\n ds_read2_b64 %1, %8 offset0:32 offset1:48\n ds_read2_b64 %2, %8 offset0:64 offset1:80\n ds_read2_b64 %3, %8 offset0:96 offset1:112\n"
                 "ds_read2_b64 %4, %8 offset0:128 offset1:144\n ds_read2_b64 %5, %8 offset0:160 offset1:176\n ds_read2_b64 %6, %8 offset0:192 offset1:208\n ds_read2_b64 %7, %8 offset0:224 offset1:240\n"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lp));
  if constexpr (KIND == K_W32)
    asm volatile("ds_write_b32 %8, %0\n ds_write_b32 %8, %1 offset:1024\n ds_write_b32 %8, %2 offset:2048\n ds_write_b32 %8, %3 offset:3072\n"
                 "ds_write_b32 %8, %4 offset:4096\n ds_write_b32 %8, %5 offset:5120\n ds_write_b32 %8, %6 offset:6144\n ds_write_b32 %8, %7 offset:7168\n"
                 :: "v"(a0.x), "v"(a1.x), "v"(a2.x), "v"(a3.x), "v"(a4.x), "v"(a5.x), "v"(a6.x), "v"(a7.x), "v"(lp) : "memory");
  if constexpr (KIND == K_W2X32)
    asm volatile("ds_write2_b32 %8, %0, %1 offset1:32\n ds_write2_b32 %8, %1, %2 offset0:64 offset1:96\n ds_write2_b32 %8, %2, %3 offset0:128 offset1:160\n ds_write2_b32 %8, %3, %4 offset0:192 offset1:224\n"
                 "ds_write2_b32 %8, %4, %5 offset0:16 offset1:48\n ds_write2_b32 %8, %5, %6 offset0:80 offset1:112\n ds_write2_b32 %8, %6, %7 offset0:144 offset1:176\n ds_write2_b32 %8, %7, %0 offset0:208 offset1:240\n"
                 :: "v"(a0.x), "v"(a1.x), "v"(a2.x), "v"(a3.x), "v"(a4.x), "v"(a5.x), "v"(a6.x), "v"(a7.x), "v"(lp) : "memory");
  if constexpr (KIND == K_W64)
    asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:1024\n ds_write_b64 %8, %2 offset:2048\n ds_write_b64 %8, %3 offset:3072\n"
                 "ds_write_b64 %8, %4 offset:4096\n ds_write_b64 %8, %5 offset:5120\n ds_write_b64 %8, %6 offset:6144\n ds_write_b64 %8, %7 offset:7168\n"
                 :: "v"(*(v2*)&a0), "v"(*(v2*)&a1), "v"(*(v2*)&a2), "v"(*(v2*)&a3), "v"(*(v2*)&a4), "v"(*(v2*)&a5), "v"(*(v2*)&a6), "v"(*(v2*)&a7), "v"(lp) : "memory");
  // the front-end's sample reads: rows of 128 bytes (n1), 16 lanes 8 bytes apart inside a row, frames 1280 bytes apart
  if constexpr (KIND == K_R2X64S)
    asm volatile("ds_read2_b64 %0, %8 offset1:16\n ds_read2_b64 %1, %8 offset0:32 offset1:48\n ds_read2_b64 %2, %8 offset0:64 offset1:80\n ds_read2_b64 %3, %8 offset0:96 offset1:112\n"
                 "ds_read2_b64 %4, %8 offset0:128 offset1:144\n ds_read2_b64 %5, %8 offset0:160 offset1:176\n ds_read2_b64 %6, %8 offset0:192 offset1:208\n ds_read2_b64 %7, %8 offset0:224 offset1:240\n"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lp));
  if constexpr (KIND == K_R64S)
    asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:128\n ds_read_b64 %2, %8 offset:256\n ds_read_b64 %3, %8 offset:384\n"
                 "ds_read_b64 %4, %8 offset:512\n ds_read_b64 %5, %8 offset:640\n ds_read_b64 %6, %8 offset:768\n ds_read_b64 %7, %8 offset:896\n"
                 : "=v"(*(v2*)&a0), "=v"(*(v2*)&a1), "=v"(*(v2*)&a2), "=v"(*(v2*)&a3), "=v"(*(v2*)&a4), "=v"(*(v2*)&a5), "=v"(*(v2*)&a6), "=v"(*(v2*)&a7) : "v"(lp));
  // read2 of two adjacent 8-byte values (= one 16-byte read)
  if constexpr (KIND == K_R2ADJ)
    asm volatile("ds_read2_b64 %0, %8 offset1:1\n ds_read2_b64 %1, %8 offset0:128 offset1:129\n ds_read2_b64 %2, %8 offset0:2 offset1:3\n ds_read2_b64 %3, %8 offset0:130 offset1:131\n"
                 "ds_read2_b64 %4, %8 offset0:4 offset1:5\n ds_read2_b64 %5, %8 offset0:132 offset1:133\n ds_read2_b64 %6, %8 offset0:6 offset1:7\n ds_read2_b64 %7, %8 offset0:134 offset1:135\n"
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lp));
  if constexpr (KIND == K_W128)
    asm volatile("ds_write_b128 %8, %0\n ds_write_b128 %8, %1 offset:1024\n ds_write_b128 %8, %2 offset:2048\n ds_write_b128 %8, %3 offset:3072\n"
                 "ds_write_b128 %8, %4 offset:4096\n ds_write_b128 %8, %5 offset:5120\n ds_write_b128 %8, %6 offset:6144\n ds_write_b128 %8, %7 offset:7168\n"
                 :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(lp) : "memory");
}

// NL8 groups of 8 LDS instructions, NV8 groups of 8 packed fmas per wave-iteration
template <int KIND, int NL8, int NV8, int DEP, int PITCH = 1280>
__global__ void __launch_bounds__(768) k(float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x;
  for (int i = t; i < 36864; i += blockDim.x) lds[i] = i;
  __syncthreads();
  v4 a0 = {1.f + t, 2.f, 3.f, 4.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  v2 f0 = {(float)t, 1.f}, f1 = f0, f2 = f0, f3 = f0, f4 = f0, f5 = f0, f6 = f0, f7 = f0;
  const v2 c = {0.999f, 1.001f};
  constexpr int W = KIND == K_R128 || KIND == K_W128 ? 16 : KIND == K_R64 || KIND == K_W64 ? 8 : KIND == K_R2X64 ? 8 : 4;
  // conflict-free: consecutive lanes W bytes apart; each wave its own 8 KB + 4 KB region
  unsigned lp = (unsigned)(size_t)(lds) + (t & 63) * W + (t >> 6) * 12288;
  if (KIND == K_R2X64S || KIND == K_R64S) lp = (unsigned)(size_t)(lds) + (t & 15) * 8 + ((t >> 4) & 3) * PITCH + (t >> 6) * 12288;
  if (KIND == K_R2ADJ) lp = (unsigned)(size_t)(lds) + (t & 63) * 16 + (t >> 6) * 12288;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < NL8; ++g) lds8<KIND>(lp + (g & 3) * (KIND == K_R2X64 ? 2048 : 0), a0, a1, a2, a3, a4, a5, a6, a7);
    if (DEP && NL8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < NV8; ++g)
      asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c));
    if (!DEP && NL8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + t] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + f0.x + f1.x + f2.x + f3.x + f4.y + f5.y + f6.y + f7.y;
}

template <int KIND, int NL8, int NV8, int DEP, int PITCH = 1280>
void run(const char* name, float* d, int waves) {
  const int iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<KIND, NL8, NV8, DEP, PITCH>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 147456, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 147456, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_it_cu = ms * 1e6 / iters;  // ns per iteration of all the CU's waves
  printf("%-44s %2d waves/CU  %8.1f ns per round of the CU", name, waves, per_it_cu);
  if (NL8) printf("  %6.2f ns per LDS instr per CU", per_it_cu / (waves * NL8 * 8.0));
  if (NV8) printf("  %6.2f ns per pk_fma per SIMD", per_it_cu / (waves / 4.0 * NV8 * 8.0));
  printf("\n");
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  run<K_NONE, 0, 16, 0>("warm-up", d, 12);
  run<K_NONE, 0, 16, 0>("warm-up", d, 12);
  for (int w : {12}) {
    run<K_NONE, 0, 16, 0>("128 pk_fma", d, w);
    run<K_R128, 4, 0, 1>("32 ds_read_b128", d, w);
    run<K_R64, 4, 0, 1>("32 ds_read_b64", d, w);
    run<K_R2X64, 4, 0, 1>("32 ds_read2_b64 (128 B apart)", d, w);
    run<K_R2ADJ, 4, 0, 1>("32 ds_read2_b64 (adjacent = 16 B)", d, w);
    run<K_R2X64S, 4, 0, 1, 1280>("32 ds_read2_b64 sample pattern, hop 320", d, w);
    run<K_R64S, 4, 0, 1, 1280>("32 ds_read_b64 sample pattern, hop 320", d, w);
    run<K_R2X64S, 4, 0, 1, 1408>("32 ds_read2_b64 sample pattern, +32 banks", d, w);
    run<K_R64S, 4, 0, 1, 1408>("32 ds_read_b64 sample pattern, +32 banks", d, w);
    run<K_R2X64S, 4, 0, 1, 1344>("32 ds_read2_b64 sample pattern, +16 banks", d, w);
    run<K_R64S, 4, 0, 1, 1344>("32 ds_read_b64 sample pattern, +16 banks", d, w);
    run<K_W32, 4, 0, 1>("32 ds_write_b32", d, w);
    run<K_W64, 4, 0, 1>("32 ds_write_b64", d, w);
    run<K_W128, 4, 0, 1>("32 ds_write_b128", d, w);
  }
  return 0;
}
