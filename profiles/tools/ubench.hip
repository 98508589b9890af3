// Issue-rate microbenchmarks for gfx950: cycles per wave-instruction per SIMD for a few instruction kinds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters) {
  __shared__ __align__(16) float lds[4096];
  const int t = threadIdx.x;
  for (int i = t; i < 4096; i += blockDim.x) lds[i] = i;
  __syncthreads();
  v2 a0 = {1.f + t, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  const v2 c = {0.999f, 1.001f};
  float f0 = t, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
  v4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
  const float* lp = lds + (t & 63) * 4;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {  // v_pk_fma_f32, 8 independent chains
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if constexpr (KIND == 1) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c.x));)
    } else if constexpr (KIND == 2) {  // dependent v_pk_fma chain
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n"
                   "v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n"
                   : "+v"(a0) : "v"(c));)
    } else if constexpr (KIND == 3) {  // dependent v_fma chain
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                   "v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                   : "+v"(f0) : "v"(c.x));)
    } else if constexpr (KIND == 4) {  // ds_read_b64 x8 then wait
      REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"((unsigned)(size_t)(lp) + (t & 63) * 0));)
    } else if constexpr (KIND == 5) {  // ds_read_b128 x4 then wait
      REP8(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)\n"
                   : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"((unsigned)(size_t)(lp)));)
    } else if constexpr (KIND == 6) {  // v_mfma_f32_4x4x1 dependent chain
      REP8(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %1, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %1, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %1, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %1, %0\n"
                   : "+v"(q0) : "v"(f1));)
    } else if constexpr (KIND == 7) {  // s_mov + v_pk_fma with sgpr pair (mix SALU/VALU)
      REP8(asm volatile("s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %0, %0, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %1, %1, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %2, %2, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %3, %3, %8, %8\n"
                   "s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %4, %4, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %5, %5, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %6, %6, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20", "s21");)
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + q0.x + q1.y + q2.z + q3.w;
}
template <int KIND>
void run(const char* name, int per_iter, float* d) {
  for (int waves : {1, 2, 3}) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * per_iter * waves;
    printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cycles)\n", name, waves, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  }
}
int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_pk_fma_f32 indep", 64, d);
  run<1>("v_fma_f32 indep", 64, d);
  run<2>("v_pk_fma_f32 dependent", 64, d);
  run<3>("v_fma_f32 dependent", 64, d);
  run<4>("ds_read_b64 (8 then wait)", 64, d);
  run<5>("ds_read_b128 (4 then wait)", 32, d);
  run<6>("mfma 4x4x1 dependent", 32, d);
  run<7>("s_mov + v_pk_fma pairs", 128, d);
  return 0;
}
