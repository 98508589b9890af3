// VALU issue cost vs operand mix on gfx950 (3 waves per SIMD, independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters) {
  const int t = threadIdx.x;
  v2 a0 = {1.f + t, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  v2 b0 = {0.5f, 0.25f}, b1 = {0.75f, 0.125f}, c0 = {0.999f, 1.001f}, c1 = {1.0f, 0.99f};
  float f0 = t, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7, g0 = 0.5f, g1 = 0.25f, h0 = 0.999f, h1 = 1.001f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {  // pk_fma, 3 distinct VGPR pairs
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %10, %11\n v_pk_fma_f32 %2, %2, %8, %11\n v_pk_fma_f32 %3, %3, %10, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %10, %11\n v_pk_fma_f32 %6, %6, %8, %11\n v_pk_fma_f32 %7, %7, %10, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0), "v"(b1), "v"(c1));)
    } else if constexpr (KIND == 1) {  // pk_fma, dst != src (as in butterflies): d = a*b + c all distinct
      REP8(asm volatile("v_pk_fma_f32 %0, %1, %8, %9\n v_pk_fma_f32 %1, %2, %10, %11\n v_pk_fma_f32 %2, %3, %8, %11\n v_pk_fma_f32 %3, %4, %10, %9\n"
                   "v_pk_fma_f32 %4, %5, %8, %9\n v_pk_fma_f32 %5, %6, %10, %11\n v_pk_fma_f32 %6, %7, %8, %11\n v_pk_fma_f32 %7, %0, %10, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0), "v"(b1), "v"(c1));)
    } else if constexpr (KIND == 2) {  // pk_add, 2 distinct pairs
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
    } else if constexpr (KIND == 3) {  // pk_fma with SGPR pair as src1
      REP8(asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %8\n v_pk_fma_f32 %1, %1, s[20:21], %9\n v_pk_fma_f32 %2, %2, s[20:21], %8\n v_pk_fma_f32 %3, %3, s[20:21], %9\n"
                   "v_pk_fma_f32 %4, %4, s[20:21], %8\n v_pk_fma_f32 %5, %5, s[20:21], %9\n v_pk_fma_f32 %6, %6, s[20:21], %8\n v_pk_fma_f32 %7, %7, s[20:21], %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "s20", "s21");)
    } else if constexpr (KIND == 4) {  // v_fma_f32 3 distinct VGPRs
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %8, %11\n v_fma_f32 %3, %3, %10, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %8, %11\n v_fma_f32 %7, %7, %10, %9\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g0), "v"(h0), "v"(g1), "v"(h1));)
    } else if constexpr (KIND == 5) {  // pk_fma with op_sel swizzles and neg (as cmul tail)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %10, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %2, %2, %8, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %10, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                   "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %10, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %6, %6, %8, %11 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %10, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0), "v"(b1), "v"(c1));)
    } else if constexpr (KIND == 6) {  // v_pk_mul_f32 2 distinct
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %9\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %9\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c0), "v"(c1));)
    } else if constexpr (KIND == 7) {  // v_add_f32 2 distinct
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g0), "v"(g1));)
    }
  }
  out[blockIdx.x * blockDim.x + t] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int KIND>
void run(const char* name, float* d) {
  const int iters = 4000, waves = 3;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves), 0, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ns per wave-instr per SIMD\n", name, ms * 1e6 / ((double)iters * 64 * waves));
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
  run<7>("warm-up (v_add_f32)", d);
  run<7>("v_add_f32, 2 distinct VGPR", d);
  run<4>("v_fma_f32, 3 distinct VGPR", d);
  run<2>("v_pk_add_f32, 2 distinct pairs", d);
  run<6>("v_pk_mul_f32, 2 distinct pairs", d);
  run<0>("v_pk_fma_f32, 3 distinct pairs (dst=src0)", d);
  run<1>("v_pk_fma_f32, 3 distinct pairs (dst!=src)", d);
  run<3>("v_pk_fma_f32, SGPR pair src1", d);
  run<5>("v_pk_fma_f32, op_sel/neg modifiers", d);
  return 0;
}
