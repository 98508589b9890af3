// What would the stage-1 -> stage-2 transposition cost in registers instead of through LDS?
// Per round every wavefront runs the fused kernel's packed arithmetic (560 v_pk_fma_f32) and its LDS traffic without the
// exchange (72 ds_read_b128, 32 ds_read_b64, 36 ds_write_b32), plus one of:
//   X0 nothing;  X1 the LDS exchange (32 ds_write_b64 + 32 ds_read_b64);
//   X2 a 16 x 16 transposition of 64 registers among the lanes of a frame: 2 steps over the row bits of the lane number
//      (v_permlane32_swap / v_permlane16_swap, one instruction per register pair = 64) and 2 steps inside rows
//      (v_cndmask_b32 with a DPP source, one per register and step = 128).
// 12 wavefronts per CU.  Timing only: the values are meaningless.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int MODE, bool VALU, bool LDS>
__global__ void __launch_bounds__(768) k(float* out, int iters) {
  extern __shared__ __align__(16) float lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* Y = lds + wave * 2816;
  const unsigned ya = (unsigned)(size_t)(Y + 4 * lane);
  const unsigned yb = (unsigned)(size_t)(Y + 2 * lane);
  v2 a = {1.0f + lane, 2.0f}, b = {0.5f, 0.25f}, c = {1e-3f, 2e-3f}, d = {3.0f, 1.0f}, e = {2.0f, 5.0f}, f = {7.0f, 1.5f};
  v4 r0, r1, r2, r3;
  v2 s0 = {0, 0}, s1 = {0, 0};
  r0 = r1 = r2 = r3 = v4{0, 0, 0, 0};
  unsigned x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = lane * 17 + i;
  const unsigned long long oddmask = 0xaaaaaaaaaaaaaaaaull;
  for (int it = 0; it < iters; ++it) {
    if (VALU) {
#pragma unroll
      for (int i = 0; i < 560 / 4; ++i) {
        a = __builtin_elementwise_fma(a, b, c);
        d = __builtin_elementwise_fma(d, b, c);
        e = __builtin_elementwise_fma(e, c, b);
        f = __builtin_elementwise_fma(f, c, b);
      }
      asm volatile("" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));
    }
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 72 / 4; ++i)
        asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
                     : "v"(ya), "n"(1024 * 0 + 64 * (i % 16)), "n"(1024 * 1 + 64 * (i % 16)), "n"(1024 * 2 + 64 * (i % 16)), "n"(1024 * 3 + 64 * (i % 16))
                     : "memory");
#pragma unroll
      for (int i = 0; i < 32 / 2; ++i)
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(s0), "=v"(s1) : "v"(yb), "n"(512 * (i % 8)), "n"(512 * (i % 8) + 4096) : "memory");
#pragma unroll
      for (int i = 0; i < 36; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(yb), "v"(a.x), "n"(8192 + 4 * (i % 32)) : "memory");
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(yb), "v"(a), "n"(512 * (i % 16)) : "memory");
#pragma unroll
      for (int i = 0; i < 32 / 2; ++i)
        asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(s0), "=v"(s1) : "v"(yb), "n"(512 * (i % 8)), "n"(512 * (i % 8) + 4096) : "memory");
    }
    if (MODE == 2) {
      asm volatile("s_mov_b64 vcc, %0" ::"s"(oddmask) : "vcc");  // (timing only: nothing else touches vcc in the block)
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {  // 4 sets of 16 registers (re / im of two phases)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[i + 8]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x[(i & 3) + 8 * (i >> 2)]), "+v"(x[(i & 3) + 8 * (i >> 2) + 4]));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = (i & 1) + 4 * (i >> 1), q = p + 2;
          unsigned np, nq;
          asm volatile("v_cndmask_b32_dpp %0, %2, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cndmask_b32_dpp %1, %3, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                       : "=&v"(np), "=&v"(nq) : "v"(x[q]), "v"(x[p]));
          x[p] = np; x[q] = nq;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = 2 * i, q = p + 1;
          unsigned np, nq;
          asm volatile("v_cndmask_b32_dpp %0, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cndmask_b32_dpp %1, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                       : "=&v"(np), "=&v"(nq) : "v"(x[q]), "v"(x[p]));
          x[p] = np; x[q] = nq;
        }
      }
    }
    if (LDS || MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(s0), "+v"(s1)::"memory");
  }
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += x[i];
  out[blockIdx.x * 768 + t] = a.x + a.y + d.x + d.y + e.x + f.y + r0.x + r1.y + r2.z + r3.w + s0.x + s1.y + (float)acc;
}
template <int MODE, bool VALU, bool LDS>
double run(float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto kern = k<MODE, VALU, LDS>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 2816 * 4, 0, d, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(768), 12 * 2816 * 4, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / iters;
}
#include <cstring>
#include <cstdlib>
int main(int argc, char** argv) {
  float* d;
  (void)hipMalloc(&d, 256 * 768 * 4);
  if (argc > 2) {  // ubench8 v|l|x1 seconds: one configuration back to back, to sample the clock meanwhile
    const double secs = atof(argv[2]);
    double total = 0, last = 0;
    while (total < secs * 1e9) {
      last = !strcmp(argv[1], "v") ? run<0, true, false>(d) : !strcmp(argv[1], "l") ? run<1, false, true>(d) : run<1, true, true>(d);
      total += last * 2000 * 2;
    }
    printf("%s: %.1f ns per round\n", argv[1], last);
    return 0;
  }
  run<0, true, true>(d);
  printf("per round of 12 wavefronts on a CU, ns:\n");
  printf("arithmetic alone (560 pk_fma)                    %7.1f\n", run<0, true, false>(d));
  printf("LDS traffic alone (no exchange)                  %7.1f\n", run<0, false, true>(d));
  printf("LDS exchange alone (32 write_b64 + 32 read_b64)  %7.1f\n", run<1, false, false>(d));
  printf("register exchange alone (64 permlane swaps + 128 cndmask dpp) %7.1f\n", run<2, false, false>(d));
  printf("X0 arithmetic + LDS traffic                      %7.1f\n", run<0, true, true>(d));
  printf("X1 ... + LDS exchange                            %7.1f\n", run<1, true, true>(d));
  printf("X2 ... + register exchange                       %7.1f\n", run<2, true, true>(d));
  return 0;
}
