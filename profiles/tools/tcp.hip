// TCP (vector L1) cost of a 64-lane 8-byte load as a function of the number of 128-byte lines it touches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
// each wave walks its own region: per iteration 32 loads; lanes (fi = lane/16, l = lane%16)
template <int MODE>
__global__ void k(const float* in, float* out, int iters, int region_floats) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int fi = lane >> 4, l = lane & 15;
  const float* base = in + (size_t)wave * region_floats;
  v2 acc = {0, 0};
  for (int it = 0; it < iters; ++it) {
    const float* p = base + it * 1280;  // hop 320 * 4 frames per iteration
    v2 v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      int line;
      if (MODE == 0) line = 10 * fi + r;                       // 4 lines per instruction (frames 10 lines apart)
      else if (MODE == 1) line = r + ((r < 10 * fi) ? 32 : 0); // rotated: <= 2 lines per instruction
      else line = r;                                           // 1 line per instruction (no overlap modelled)
      v[r] = *reinterpret_cast<const v2*>(p + 32 * line + 2 * l);
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) acc += v[r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}
template <int MODE>
void run(const char* name, const float* in, float* out) {
  const int iters = 21, region = 21 * 1280 + 4096;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int waves : {4, 8, 12}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, in, out, iters, region);
    (void)hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, in, out, iters, region);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-26s waves/CU %2d: %.1f us per launch (%d wave-iterations/CU)\n", name, waves, ms * 100, waves * iters);
  }
}
int main() {
  const size_t n = (size_t)256 * 12 * (21 * 1280 + 4096);
  float *in, *out; (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, 256 * 12 * 64 * 4);
  (void)hipMemset(in, 0, n * 4);
  run<0>("4 lines/instr (direct)", in, out);
  run<1>("<=2 lines/instr (rotated)", in, out);
  run<2>("1 line/instr", in, out);
  return 0;
}
