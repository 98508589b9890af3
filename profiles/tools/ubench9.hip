// Cost of agent-scope atomics on per-workgroup words in global memory (the owner's pops and the thieves' steals of a
// cross-CU work queue): 256 workgroups x 12 wavefronts, lane 0 of every wavefront issues K dependent
// atomicAdd(relaxed, agent) on (a) its workgroup's word, one 128-byte line per workgroup; (b) a word of its own;
// (c) the word of another workgroup (blockIdx + 37, another XCD); (d) one word for the whole device.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(768) k(unsigned long long* w, unsigned long long* out, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long* p;
  if (MODE == 0) p = w + 16 * blockIdx.x;
  else if (MODE == 1) p = w + 16 * (blockIdx.x * 12 + wave);
  else if (MODE == 2) p = w + 16 * ((blockIdx.x + 37) % gridDim.x);
  else p = w;
  unsigned long long acc = 0;
  if (lane == 0)
    for (int i = 0; i < K; ++i) acc += __hip_atomic_fetch_add(p, 1ull + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) out[blockIdx.x * 12 + wave] = acc;
}
template <int MODE>
void run(unsigned long long* w, unsigned long long* out, int K, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, w, out, K);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, w, out, K);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s K=%4d: %8.1f us total, %7.1f ns per dependent atomic of a wavefront, %6.1f atomics/us device-wide\n", name, K, ms * 1e3,
         ms * 1e6 / K, 3072.0 * K / (ms * 1e3));
}
int main() {
  unsigned long long *w, *out;
  (void)hipMalloc(&w, 16 * 8 * 4096); (void)hipMemset(w, 0, 16 * 8 * 4096);
  (void)hipMalloc(&out, 8 * 4096);
  run<0>(w, out, 100, "word per workgroup (12 wavefronts share it)");
  run<1>(w, out, 100, "word per wavefront");
  run<2>(w, out, 100, "word of another workgroup (other XCD)");
  run<3>(w, out, 20, "one word for the device");
  return 0;
}
