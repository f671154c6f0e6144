// Probe (round 6): what does a device-wide barrier between the phases of a persistent kernel cost on MI355X?  N workgroups of 256
// lanes, K barriers, a little dependent work in between (every workgroup writes a slot, reads its neighbour's after the barrier - a
// wrong value means the barrier or the fences do not do what they must).  A spin limit makes a lost participant an error, not a hang.
// build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe scripts/probes/grid_barrier_probe.hip ; run: ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Bar { unsigned int count; unsigned int gen; unsigned int error; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned int nwg, unsigned int& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();                                             // release: this workgroup's writes are visible device-wide
    const unsigned int g = my_gen;
    const unsigned int arrived = atomicAdd(&b->count, 1u) + 1u;
    if (arrived == nwg) {
      b->count = 0u;
      __threadfence();
      atomicAdd(&b->gen, 1u);                                    // open the barrier
    } else {
      unsigned int spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000u) { atomicExch(&b->error, 1u); ok = false; break; }
      }
    }
    __threadfence();                                             // acquire
  }
  ++my_gen;
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void probe(Bar* b, double* slots, int K, int* bad) {
  unsigned int gen = 0;
  const unsigned int nwg = gridDim.x;
  const int w = blockIdx.x;
  for (int k = 0; k < K; ++k) {
    if (threadIdx.x == 0) slots[(size_t)(k & 1) * nwg + w] = (double)(k * 1000 + w);
    if (!grid_barrier(b, nwg, gen)) return;
    if (threadIdx.x == 0) {
      const int nb = (w + 1) % (int)nwg;
      const double v = __hip_atomic_load(&slots[(size_t)(k & 1) * nwg + nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (double)(k * 1000 + nb)) atomicAdd(bad, 1);
    }
  }
}
__global__ void empty_kernel() {}

int main() {
  Bar* b; double* slots; int* bad;
  CHECK(hipMalloc(&b, sizeof(Bar))); CHECK(hipMalloc(&slots, 2 * 1024 * sizeof(double))); CHECK(hipMalloc(&bad, sizeof(int)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int K = 200;
  for (int nwg : {8, 16, 32, 64, 128, 256}) {
    CHECK(hipMemset(b, 0, sizeof(Bar))); CHECK(hipMemset(bad, 0, sizeof(int)));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, b, slots, 4, bad);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(b, 0, sizeof(Bar)));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, b, slots, K, bad);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; int hbad; CHECK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
    printf("workgroups %3d: %d barriers in %.1f us = %.2f us per barrier, wrong neighbour values %d, barrier error %u\n", nwg, K, ms * 1e3, ms * 1e3 / K, hbad, hb.error);
  }
  // for comparison: the cost of a kernel boundary (back-to-back empty launches in one stream)
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, 0);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("200 empty launches: %.2f us each\n", ms * 1e3 / 200);
  return 0;
}
