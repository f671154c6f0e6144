// Probe: cost of a grid-wide barrier (agent-scope atomics + fences) between dependent phases of one persistent launch,
// against the same phases as separate launches of a captured graph.  hipcc --offload-arch=gfx950 -O3 gridbar_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 200000000LL) { *err = 1; ok = false; break; }   // 2 s at 100 MHz: give up, never hang
    }
  }
  __syncthreads();
  return ok;
}

// phase p: y[i] = 0.5 * (x[i-1] + x[i+1]) ping-pong (needs the neighbours' values of the previous phase)
__global__ __launch_bounds__(256) void persistent(double* a, double* b, int n, int phases, unsigned* counter, unsigned* err) {
  const int nth = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  double* x = a;
  double* y = b;
  for (int p = 0; p < phases; ++p) {
    for (int i = tid; i < n; i += nth) y[i] = 0.5 * (x[i > 0 ? i - 1 : i] + x[i < n - 1 ? i + 1 : i]);
    if (!grid_barrier(counter, (unsigned)(p + 1) * gridDim.x, err)) return;
    double* t = x; x = y; y = t;
  }
}
__global__ __launch_bounds__(256) void one_phase(const double* x, double* y, int n) {
  const int nth = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = tid; i < n; i += nth) y[i] = 0.5 * (x[i > 0 ? i - 1 : i] + x[i < n - 1 ? i + 1 : i]);
}

int main() {
  const int phases = 16, reps = 50;
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int n : {4913, 35937, 274625}) {
    double *a, *b, *a2, *b2;
    unsigned *cnt, *err;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&a2, n * 8)); CK(hipMalloc(&b2, n * 8));
    CK(hipMalloc(&cnt, 4 * (reps + 4))); CK(hipMalloc(&err, 4));
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = (i * 37 % 101) / 101.0;
    for (int wg : {16, 32, 64, 128, 256, 512}) {
      CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
      CK(hipMemcpy(a2, h.data(), n * 8, hipMemcpyHostToDevice));
      CK(hipMemset(cnt, 0, 4 * (reps + 4))); CK(hipMemset(err, 0, 4));
      // persistent: one launch per rep (a fresh counter each)
      hipLaunchKernelGGL(persistent, dim3(wg), dim3(256), 0, s, a, b, n, phases, cnt + reps + 1, err);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(persistent, dim3(wg), dim3(256), 0, s, a, b, n, phases, cnt + r, err);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms_p = 0;
      CK(hipEventElapsedTime(&ms_p, e0, e1));
      unsigned herr = 0;
      CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      // graph of separate launches
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_phase, dim3(wg), dim3(256), 0, s, (p & 1) ? b2 : a2, (p & 1) ? a2 : b2, n);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(a2, h.data(), n * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms_g = 0;
      CK(hipEventElapsedTime(&ms_g, e0, e1));
      std::vector<double> r1(n), r2(n);
      CK(hipMemcpy(r1.data(), a, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r2.data(), a2, n * 8, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < n; ++i) bad += r1[i] != r2[i];
      std::printf("n %7d  wg %4d  persistent %.2f us/phase  graph %.2f us/phase  mismatches %d  timeout %u\n", n, wg,
                  ms_p * 1000.0 / (reps * phases), ms_g * 1000.0 / (reps * phases), bad, herr);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(a2)); CK(hipFree(b2)); CK(hipFree(cnt)); CK(hipFree(err));
  }
  return 0;
}
