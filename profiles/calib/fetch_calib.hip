// FETCH_SIZE calibration (MI355X_MICROARCH.md "HBM": calibrate the counter on a known byte count in
// your own access widths).  Streams a 512 MiB buffer once with 16-, 8- and 4-byte-per-lane loads.
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ in, size_t n, double* out) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    T v = in[i];
    const int* w = reinterpret_cast<const int*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += (double)w[k];
  }
  if (acc == 123456789.0) out[0] = acc;  // keep the loads alive
}
int main() {
  const size_t bytes = 512ull << 20;
  void* buf; double* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 8);
  hipMemset(buf, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream_read<double2>, dim3(2048), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out);
    hipLaunchKernelGGL(stream_read<double>, dim3(2048), dim3(256), 0, 0, (const double*)buf, bytes / 8, out);
    hipLaunchKernelGGL(stream_read<int>, dim3(2048), dim3(256), 0, 0, (const int*)buf, bytes / 4, out);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per launch\n", bytes);
  return 0;
}
