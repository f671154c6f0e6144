// What a streaming kernel with the fine-level sweep's operand mix can reach on this chip: the ceiling the z-marching
// row-class kernel (2 B class id + 8 B b + 8 B x read, 8 B written per row = 26 B/row, N = 257^3 rows) is measured
// against, next to the plain copy figure of MI355X_MICROARCH.md (6.29 TB/s).
// build: hipcc -O3 --offload-arch=gfx950 -o triad_calib triad_calib.hip ; run: ./triad_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2_t __attribute__((ext_vector_type(2)));

// one element (pair) per thread, no loop: the shape of a row-parallel launch
__global__ __launch_bounds__(256) void copy16(const d2_t* __restrict__ x, d2_t* __restrict__ y, size_t n2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) y[i] = x[i];
}
__global__ __launch_bounds__(256) void triad16(const d2_t* __restrict__ b, const d2_t* __restrict__ x, d2_t* __restrict__ y,
                                               size_t n2, double s) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) y[i] = b[i] + s * x[i];
}
__global__ __launch_bounds__(256) void triad8(const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ y,
                                              size_t n, double s) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = b[i] + s * x[i];
}
// + the 2-byte class-id stream
__global__ __launch_bounds__(256) void triad8c(const double* __restrict__ b, const double* __restrict__ x,
                                               const unsigned short* __restrict__ c, double* __restrict__ y, size_t n, double s) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = b[i] + (s + (double)c[i]) * x[i];
}
__global__ __launch_bounds__(256) void triad16c(const d2_t* __restrict__ b, const d2_t* __restrict__ x,
                                                const unsigned int* __restrict__ c, d2_t* __restrict__ y, size_t n2, double s) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) {
    const unsigned int cc = c[i];
    d2_t o = b[i];
    const d2_t xv = x[i];
    o.x += (s + (double)(cc & 0xFFFFu)) * xv.x;
    o.y += (s + (double)(cc >> 16)) * xv.y;
    y[i] = o;
  }
}
// persistent form: 512 workgroups of 1024 threads, each walking a contiguous range plane by plane (the march's shape)
__global__ __launch_bounds__(1024) void triad8c_persist(const double* __restrict__ b, const double* __restrict__ x,
                                                        const unsigned short* __restrict__ c, double* __restrict__ y,
                                                        size_t n, double s) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) y[i] = b[i] + (s + (double)c[i]) * x[i];
}

// persistent, interleaved: workgroup w takes the 1024-row blocks w, w + G, w + 2G, ... (the chip sweeps one compact window)
__global__ __launch_bounds__(1024) void triad8c_persist_il(const double* __restrict__ b, const double* __restrict__ x,
                                                           const unsigned short* __restrict__ c, double* __restrict__ y,
                                                           size_t n, double s) {
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) y[i] = b[i] + (s + (double)c[i]) * x[i];
}
// persistent, the march's item order: P rows per plane in chunks of 1024, a workgroup walks consecutive planes of ONE chunk
// (equal contiguous ranges of the chunk-major (chunk, plane) list)
__global__ __launch_bounds__(1024) void triad8c_march(const double* __restrict__ b, const double* __restrict__ x,
                                                      const unsigned short* __restrict__ c, double* __restrict__ y,
                                                      int P, int nplanes, double s, int planemajor) {
  const int chunks = (P + 1023) / 1024;
  const long long tot = (long long)chunks * nplanes;
  long long it = tot * blockIdx.x / gridDim.x;
  const long long it_end = tot * (blockIdx.x + 1) / gridDim.x;
  for (; it < it_end; ++it) {
    const int ch = planemajor ? (int)(it % chunks) : (int)(it / nplanes), z = planemajor ? (int)(it / chunks) : (int)(it % nplanes);
    const int ip = ch * 1024 + threadIdx.x;
    if (ip < P) {
      const size_t i = (size_t)z * P + ip;
      y[i] = b[i] + (s + (double)c[i]) * x[i];
    }
  }
}
// the same with the loads of the next plane issued before the current plane is stored (one plane of prefetch)
__global__ __launch_bounds__(1024) void triad8c_march_pf(const double* __restrict__ b, const double* __restrict__ x,
                                                         const unsigned short* __restrict__ c, double* __restrict__ y,
                                                         int P, int nplanes, double s) {
  const int chunks = (P + 1023) / 1024;
  const long long tot = (long long)chunks * nplanes;
  long long it = tot * blockIdx.x / gridDim.x;
  const long long it_end = tot * (blockIdx.x + 1) / gridDim.x;
  auto idx = [&](long long t) {
    const int ch = (int)(t / nplanes), z = (int)(t % nplanes);
    const int ip = ch * 1024 + threadIdx.x;
    return ip < P ? (long long)z * P + ip : -1LL;
  };
  long long i0 = idx(it);
  double vb = 0, vx = 0; unsigned short vc = 0;
  if (i0 >= 0) { vb = b[i0]; vx = x[i0]; vc = c[i0]; }
  for (; it < it_end; ++it) {
    const long long i1 = it + 1 < it_end ? idx(it + 1) : -1;
    double nb = 0, nx = 0; unsigned short nc = 0;
    if (i1 >= 0) { nb = b[i1]; nx = x[i1]; nc = c[i1]; }
    if (i0 >= 0) y[i0] = vb + (s + (double)vc) * vx;
    i0 = i1; vb = nb; vx = nx; vc = nc;
  }
}
__global__ __launch_bounds__(256) void read2(const d2_t* __restrict__ b, const d2_t* __restrict__ x, double* out, size_t n2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) {
    const d2_t v = b[i] + x[i];
    if (v.x + v.y == 123456.789) out[0] = v.x;
  }
}
__global__ __launch_bounds__(256) void write1(d2_t* __restrict__ y, size_t n2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) y[i] = d2_t{1.0, 2.0};
}

int main() {
  const size_t n = 16974593, n2 = n / 2;
  double *b, *x, *y;
  unsigned short* c;
  hipMalloc(&b, (n + 2) * 8); hipMalloc(&x, (n + 2) * 8); hipMalloc(&y, (n + 2) * 8); hipMalloc(&c, (n + 2) * 2);
  hipMemset(b, 0, n * 8); hipMemset(x, 0, n * 8); hipMemset(y, 0, n * 8); hipMemset(c, 0, n * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-28s %8.1f us  %7.1f MB  %6.2f TB/s  frac of 8 TB/s %.3f\n", name, us, bytes / 1e6, bytes / us / 1e6, bytes / us / 8e6);
  };
  const dim3 g2((unsigned)((n2 + 255) / 256)), g1((unsigned)((n + 255) / 256)), t(256);
  timeit("copy 16B/lane (16 B/row)", 16.0 * n, [&] { hipLaunchKernelGGL(copy16, g2, t, 0, 0, (const d2_t*)x, (d2_t*)y, n2); });
  timeit("triad 16B/lane (24 B/row)", 24.0 * n, [&] { hipLaunchKernelGGL(triad16, g2, t, 0, 0, (const d2_t*)b, (const d2_t*)x, (d2_t*)y, n2, 0.5); });
  timeit("triad 8B/lane (24 B/row)", 24.0 * n, [&] { hipLaunchKernelGGL(triad8, g1, t, 0, 0, b, x, y, n, 0.5); });
  timeit("triad 8B + cls 2B (26 B/row)", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c, g1, t, 0, 0, b, x, c, y, n, 0.5); });
  timeit("triad 16B + cls 4B (26 B/row)", 26.0 * n, [&] { hipLaunchKernelGGL(triad16c, g2, t, 0, 0, (const d2_t*)b, (const d2_t*)x, (const unsigned int*)c, (d2_t*)y, n2, 0.5); });
  timeit("persistent 512x1024, 8B + cls", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_persist, dim3(512), dim3(1024), 0, 0, b, x, c, y, n, 0.5); });
  timeit("persistent interleaved", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_persist_il, dim3(512), dim3(1024), 0, 0, b, x, c, y, n, 0.5); });
  timeit("persistent march order", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_march, dim3(512), dim3(1024), 0, 0, b, x, c, y, 257 * 257, 257, 0.5, 0); });
  timeit("persistent plane-major order", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_march, dim3(512), dim3(1024), 0, 0, b, x, c, y, 257 * 257, 257, 0.5, 1); });
  timeit("march order + 1 plane prefetch", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_march_pf, dim3(512), dim3(1024), 0, 0, b, x, c, y, 257 * 257, 257, 0.5); });
  timeit("march order, 1024 WGs", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_march, dim3(1024), dim3(1024), 0, 0, b, x, c, y, 257 * 257, 257, 0.5, 0); });
  timeit("march order, 2048 WGs", 26.0 * n, [&] { hipLaunchKernelGGL(triad8c_march, dim3(2048), dim3(1024), 0, 0, b, x, c, y, 257 * 257, 257, 0.5, 0); });
  timeit("read 2 streams 16B (16 B/row)", 16.0 * n, [&] { hipLaunchKernelGGL(read2, g2, t, 0, 0, (const d2_t*)b, (const d2_t*)x, y, n2); });
  timeit("write 1 stream 16B (8 B/row)", 8.0 * n, [&] { hipLaunchKernelGGL(write1, g2, t, 0, 0, (d2_t*)y, n2); });
  return 0;
}
