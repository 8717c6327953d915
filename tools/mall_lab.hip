// mall_lab.hip -- what do "warm" (operands resident in the 256 MB Infinity Cache) and "cold" (rotating operand sets,
// > 256 MB in total) streaming rates look like on this chip?  Pure 16-byte-per-lane read-reduce and copy kernels over
// buffers of the size of the 100k-pose Q (123 MB), cycling through K private buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const double2_t* __restrict__ x, double* __restrict__ out, size_t n2) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
    double2_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y;
  }
  for (; i < n2; i += stride) acc += x[i].x + x[i].y;
  if (acc == 1.2345e300) out[0] = acc;  // never true; keeps the loads alive
}
__global__ __launch_bounds__(256) void k_copy(const double2_t* __restrict__ x, double2_t* __restrict__ y, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) y[i] = x[i];
}

int main() {
  const size_t bytes = 123ull << 20;
  const size_t n2 = bytes / 16;
  const int KMAX = 6;
  std::vector<double2_t*> bufs(KMAX), dst(KMAX);
  for (int k = 0; k < KMAX; ++k) {
    HC(hipMalloc(&bufs[k], bytes));
    HC(hipMalloc(&dst[k], bytes));
    HC(hipMemset(bufs[k], 0, bytes));
    HC(hipMemset(dst[k], 0, bytes));
  }
  double* out;
  HC(hipMalloc(&out, 8));
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  const int reps = 60;
  for (int grid : {2048, 8192}) {
    for (int K : {1, 2, 4, 6}) {
      auto time = [&](auto launch) {
        for (int i = 0; i < 6; ++i) launch(i);
        HC(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(i);
        HC(hipEventRecord(e1));
        HC(hipEventSynchronize(e1));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps * 1e3;
      };
      double t1 = time([&](int i) { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, bufs[i % K], out, n2); });
      double t4 = time([&](int i) { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, bufs[i % K], out, n2); });
      double t8 = time([&](int i) { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, 0, bufs[i % K], out, n2); });
      double tc = time([&](int i) { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, bufs[i % K], dst[i % K], n2); });
      printf("grid %5d K %d (%4zu MB total): read u1 %6.1f us %5.2f TB/s | u4 %6.1f us %5.2f TB/s | u8 %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s (r+w)\n",
             grid, K, K * bytes >> 20, t1, bytes / t1 * 1e-6, t4, bytes / t4 * 1e-6, t8, bytes / t8 * 1e-6, tc, 2.0 * bytes / tc * 1e-6);
    }
  }
  return 0;
}
