// stream_lab.hip -- does the "lane = tile column" access pattern (5 x 8-byte loads per lane at a
// 40-byte stride) cost bandwidth against 16-byte-per-lane streaming?  z = a*x + y over 3 x 16 MB vectors
// (n = 100k poses x 20 doubles), plus a 6-stream variant shaped like k_tcg_update (4 reads + 2 writes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double double2_t __attribute__((ext_vector_type(2)));
constexpr int R = 5, T = 20;

__global__ __launch_bounds__(256) void k_vec16(const double2_t* __restrict__ x, const double2_t* __restrict__ y, double2_t* __restrict__ z, double a, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    double2_t u = x[i], v = y[i];
    u.x = fma(a, u.x, v.x); u.y = fma(a, u.y, v.y);
    z[i] = u;
  }
}
// PC layout: lane owns 5 contiguous doubles; a wave covers 64*40 B; tiles of 64 poses per block
__global__ __launch_bounds__(256) void k_pc(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ z, double a, int ncol) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < ncol; c += gridDim.x * 256) {
    double u[R], v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) u[k] = x[(size_t)c * R + k];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = y[(size_t)c * R + k];
#pragma unroll
    for (int k = 0; k < R; ++k) z[(size_t)c * R + k] = fma(a, u[k], v[k]);
  }
}
// PC layout through LDS: the wave's 2560-byte span is moved with 16-byte lane-linear accesses
__device__ __forceinline__ void span_load(const double* __restrict__ g, double* lds, int lane) {
  const double2_t* __restrict__ s = reinterpret_cast<const double2_t*>(g);
  double2_t* d = reinterpret_cast<double2_t*>(lds);
  d[lane] = s[lane]; d[lane + 64] = s[lane + 64];
  if (lane < 32) d[lane + 128] = s[lane + 128];
}
__device__ __forceinline__ void span_store(double* __restrict__ g, const double* lds, int lane) {
  double2_t* __restrict__ d = reinterpret_cast<double2_t*>(g);
  const double2_t* s = reinterpret_cast<const double2_t*>(lds);
  d[lane] = s[lane]; d[lane + 64] = s[lane + 64];
  if (lane < 32) d[lane + 128] = s[lane + 128];
}
__global__ __launch_bounds__(256) void k_pc_lds(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ z, double a, int nspans) {
  __shared__ __attribute__((aligned(16))) double sm[4][3][320];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int sp = blockIdx.x * 4 + w; sp < nspans; sp += gridDim.x * 4) {
    span_load(x + (size_t)sp * 320, sm[w][0], lane);
    span_load(y + (size_t)sp * 320, sm[w][1], lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < R; ++k) sm[w][2][lane * R + k] = fma(a, sm[w][0][lane * R + k], sm[w][1][lane * R + k]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    span_store(z + (size_t)sp * 320, sm[w][2], lane);
    __builtin_amdgcn_wave_barrier();
  }
}
// 6-stream variants (eta += a*delta ; r += a*Hd): 4 reads, 2 writes
__global__ __launch_bounds__(256) void k6_vec16(double2_t* __restrict__ eta, const double2_t* __restrict__ dl, double2_t* __restrict__ r, const double2_t* __restrict__ hd, double a, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    double2_t e = eta[i], d = dl[i], rr = r[i], h = hd[i];
    e.x = fma(a, d.x, e.x); e.y = fma(a, d.y, e.y); rr.x = fma(a, h.x, rr.x); rr.y = fma(a, h.y, rr.y);
    eta[i] = e; r[i] = rr;
  }
}
__global__ __launch_bounds__(256) void k6_pc(double* __restrict__ eta, const double* __restrict__ dl, double* __restrict__ r, const double* __restrict__ hd, double a, int ncol) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < ncol; c += gridDim.x * 256) {
    double e[R], d[R], rr[R], h[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { e[k] = eta[(size_t)c * R + k]; d[k] = dl[(size_t)c * R + k]; rr[k] = r[(size_t)c * R + k]; h[k] = hd[(size_t)c * R + k]; }
#pragma unroll
    for (int k = 0; k < R; ++k) { eta[(size_t)c * R + k] = fma(a, d[k], e[k]); r[(size_t)c * R + k] = fma(a, h[k], rr[k]); }
  }
}
int main() {
  const int n = 100000; const size_t N = (size_t)n * T;
  std::vector<double*> v(6);
  for (auto& p : v) { HC(hipMalloc(&p, N * 8)); HC(hipMemset(p, 0, N * 8)); }
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  auto run = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    HC(hipEventRecord(e0)); for (int i = 0; i < 100; ++i) launch(); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1)); HC(hipGetLastError());
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.2f us  %7.1f GB/s\n", name, ms * 10, bytes / (ms * 10) / 1e3);
  };
  for (int grid : {1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "axpy 16B/lane grid=%d", grid);
    run(nm, 3.0 * N * 8, [&] { hipLaunchKernelGGL(k_vec16, dim3(grid), dim3(256), 0, 0, (double2_t*)v[0], (double2_t*)v[1], (double2_t*)v[2], 0.5, N / 2); });
    snprintf(nm, 64, "axpy PC 5x8B/lane grid=%d", grid);
    run(nm, 3.0 * N * 8, [&] { hipLaunchKernelGGL(k_pc, dim3(grid), dim3(256), 0, 0, v[0], v[1], v[2], 0.5, n * 4); });
    snprintf(nm, 64, "axpy PC via LDS 16B grid=%d", grid);
    run(nm, 3.0 * N * 8, [&] { hipLaunchKernelGGL(k_pc_lds, dim3(grid), dim3(256), 0, 0, v[0], v[1], v[2], 0.5, n * 4 / 64); });
    snprintf(nm, 64, "6-stream 16B/lane grid=%d", grid);
    run(nm, 6.0 * N * 8, [&] { hipLaunchKernelGGL(k6_vec16, dim3(grid), dim3(256), 0, 0, (double2_t*)v[0], (double2_t*)v[1], (double2_t*)v[2], (double2_t*)v[3], 0.5, N / 2); });
    snprintf(nm, 64, "6-stream PC 5x8B/lane grid=%d", grid);
    run(nm, 6.0 * N * 8, [&] { hipLaunchKernelGGL(k6_pc, dim3(grid), dim3(256), 0, 0, v[0], v[1], v[2], v[3], 0.5, n * 4); });
  }
  return 0;
}
