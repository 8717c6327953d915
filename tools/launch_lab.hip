// launch_lab.hip -- what is the "fixed ~10 us" of a launch in the multi-launch tCG loop made of?  
//
// Chains of DEPENDENT launches on one stream, timed with HIP events over the whole chain (per-launch figure = total / N):
//   empty        nothing but the dispatch of a dependent kernel (the kernel boundary itself)
//   state        + every workgroup loads the 200-byte solver state record (scalar loads), workgroup 0 stores it back
//   partials     + the scalar prologue of the tCG kernels: every workgroup re-reduces the previous kernel's per-workgroup
//                  partial sums (G entries, stride kNP doubles; block all-reduce), then writes its own partial at the end
//   one_record   + the prologue reads ONE record instead; the LAST workgroup to finish (device-scope counter) reduces the
//                  partials in a fixed order and writes the record (pattern of TcgStopCheck, kernels/multilevel.h)
// and the same four in front of / behind a pure stream over B bytes (16-byte non-temporal loads, one 8-byte FMA reduction
// per lane), at two sizes, so that ramp-up / tail and the streaming rate separate from the fixed part:
//   t(B) = fixed + B / rate.
// The same chains replayed from a hipGraph (stream capture) show how much of the boundary is host-side enqueue.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/launch_lab.hip -o tools/launch_lab ; run: tools/launch_lab [grid] [chain]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int kBlock = 256, kNP = 4;
typedef double dbl2 __attribute__((ext_vector_type(2)));
struct State { double v[20]; int w[10]; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// MODE 0 empty, 1 state, 2 partials, 3 one_record (+ last workgroup reduces).  STREAM: also stream `n2` 16-byte pieces.
template <int MODE, bool STREAM>
__global__ __launch_bounds__(kBlock) void k_chain(const State* __restrict__ sin, State* __restrict__ sout,
                                                  const double* __restrict__ pin, double* __restrict__ pout,
                                                  const double* __restrict__ rec_in, double* __restrict__ rec_out,
                                                  unsigned* __restrict__ counter, const dbl2* __restrict__ data, size_t n2) {
  __shared__ double red[4];
  __shared__ int s_last;
  double scal = 0.0;
  State st;
  if constexpr (MODE >= 1) {
#pragma unroll
    for (int k = 0; k < 20; ++k) st.v[k] = sin->v[k];
#pragma unroll
    for (int k = 0; k < 10; ++k) st.w[k] = sin->w[k];
  }
  if constexpr (MODE == 2) {
    double a = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock) a += pin[i * kNP];
    scal = block_sum(a, red);
  }
  if constexpr (MODE == 3) scal = rec_in[0];
  if constexpr (MODE >= 1) {
    st.v[0] = st.v[1] / (scal + 2.0);
    st.w[0] += 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 20; ++k) sout->v[k] = st.v[k];
#pragma unroll
      for (int k = 0; k < 10; ++k) sout->w[k] = st.w[k];
    }
  }
  double acc = 0.0;
  if constexpr (STREAM) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (size_t)gridDim.x * kBlock) {
      const dbl2 u = __builtin_nontemporal_load(data + i);
      acc = fma(u.x, st.v[0] + 1.0, acc) + u.y;
    }
  }
  if constexpr (MODE >= 2) {
    const double s = block_sum(acc + 1.0, red);
    if (threadIdx.x == 0) pout[blockIdx.x * kNP] = s;
  } else if constexpr (STREAM) {
    if (acc == 12345.678) pout[blockIdx.x * kNP] = acc;  // keep the loads
  }
  if constexpr (MODE == 3) {
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned t = atomicAdd(counter, 1u);
      s_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      double a = 0.0;
      for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock) a += __hip_atomic_load(pout + i * kNP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double s = block_sum(a, red);
      if (threadIdx.x == 0) {
        rec_out[0] = s;
        *counter = 0;
      }
    }
  }
}

struct Bufs {
  State* st[2];
  double* part[2];
  double* rec[2];
  unsigned* counter;
  dbl2* data;
};

template <int MODE, bool STREAM>
void enqueue(const Bufs& b, int grid, int n, size_t n2, hipStream_t s) {
  for (int i = 0; i < n; ++i)
    hipLaunchKernelGGL((k_chain<MODE, STREAM>), dim3(grid), dim3(kBlock), 0, s, b.st[i & 1], b.st[(i + 1) & 1], b.part[i & 1],
                       b.part[(i + 1) & 1], b.rec[i & 1], b.rec[(i + 1) & 1], b.counter, b.data, n2);
}

template <int MODE, bool STREAM>
double time_chain(const Bufs& b, int grid, int n, size_t n2, hipStream_t s, bool graph) {
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  float best = 1e30f;
  hipGraphExec_t exec = nullptr;
  if (graph) {
    hipGraph_t g;
    HC(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    enqueue<MODE, STREAM>(b, grid, n, n2, s);
    HC(hipStreamEndCapture(s, &g));
    HC(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    HC(hipGraphDestroy(g));
  }
  for (int rep = 0; rep < 4; ++rep) {
    HC(hipStreamSynchronize(s));
    HC(hipEventRecord(e0, s));
    if (graph) HC(hipGraphLaunch(exec, s));
    else enqueue<MODE, STREAM>(b, grid, n, n2, s);
    HC(hipEventRecord(e1, s));
    HC(hipEventSynchronize(e1));
    float ms = 0;
    HC(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  if (exec) HC(hipGraphExecDestroy(exec));
  HC(hipEventDestroy(e0));
  HC(hipEventDestroy(e1));
  return 1e3 * best / n;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 768;
  const int chain = argc > 2 ? atoi(argv[2]) : 300;
  hipStream_t s;
  HC(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Bufs b;
  for (int k = 0; k < 2; ++k) {
    HC(hipMalloc(&b.st[k], sizeof(State)));
    HC(hipMemset(b.st[k], 0, sizeof(State)));
    HC(hipMalloc(&b.part[k], sizeof(double) * kNP * 4096));
    HC(hipMemset(b.part[k], 0, sizeof(double) * kNP * 4096));
    HC(hipMalloc(&b.rec[k], 64));
    HC(hipMemset(b.rec[k], 0, 64));
  }
  HC(hipMalloc(&b.counter, 64));
  HC(hipMemset(b.counter, 0, 64));
  const size_t big = (size_t)1200 << 20;  // rotate through > 256 MB so that the stream comes from HBM
  HC(hipMalloc(&b.data, big));
  HC(hipMemset(b.data, 0, big));
  HC(hipDeviceSynchronize());
  const char* names[4] = {"empty", "state", "partials", "one_record"};
  printf("grid %d workgroups x %d threads, chains of %d dependent launches, us per launch\n", grid, kBlock, chain);
  printf("%-12s %10s %10s\n", "prologue", "stream", "hipGraph");
  double r[4][2];
  r[0][0] = time_chain<0, false>(b, grid, chain, 0, s, false); r[0][1] = time_chain<0, false>(b, grid, chain, 0, s, true);
  r[1][0] = time_chain<1, false>(b, grid, chain, 0, s, false); r[1][1] = time_chain<1, false>(b, grid, chain, 0, s, true);
  r[2][0] = time_chain<2, false>(b, grid, chain, 0, s, false); r[2][1] = time_chain<2, false>(b, grid, chain, 0, s, true);
  r[3][0] = time_chain<3, false>(b, grid, chain, 0, s, false); r[3][1] = time_chain<3, false>(b, grid, chain, 0, s, true);
  for (int m = 0; m < 4; ++m) printf("%-12s %10.2f %10.2f\n", names[m], r[m][0], r[m][1]);
  // with a stream of B bytes per launch (the same buffer every launch: 20 MB is Infinity-Cache resident, 300 and 600 MB are not)
  const size_t sizes[3] = {(size_t)20 << 20, (size_t)300 << 20, (size_t)600 << 20};
  printf("\n%-12s %12s %12s %12s   (us per launch with a stream of 20 / 300 / 600 MB, the two large ones beyond the Infinity Cache; fixed = extrapolation to 0 bytes from the two large sizes)\n",
         "prologue", "20 MB", "300 MB", "600 MB");
  for (int m = 0; m < 4; ++m) {
    double t[3];
    for (int q = 0; q < 3; ++q) {
      const size_t n2 = sizes[q] / 16;
      const int n = q == 2 ? 60 : 150;
      switch (m) {
        case 0: t[q] = time_chain<0, true>(b, grid, n, n2, s, false); break;
        case 1: t[q] = time_chain<1, true>(b, grid, n, n2, s, false); break;
        case 2: t[q] = time_chain<2, true>(b, grid, n, n2, s, false); break;
        default: t[q] = time_chain<3, true>(b, grid, n, n2, s, false); break;
      }
    }
    const double slope = (t[2] - t[1]) / (600.0 - 300.0);
    printf("%-12s %12.2f %12.2f %12.2f   slope %.4f us/MB = %.2f TB/s, fixed %.2f us\n", names[m], t[0], t[1], t[2], slope,
           1.048576 / slope, t[1] - slope * 300.0);
  }
  return 0;
}
