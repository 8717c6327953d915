// kernels/init.h -- vector kernels of the device-side chordal initialisation (masked conjugate gradients).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// Chordal initialisation (src/DPGO_solver.cpp:220-269; constructBMatrices / recoverTranslations,
// src/DPGO_utils.cpp:346-462): two linear least-squares problems -- rotations with pose 0 pinned to the identity, then
// translations with the rotations fixed and t_0 = 0.  The reference solves them with SPQR; here their normal equations
// are solved by Jacobi-preconditioned conjugate gradients whose operator is the block-SpMM of the hot path (k_spmm)
// restricted to a subset of the unknowns by a MASK: mode 0 = rotation columns (c < D), mode 1 = the translation column
// (c == D), pose 0 always excluded.  Vectors are pose tiles [n][D+1][R] with R = D.
__device__ __forceinline__ bool init_mask(size_t e, int T, int R, int D, int mode) {
  const size_t i = e / (size_t)T;
  const int c = (int)((e - i * (size_t)T) / (size_t)R);
  return i > 0 && (mode == 0 ? c < D : c == D);
}

// y = mask(a x + b y)   (y may alias x)
static __global__ __launch_bounds__(kBlock) void k_init_axpby(double a, const double* x, double b, double* y, size_t total,
                                                       int T, int R, int D, int mode) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock)
    y[e] = init_mask(e, T, R, D, mode) ? (b == 0.0 ? a * x[e] : fma(a, x[e], b * y[e])) : 0.0;  // b = 0: y is output only
                                                                                                // (it may be uninitialised memory: 0 * NaN)
}

// z = mask(r / diag),  diag[i][c] = A_ii[c][c]  (Jacobi)
static __global__ __launch_bounds__(kBlock) void k_init_jacobi(const double* __restrict__ r, const double* __restrict__ diag,
                                                        double* __restrict__ z, size_t total, int T, int R, int D,
                                                        int mode) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const double dg = diag[e / (size_t)R];
    z[e] = (init_mask(e, T, R, D, mode) && dg > 0.0) ? r[e] / dg : 0.0;
  }
}

// per-workgroup partial sums of <x, y> (fixed order; the host adds the partials in index order)
static __global__ __launch_bounds__(kBlock) void k_init_dot(const double* __restrict__ x, const double* __restrict__ y,
                                                     double* __restrict__ partial, size_t total) {
  __shared__ double red[kWaves * kNP];
  double v[1] = {0.0};
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock)
    v[0] = fma(x[e], y[e], v[0]);
  block_allreduce<1>(v, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
}

// diag[i * B + c] = A_ii[c][c]
template <int D>
__global__ __launch_bounds__(kBlock) void k_init_diag(BsrDev A, double* __restrict__ diag, int n) {
  constexpr int B = D + 1, BB = B * B;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    for (int t = A.rowptr[i]; t < A.rowptr[i + 1]; ++t)
      if (A.colidx[t] == i) {
#pragma unroll
        for (int c = 0; c < B; ++c) diag[(size_t)i * B + c] = A.vals[(size_t)t * BB + c * B + c];
      }
  }
}

// y = a x + b y, no mask
static __global__ __launch_bounds__(kBlock) void k_axpby_plain(double a, const double* __restrict__ x, double b,
                                                        double* __restrict__ y, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock)
    y[e] = (b == 0.0) ? a * x[e] : fma(a, x[e], b * y[e]);
}
