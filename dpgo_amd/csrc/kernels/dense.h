// kernels/dense.h -- in-place inverse of the dense SPD coarsest operator of the multilevel preconditioner (blocked Gauss-Jordan).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// The reference factors Q + 0.1 I with CHOLMOD on the host (PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613).
// The device path needs an operator it can APPLY as a stream, so the coarsest Galerkin operator (<= ~6400 unknowns) is
// inverted once per Q into a dense array in HBM:  blocked Gauss-Jordan without pivoting (the matrix is SPD), block size 64,
// two launches per block step over the lda x lda array (lda = multiple of 64; padding rows carry a unit diagonal):
//   k_gj_panel  : every workgroup inverts the 64x64 pivot block D in LDS (redundantly: it saves a launch and a
//                 dependency), forms its tile of the scaled pivot row  Rx = D^-1 A[kb,:]  (Rx[:,kb] = D^-1) and saves its
//                 tile of the pivot column  W = A[:,kb]  (W[kb] = 0);
//   k_gj_update : A[i,j] <- (j in kb ? 0 : A[i,j]) - W_i Rx_j  for the other block rows, A[kb,:] <- Rx.
// The rank-64 update is the only O(N^3) piece; its 64x64x64 tile products run either on plain fp64 FMAs (4x4 register
// tiles) or on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), selected by the MFMA template flag.
constexpr int kNB = 64;
typedef double dbl4 __attribute__((ext_vector_type(4)));

// In-place Gauss-Jordan inverse of a 64x64 SPD block in LDS (row pitch 65: column accesses hit distinct banks).
// 256 threads; thread t owns row t/4, columns (t%4)*16 .. +15.
__device__ __forceinline__ void lds_gj_invert(double (*Ds)[kNB + 1]) {
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  for (int p = 0; p < kNB; ++p) {
    __syncthreads();
    const double inv = 1.0 / Ds[p][p];
    const double f = Ds[i][p];
    double prow[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) prow[q] = Ds[p][q0 + q];
    __syncthreads();
    if (i == p) {
#pragma unroll
      for (int q = 0; q < 16; ++q) Ds[p][q0 + q] = (q0 + q == p) ? inv : prow[q] * inv;
    } else {
      const double g = f * inv;
#pragma unroll
      for (int q = 0; q < 16; ++q) Ds[i][q0 + q] = (q0 + q == p) ? -g : fma(-g, prow[q], Ds[i][q0 + q]);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_gj_panel(const double* __restrict__ M, int lda, int kb,
                                                     double* __restrict__ W, double* __restrict__ Rx) {
  __shared__ double Ds[kNB][kNB + 1];
  __shared__ double As[kNB][kNB + 1];
  const int tj = blockIdx.x;
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  const double* __restrict__ drow = M + (size_t)(kb * kNB + i) * lda;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    Ds[i][q0 + q] = drow[kb * kNB + q0 + q];
    As[i][q0 + q] = drow[tj * kNB + q0 + q];
  }
  {
    const double* __restrict__ wrow = M + (size_t)(tj * kNB + i) * lda + kb * kNB;
    double* __restrict__ wout = W + (size_t)(tj * kNB + i) * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) wout[q0 + q] = (tj == kb) ? 0.0 : wrow[q0 + q];
  }
  lds_gj_invert(Ds);
  double acc[16];
  if (tj == kb) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = Ds[i][q0 + q];
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int k = 0; k < kNB; ++k) {
      const double dv = Ds[i][k];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = fma(dv, As[k][q0 + q], acc[q]);
    }
  }
  double* __restrict__ rout = Rx + (size_t)i * lda + tj * kNB;
#pragma unroll
  for (int q = 0; q < 16; ++q) rout[q0 + q] = acc[q];
}

template <bool MFMA>
__global__ __launch_bounds__(kBlock) void k_gj_update(double* __restrict__ M, int lda, int kb,
                                                      const double* __restrict__ W, const double* __restrict__ Rx) {
  __shared__ double Ws[kNB][kNB + 1];
  __shared__ double Rs[kNB][kNB + 1];
  const int tj = blockIdx.x, ti = blockIdx.y;
  {
    const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
    const double* __restrict__ rrow = Rx + (size_t)i * lda + tj * kNB;
    if (ti == kb) {  // the pivot block row becomes the scaled row
      double* __restrict__ out = M + (size_t)(kb * kNB + i) * lda + tj * kNB;
#pragma unroll
      for (int q = 0; q < 16; ++q) out[q0 + q] = rrow[q0 + q];
      return;
    }
    const double* __restrict__ wrow = W + (size_t)(ti * kNB + i) * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      Ws[i][q0 + q] = wrow[q0 + q];
      Rs[i][q0 + q] = rrow[q0 + q];
    }
  }
  __syncthreads();
  double* __restrict__ C = M + (size_t)(ti * kNB) * lda + tj * kNB;
  const bool zero = (tj == kb);
  if constexpr (MFMA) {
    // one wavefront per 16 rows, four 16x16 accumulators across the 64 columns; v_mfma_f64_16x16x4_f64 fragments:
    // A: lane l holds A[l & 15][l >> 4], B: lane l holds B[l >> 4][l & 15], C/D reg g: row (l >> 4) + 4 g, col l & 15
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = l & 15, lk = l >> 4;
    dbl4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = dbl4{0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < kNB; k0 += 4) {
      const double a = Ws[w * 16 + lr][k0 + lk];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const double b = Rs[k0 + lk][nt * 16 + lr];
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        double* __restrict__ c = C + (size_t)(w * 16 + lk + 4 * g) * lda + nt * 16 + lr;
        const double old = zero ? 0.0 : *c;
        *c = old - acc[nt][g];
      }
    }
  } else {
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;  // 4 x 4 register tile: rows 4 ty.., columns 4 tx..
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k = 0; k < kNB; ++k) {
      double wv[4], rv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wv[a] = Ws[ty * 4 + a][k];
#pragma unroll
      for (int b = 0; b < 4; ++b) rv[b] = Rs[k][tx * 4 + b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(wv[a], rv[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      double* __restrict__ c = C + (size_t)(ty * 4 + a) * lda + tx * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const double old = zero ? 0.0 : c[b];
        c[b] = old - acc[a][b];
      }
    }
  }
}

// unit diagonal on the padding rows N .. lda-1 of a zero-filled lda x lda array
__global__ __launch_bounds__(kBlock) void k_dense_pad_identity(double* __restrict__ M, int lda, int N) {
  for (int i = N + blockIdx.x * kBlock + threadIdx.x; i < lda; i += gridDim.x * kBlock) M[(size_t)i * lda + i] = 1.0;
}

// fp32 storage of the finished inverse; the fp64 array keeps the SAME (rounded) values, so that what the caller can read
// back (dpgo_problem_multilevel_get) is exactly what the cycle applies
__global__ __launch_bounds__(kBlock) void k_dense_round_f32(double* __restrict__ M, float* __restrict__ M32, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const float v = (float)M[e];
    M32[e] = v;
    M[e] = (double)v;
  }
}
