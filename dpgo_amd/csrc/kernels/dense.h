// kernels/dense.h -- in-place inverse of the dense SPD coarsest operator of the multilevel preconditioner (blocked Gauss-Jordan).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// The reference factors Q + 0.1 I with CHOLMOD on the host (PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613).
// The device path needs an operator it can APPLY as a stream, so the coarsest Galerkin operator (<= ~6400 unknowns) is
// inverted once per Q into a dense array in HBM by the blocked SYMMETRIC SWEEP (Gauss-Jordan in the sign convention that
// keeps every intermediate matrix symmetric: sweeping all pivots of A leaves -A^-1), block size 64, no pivoting (SPD).  Only
// the lower block triangle is maintained -- half the bytes and flops of the plain in-place Gauss-Jordan of the first version
// (44 ms -> see DESIGN.md section 5 at 6 252 unknowns) -- and the result is exactly symmetric.  Per block step kb over the
// lda x lda array (lda = multiple of 64; padding rows carry a unit diagonal), two launches:
//   k_sweep_panel  : every workgroup inverts the 64x64 pivot block D in LDS (redundantly: it saves a launch and a
//                    dependency), reads its tile A_kj of the pivot block row (from the lower triangle: transposed for j > kb),
//                    forms  Rx_j = D^-1 A_kj  (Rx_kb = D^-1)  and saves  W_j = A_kj^T  (W_kb = 0);
//   k_sweep_update : lower tiles (i >= j):  A_ij <- A_ij - W_i Rx_j;  A_kj <- Rx_j,  A_ik <- Rx_i^T,  A_kk <- -D^-1.
// k_sweep_finish negates and mirrors the lower triangle into the full array the apply kernel streams.
// The rank-64 update is the only O(N^3) piece; its 64x64x64 tile products run either on plain fp64 FMAs (4x4 register
// tiles) or on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), selected by the MFMA template flag; the tile of A is requested
// before the operands are staged, so that its latency overlaps the products.
constexpr int kNB = 64;
typedef double dbl4 __attribute__((ext_vector_type(4)));

// In-place Gauss-Jordan inverse of a 64x64 SPD block in LDS (row pitch 65).  256 threads; thread t owns row t/4, columns
// (t%4)*16 .. +15 and keeps them in REGISTERS for all 64 pivot steps.  The pivot row travels through a double-buffered
// 64-entry LDS line written by its owners at the end of the previous step, so one __syncthreads per pivot suffices (the
// first version re-read everything from LDS with two barriers per pivot: 133 us per block step, the serial part of the
// whole setup); the pivot-column entry of a row comes from the owning lane of the row's quad by a shuffle.
__device__ __forceinline__ void lds_gj_invert(double (*Ds)[kNB + 1]) {
  __shared__ double pr[2][kNB];
  const int i = threadIdx.x >> 2, sg = threadIdx.x & 3, q0 = sg * 16;
  const int lane = threadIdx.x & 63;
  double row[16];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) row[q] = Ds[i][q0 + q];
  if (i == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) pr[0][q0 + q] = row[q];
  }
  for (int seg = 0; seg < 4; ++seg) {
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      const int p = seg * 16 + pp;
      __syncthreads();
      const double* __restrict__ prp = pr[p & 1];
      double prow[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) prow[q] = prp[q0 + q];
      const double inv = 1.0 / prp[p];
      const double f = __shfl(row[pp], (lane & ~3) | seg);  // element (i, p): lane `seg` of the row's quad holds it
      if (i == p) {
#pragma unroll
        for (int q = 0; q < 16; ++q) row[q] = (q0 + q == p) ? inv : prow[q] * inv;
      } else {
        const double g = f * inv;
#pragma unroll
        for (int q = 0; q < 16; ++q) row[q] = (q0 + q == p) ? -g : fma(-g, prow[q], row[q]);
      }
      if (i == p + 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) pr[(p + 1) & 1][q0 + q] = row[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) Ds[i][q0 + q] = row[q];
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_sweep_panel(const double* __restrict__ M, int lda, int kb,
                                                        double* __restrict__ W, double* __restrict__ Rx) {
  __shared__ double Ds[kNB][kNB + 1];
  __shared__ double As[kNB][kNB + 1];  // A_kj, row-major
  const int tj = blockIdx.x;
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  {
    const double* __restrict__ drow = M + (size_t)(kb * kNB + i) * lda + kb * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) Ds[i][q0 + q] = drow[q0 + q];
  }
  if (tj < kb) {  // stored as tile (kb, tj)
    const double* __restrict__ arow = M + (size_t)(kb * kNB + i) * lda + tj * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) As[i][q0 + q] = arow[q0 + q];
  } else if (tj > kb) {  // stored as tile (tj, kb) = A_kj^T
    const double* __restrict__ arow = M + (size_t)(tj * kNB + i) * lda + kb * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) As[q0 + q][i] = arow[q0 + q];
  }
  __syncthreads();
  {
    double* __restrict__ wout = W + (size_t)(tj * kNB + i) * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) wout[q0 + q] = (tj == kb) ? 0.0 : As[q0 + q][i];
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
__global__ __launch_bounds__(kBlock) void k_sweep_update(double* __restrict__ M, int lda, int kb,
                                                         const double* __restrict__ W, const double* __restrict__ Rx) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;  // lower block triangle only
  __shared__ double Ws[kNB][kNB + 1];
  __shared__ double Rs[kNB][kNB + 1];
  double* __restrict__ C = M + (size_t)(ti * kNB) * lda + tj * kNB;
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  if (ti == kb || tj == kb) {  // the swept block row / column: A_kj <- Rx_j, A_ik <- Rx_i^T, A_kk <- -D^-1
    const int src = (ti == kb) ? tj : ti;
    const double* __restrict__ rrow = Rx + (size_t)i * lda + src * kNB;
    if (ti == kb) {
      const double sgn = (tj == kb) ? -1.0 : 1.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) C[(size_t)i * lda + q0 + q] = sgn * rrow[q0 + q];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) Rs[i][q0 + q] = rrow[q0 + q];
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) C[(size_t)i * lda + q0 + q] = Rs[q0 + q][i];
    }
    return;
  }
  if constexpr (MFMA) {
    // one wavefront per 16 rows, four 16x16 accumulators across the 64 columns; v_mfma_f64_16x16x4_f64 fragments:
    // A: lane l holds A[l & 15][l >> 4], B: lane l holds B[l >> 4][l & 15], C/D reg g: row (l >> 4) + 4 g, col l & 15
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = l & 15, lk = l >> 4;
    double old[4][4];  // requested first: the tile's latency overlaps the staging of W, Rx and the products
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) old[nt][g] = C[(size_t)(w * 16 + lk + 4 * g) * lda + nt * 16 + lr];
    {
      const double* __restrict__ wrow = W + (size_t)(ti * kNB + i) * kNB;
      const double* __restrict__ rrow = Rx + (size_t)i * lda + tj * kNB;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        Ws[i][q0 + q] = wrow[q0 + q];
        Rs[i][q0 + q] = rrow[q0 + q];
      }
    }
    __syncthreads();
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
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) C[(size_t)(w * 16 + lk + 4 * g) * lda + nt * 16 + lr] = old[nt][g] - acc[nt][g];
  } else {
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;  // 4 x 4 register tile: rows 4 ty.., columns 4 tx..
    double old[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) old[a][b] = C[(size_t)(ty * 4 + a) * lda + tx * 4 + b];
    {
      const double* __restrict__ wrow = W + (size_t)(ti * kNB + i) * kNB;
      const double* __restrict__ rrow = Rx + (size_t)i * lda + tj * kNB;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        Ws[i][q0 + q] = wrow[q0 + q];
        Rs[i][q0 + q] = rrow[q0 + q];
      }
    }
    __syncthreads();
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
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) C[(size_t)(ty * 4 + a) * lda + tx * 4 + b] = old[a][b] - acc[a][b];
  }
}

// A^-1 = -(swept lower triangle), mirrored into the full array (exactly symmetric)
__global__ __launch_bounds__(kBlock) void k_sweep_finish(double* __restrict__ M, int lda) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  __shared__ double Ts[kNB][kNB + 1];
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  double* __restrict__ lo = M + (size_t)(ti * kNB) * lda + tj * kNB;
  double v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    v[q] = -lo[(size_t)i * lda + q0 + q];
    Ts[i][q0 + q] = v[q];
  }
  __syncthreads();
  if (ti == tj) {  // diagonal tile: its own lower part decides
#pragma unroll
    for (int q = 0; q < 16; ++q) lo[(size_t)i * lda + q0 + q] = (q0 + q <= i) ? v[q] : Ts[q0 + q][i];
  } else {
    double* __restrict__ up = M + (size_t)(tj * kNB) * lda + ti * kNB;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      lo[(size_t)i * lda + q0 + q] = v[q];
      up[(size_t)i * lda + q0 + q] = Ts[q0 + q][i];
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
