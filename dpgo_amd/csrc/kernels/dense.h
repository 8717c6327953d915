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

static __global__ __launch_bounds__(kBlock) void k_sweep_panel(const double* __restrict__ M, int lda, int kb,
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
static __global__ __launch_bounds__(kBlock) void k_sweep_finish(double* __restrict__ M, int lda) {
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

// out[i] = (float)in[i]: the fp32 operator copies of the multilevel cycle (A P, prolongation blocks, symmetric Q)
static __global__ __launch_bounds__(kBlock) void k_copy_f32(const double* __restrict__ in, float* __restrict__ out, size_t total) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) out[i] = (float)in[i];
}

// unit diagonal on the padding rows N .. lda-1 of a zero-filled lda x lda array
static __global__ __launch_bounds__(kBlock) void k_dense_pad_identity(double* __restrict__ M, int lda, int N) {
  for (int i = N + blockIdx.x * kBlock + threadIdx.x; i < lda; i += gridDim.x * kBlock) M[(size_t)i * lda + i] = 1.0;
}

// fp32 storage of the finished inverse; the fp64 array keeps the SAME (rounded) values, so that what the caller can read
// back (dpgo_problem_multilevel_get) is exactly what the cycle applies
static __global__ __launch_bounds__(kBlock) void k_dense_round_f32(double* __restrict__ M, float* __restrict__ M32, size_t total) {
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const float v = (float)M[e];
    M32[e] = v;
    M[e] = (double)v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Applying the inverse from its LOWER block triangle (the inverse is exactly symmetric, k_sweep_finish): half the bytes of
// the row-streaming kernel (k_ml_coarse_prolong) for the same fp64 result.  Packed storage: tile (I, J <= I) at index
// I (I + 1) / 2 + J, 64 x 64 row-major.  Every tile is used twice -- xc_I += T rc_J and, off the diagonal,
// xc_J += T^T rc_I -- so a workgroup (one chunk = kDenseChunk consecutive tiles of one block row: small chunks, many workgroups -- the kernel is latency-bound per tile) produces partial
// sums: its share of the DIRECT products of block row I (pd[chunk]) and the TRANSPOSED product of each of its tiles
// (pt[I][J], one writer each).  k_dense_sym_finish adds them in a fixed order.  Tiles are streamed with non-temporal loads
// (read once per cycle), the next tile is requested before the current one is used.
constexpr int kDenseChunk = 2;  // measured at 6 252 unknowns: 2 -> 47.9 us (apply + finish), 4 -> 50.4, 8 -> 53.5, 16 -> 71.0
struct DenseChunk {
  int I, J0, cnt, pad;
};

static __global__ __launch_bounds__(kBlock) void k_dense_pack_lower(const double* __restrict__ M, int lda,
                                                             double* __restrict__ packed) {
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  const int i = threadIdx.x >> 2, q0 = (threadIdx.x & 3) * 16;
  const double* __restrict__ src = M + (size_t)(ti * kNB + i) * lda + tj * kNB + q0;
  double* __restrict__ dst = packed + ((size_t)ti * (ti + 1) / 2 + tj) * (kNB * kNB) + i * kNB + q0;
#pragma unroll
  for (int q = 0; q < 16; ++q) dst[q] = src[q];
}

template <int R>
__global__ __launch_bounds__(kBlock) void k_dense_sym_apply(const double* __restrict__ packed,
                                                            const DenseChunk* __restrict__ chunks,
                                                            const double* __restrict__ rc, int N, int lda,
                                                            double* __restrict__ pd, double* __restrict__ pt,
                                                            const DevState* __restrict__ gate) {
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double Ts[kNB][kNB + 1];
  __shared__ double rcI[kNB][R];
  __shared__ double rcJ[kNB][R];
  const DenseChunk ch = chunks[blockIdx.x];
  auto load_rc = [&](double (*dst)[R], int blk) {
    for (int e = threadIdx.x; e < kNB * R; e += kBlock) {
      const int rr = blk * kNB + e / R;
      dst[e / R][e % R] = (rr < N) ? rc[(size_t)rr * R + e % R] : 0.0;
    }
  };
  // lane-contiguous 16-byte pieces (piece v * 256 + t of the tile's 2048): every load instruction of a wave covers 1 KB of
  // consecutive bytes.  (A thread reading its own 128 contiguous bytes makes each instruction touch 64 different cache
  // lines, which non-temporal loads do not keep: 123 us instead of 30.)
  auto load_tile = [&](int J, dbl2 (&t)[8]) {
    const dbl2* __restrict__ src = reinterpret_cast<const dbl2*>(
        packed + ((size_t)ch.I * (ch.I + 1) / 2 + J) * (kNB * kNB)) + threadIdx.x;
#pragma unroll
    for (int v = 0; v < 8; ++v) t[v] = __builtin_nontemporal_load(src + v * kBlock);
  };
  load_rc(rcI, ch.I);
  // right-hand-side rows of a block: kNB * R values, threads 0 .. kNB*R/2 - 1 carry two each (requested one tile ahead)
  constexpr int kRcPairs = kNB * R / 2;
  auto fetch_rc = [&](int blk, double (&v)[2]) {
    v[0] = v[1] = 0.0;
    if ((int)threadIdx.x < kRcPairs) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * threadIdx.x + h, rr = blk * kNB + e / R;
        if (rr < N) v[h] = rc[(size_t)rr * R + e % R];
      }
    }
  };
  // Both products of a tile run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; fragments: A lane l = A[l & 15][l >> 4],
  // B lane l = B[l >> 4][l & 15], C/D register g = row (l >> 4) + 4 g, column l & 15): wave w owns rows 16 w .. 16 w + 15 of
  // T (direct) and of T^T (transposed); the R right-hand sides sit in the first R of the 16 B columns.  With plain FMAs the
  // kernel was bound by LDS reads of the right-hand sides (192 per thread and tile; now 64).
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lr = l & 15, lk = l >> 4;
  const bool bcol = lr < R;
  dbl4 accd = dbl4{0.0, 0.0, 0.0, 0.0};
  dbl2 tnext[8];
  double rnext[2];
  load_tile(ch.J0, tnext);
  fetch_rc(ch.J0, rnext);
  for (int jj = 0; jj < ch.cnt; ++jj) {
    const int J = ch.J0 + jj;
    dbl2 tcur[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) tcur[v] = tnext[v];
    const double rcur[2] = {rnext[0], rnext[1]};
    if (jj + 1 < ch.cnt) {
      load_tile(J + 1, tnext);
      fetch_rc(J + 1, rnext);
    }
    __syncthreads();  // the previous tile's LDS reads are done
#pragma unroll
    for (int v = 0; v < 8; ++v) {  // piece v * 256 + t = row v * 8 + t / 32, columns 2 (t % 32), + 1
      const int pr = v * 8 + (threadIdx.x >> 5), pc = (threadIdx.x & 31) * 2;
      Ts[pr][pc] = tcur[v].x;
      Ts[pr][pc + 1] = tcur[v].y;
    }
    if ((int)threadIdx.x < kRcPairs) {
      (&rcJ[0][0])[2 * threadIdx.x] = rcur[0];
      (&rcJ[0][0])[2 * threadIdx.x + 1] = rcur[1];
    }
    __syncthreads();
    // (the K index a lane group lk supplies in step s is 16 lk + s -- any assignment works as long as A and B agree -- which
    // makes the fragment reads of both products conflict-free on the pitch-65 tile: bank = lr + 16 lk (mod 32))
#pragma unroll 4
    for (int st = 0; st < 16; ++st) {  // xc_I += T rc_J
      const double a = Ts[w * 16 + lr][16 * lk + st];
      const double b = bcol ? rcJ[16 * lk + st][lr] : 0.0;
      accd = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, accd, 0, 0, 0);
    }
    if (J != ch.I) {  // xc_J += T^T rc_I
      dbl4 acct = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
      for (int st = 0; st < 16; ++st) {
        const double a = Ts[16 * lk + st][w * 16 + lr];
        const double b = bcol ? rcI[16 * lk + st][lr] : 0.0;
        acct = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acct, 0, 0, 0);
      }
      if (bcol) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pt[((size_t)ch.I * lda + (size_t)J * kNB + w * 16 + lk + 4 * g) * R + lr] = acct[g];
      }
    }
  }
  if (bcol) {
#pragma unroll
    for (int g = 0; g < 4; ++g) pd[((size_t)blockIdx.x * kNB + w * 16 + lk + 4 * g) * R + lr] = accd[g];
  }
}

// xc rows of block J: the transposed products of the tiles below it (block rows I > J) + its direct partial sums (the chunks
// first[J] .. first[J + 1] - 1 of block row J).  One workgroup per (J, 16 rows): the 16 lanes of a row take every 16th
// partial each (all loads of a lane independent, <= 7 of them) and join by a butterfly: fixed order.
template <int R>
__global__ __launch_bounds__(kBlock) void k_dense_sym_finish(const double* __restrict__ pd, const double* __restrict__ pt,
                                                             const int* __restrict__ first, int nT, int N, int lda,
                                                             double* __restrict__ xc,
                                                             const DevState* __restrict__ gate) {
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  const int J = blockIdx.x, row = blockIdx.y * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
  double sum[R];
#pragma unroll
  for (int r = 0; r < R; ++r) sum[r] = 0.0;
  const double* __restrict__ col = pt + ((size_t)J * kNB + row) * R;
#pragma unroll 8
  for (int I = J + 1 + part; I < nT; I += 16) {
    const double* __restrict__ src = col + (size_t)I * lda * R;
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] += src[r];
  }
  for (int c = first[J] + part; c < first[J + 1]; c += 16) {
    const double* __restrict__ src = pd + ((size_t)c * kNB + row) * R;
#pragma unroll
    for (int r = 0; r < R; ++r) sum[r] += src[r];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sum[r] += __shfl_xor(sum[r], o);
  }
  if (part == 0 && J * kNB + row < N) {
#pragma unroll
    for (int r = 0; r < R; ++r) xc[((size_t)J * kNB + row) * R + r] = sum[r];
  }
}
