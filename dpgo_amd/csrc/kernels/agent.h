// kernels/agent.h -- agent-level kernels: public-pose packing, edge residuals + GNC-TLS weights, value rebuild of Q / coupling blocks.
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ K11: pack public poses
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_gather_tiles(const double* __restrict__ src,
                                                         const int32_t* __restrict__ idx, int count,
                                                         double* __restrict__ dst) {
  constexpr int T = (D + 1) * R;
  const size_t total = (size_t)count * T;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const int k = (int)(e / T), w = (int)(e - (size_t)k * T);
    dst[e] = src[(size_t)idx[k] * T + w];
  }
}

// Inverse of the above: dst tile idx[k] = src tile k (a permutation of whole pose vectors: dpgo_permute_tiles_device).
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_scatter_tiles(const double* __restrict__ src,
                                                          const int32_t* __restrict__ idx, int count,
                                                          double* __restrict__ dst) {
  constexpr int T = (D + 1) * R;
  const size_t total = (size_t)count * T;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const int k = (int)(e / T), w = (int)(e - (size_t)k * T);
    dst[(size_t)idx[k] * T + w] = src[e];
  }
}

// Ordering words of the peer-store transport (dpgo_amd/ipc.py): 64-bit epochs in device memory that TWO PROCESSES map
// (hipIpc).  A producer's stream writes a word AFTER the kernel that produced the data (stream order: that kernel's
// end-of-kernel release has made its stores visible device-wide); a consumer's stream runs k_flags_wait BEFORE the kernel
// that reads the data (whose start-of-kernel acquire then sees it).  System-scope atomics (sc0 sc1: write-through stores,
// cache-bypassing loads), so the words themselves need no kernel boundary.  The wait can be bounded (`timeout_ticks` of the
// 100 MHz wall clock): a word that does not arrive in time is reported through an error word the host polls, or traps.
constexpr int kFlagCap = 32;
struct FlagTable {
  unsigned long long* p[kFlagCap];
  unsigned long long v[kFlagCap];
  int n;
};
static __global__ void k_flags_write(FlagTable t) {
  const int i = threadIdx.x;
  if (i < t.n) __hip_atomic_store(t.p[i], t.v[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// timeout_ticks <= 0: no limit.  err (host-coherent memory, may be NULL): a word that times out is REPORTED there (1 + its
// index, system-scope store) and the kernel returns -- the host reads the word at its next exchange and raises; without an
// error word the kernel traps (the process's HIP context is lost: the round-5 behaviour, kept for dpgo_flags_wait_device).
static __global__ void k_flags_wait(FlagTable t, long long timeout_ticks, unsigned long long* err) {
  const int i = threadIdx.x;
  if (i < t.n) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(t.p[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < t.v[i]) {
      __builtin_amdgcn_s_sleep(16);
      if (timeout_ticks > 0 && wall_clock64() - t0 > timeout_ticks) {
        if (!err) __builtin_trap();
        __hip_atomic_store(err, (unsigned long long)(i + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}

// Batched form for agents that live in ONE process: message m copies cnt pose tiles src[m][idx[m][k]] -> dst[m][k]; one
// launch for a whole exchange phase (a 16-agent sweep spent 0.5 ms in ~60 tiny pack / copy launches).  first[m] = tiles of
// the messages before m (first[nmsg] = total).
struct ExchangeTable {
  const double* const* src;
  const int32_t* const* idx;
  double* const* dst;
  const int32_t* first;
  int nmsg;
};
template <int T>
__global__ __launch_bounds__(kBlock) void k_gather_tiles_batched(ExchangeTable tb) {
  const int total = tb.first[tb.nmsg];
  for (int k = blockIdx.x * (kBlock / 4) + (int)(threadIdx.x >> 2); k < total; k += gridDim.x * (kBlock / 4)) {
    int lo = 0, hi = tb.nmsg;  // message of tile k: first[lo] <= k < first[lo + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tb.first[mid] <= k) lo = mid; else hi = mid;
    }
    const int kk = k - tb.first[lo];
    const double* __restrict__ s = tb.src[lo] + (size_t)tb.idx[lo][kk] * T;
    double* __restrict__ o = tb.dst[lo] + (size_t)kk * T;
    for (int w = threadIdx.x & 3; w < T; w += 4) o[w] = s[w];
  }
}

// ================================================================ agent status: relative change of the iterate
// LiftedPoseArray::maxTranslationDistance (src/manifold/Poses.cpp:86-94) = max_i |p_i - p_i'| over the translation columns
// of two lifted pose arrays; PGOAgent::iterate stores it as mStatus.relativeChange (src/PGOAgent.cpp:406).  One thread per
// pose, wave maximum by shuffles, one atomic maximum per wave on the BIT PATTERN (non-negative doubles order like their
// bits; *out is zeroed before the launch), so the result does not depend on the order of arrival.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_max_translation_distance(const double* __restrict__ X,
                                                                     const double* __restrict__ Xp, int n,
                                                                     unsigned long long* __restrict__ out) {
  constexpr int T = (D + 1) * R;
  double m = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double* a = X + (size_t)i * T + D * R;
    const double* b = Xp + (size_t)i * T + D * R;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double dlt = a[k] - b[k];
      s = fma(dlt, dlt, s);
    }
    m = fmax(m, sqrt(s));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// Block-Jacobi factors: Dinv_i = (Q_ii + shift I)^-1 by Gauss-Jordan on the SPD (D+1)x(D+1) block.
template <int D>
__global__ __launch_bounds__(kBlock) void k_build_dinv(BsrDev Q, double shift, double* __restrict__ dinv, int n) {
  constexpr int B = D + 1;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double A[B][B], I[B][B];
    bool found = false;
    for (int t = Q.rowptr[i]; t < Q.rowptr[i + 1]; ++t) {
      if (Q.colidx[t] == i) {
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int q = 0; q < B; ++q) A[p][q] = Q.vals[(size_t)t * B * B + p * B + q];
        found = true;
      }
    }
    if (!found) {
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) A[p][q] = 0.0;
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) I[p][q] = (p == q) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < B; ++p) A[p][p] += shift;
#pragma unroll
    for (int p = 0; p < B; ++p) {
      const double inv = 1.0 / A[p][p];
#pragma unroll
      for (int q = 0; q < B; ++q) {
        A[p][q] *= inv;
        I[p][q] *= inv;
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        if (k != p) {
          const double f = A[k][p];
#pragma unroll
          for (int q = 0; q < B; ++q) {
            A[k][q] = fma(-f, A[p][q], A[k][q]);
            I[k][q] = fma(-f, I[p][q], I[k][q]);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) dinv[(size_t)i * B * B + p * B + q] = 0.5 * (I[p][q] + I[q][p]);
  }
}


// ================================================================ K10: edge residuals + GNC-TLS weights
// One lane per (re-weightable) edge e = (i -> j): squared residual of computeMeasurementError
// (reference src/DPGO_utils.cpp:501-507), rSq = kappa |Y_i R - Y_j|_F^2 + tau |p_j - p_i - Y_i t|^2, then
// RobustCost::weight for GNC_TLS (src/DPGO_robust.cpp:80-92, eq. (14) of the GNC paper) unless the edge has a
// fixed weight.  counts[0..2] = inliers (w > 1 - tol) / outliers (w < tol) / undecided among the non-fixed edges
// (integer atomics: exact and order-independent).
struct EdgeDev {
  const int32_t* p1;
  const int32_t* p2;
  const double* Rm;     // m x D x D, row-major per edge
  const double* t;      // m x D
  const double* kappa;
  const double* tau;
  const uint8_t* fixed;
  const uint8_t* role;   // 0 private, 1 shared outgoing (p1 mine, other pose = neighbour slot), 2 shared incoming
  const int32_t* slot;   // neighbour-tile slot of the other pose (roles 1, 2)
  double* weight;
  double* rsq;
  int m;
};

template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_edge_weights(EdgeDev E, const double* __restrict__ X,
                                                         const double* __restrict__ nbr, double mu, double barc,
                                                         double w_tol, int update_weights, int* __restrict__ counts) {
  constexpr int B = D + 1, T = B * R;
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < E.m; e += gridDim.x * kBlock) {
    // shared edges (PGOAgent::computeMeasurementResidual, src/PGOAgent.cpp:1048-1102): the pose owned by the
    // neighbour comes from the public-pose buffer
    const int role = E.role[e];
    const double* __restrict__ xi = (role == 2) ? nbr + (size_t)E.slot[e] * T : X + (size_t)E.p1[e] * T;
    const double* __restrict__ xj = (role == 1) ? nbr + (size_t)E.slot[e] * T : X + (size_t)E.p2[e] * T;
    const double* __restrict__ Rm = E.Rm + (size_t)e * D * D;
    const double* __restrict__ tv = E.t + (size_t)e * D;
    double rot = 0.0, tr = 0.0;
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double v = -xj[c * R + a];
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(xi[k * R + a], Rm[k * D + c], v);
        rot = fma(v, v, rot);
      }
      double u = xj[D * R + a] - xi[D * R + a];
#pragma unroll
      for (int k = 0; k < D; ++k) u = fma(-xi[k * R + a], tv[k], u);
      tr = fma(u, u, tr);
    }
    const double rSq0 = E.kappa[e] * rot + E.tau[e] * tr;
    E.rsq[e] = rSq0;
    if (!E.fixed[e]) {
      double w = E.weight[e];
      if (update_weights) {
        const double r = sqrt(rSq0), rSq = r * r, bSq = barc * barc;
        const double upper = (mu + 1.0) / mu * bSq, lower = mu / (mu + 1.0) * bSq;
        if (rSq >= upper) w = 0.0;
        else if (rSq <= lower) w = 1.0;
        else w = sqrt(bSq * mu * (mu + 1.0) / rSq) - mu;
        E.weight[e] = w;
      }
      if (counts && role != 2) {  // a shared edge is counted by the agent that owns its source pose
        if (w < w_tol) atomicAdd(&counts[1], 1);
        else if (w > 1.0 - w_tol) atomicAdd(&counts[0], 1);
        else atomicAdd(&counts[2], 1);
      }
    }
  }
}

// ================================================================ K9: rebuild the values of Q from edge weights
// Gather form of constructConnectionLaplacianSE (reference src/DPGO_utils.cpp:272-344): the block-CSR pattern
// is fixed by the edge list, GNC changes values only.  One lane per BSR slot sums, in a fixed order, the
// contributions of the edges incident to that slot (host-built lists):
//   kind 0: +T Om T^T (diagonal, source pose)   kind 1: +Om (diagonal, destination pose)
//   kind 2: -T Om (block (i,j))                 kind 3: -Om T^T (block (j,i))
// with T = [R t; 0 1], Om = w diag(kappa.., tau).  vals = base + sign * sum.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rebuild_Q(EdgeDev E, const int32_t* __restrict__ cptr,
                                                      const int32_t* __restrict__ cedge,
                                                      const uint8_t* __restrict__ ckind,
                                                      const double* __restrict__ base, double sign,
                                                      double* __restrict__ vals, int nnzb) {
  constexpr int B = D + 1, BB = B * B;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < nnzb; s += gridDim.x * kBlock) {
    double acc[BB];
#pragma unroll
    for (int q = 0; q < BB; ++q) acc[q] = 0.0;
    for (int k = cptr[s]; k < cptr[s + 1]; ++k) {
      const int e = cedge[k];
      const int kind = ckind[k];
      double Tm[B][B], om[B];
      const double w = E.weight[e];
#pragma unroll
      for (int p = 0; p < D; ++p) {
#pragma unroll
        for (int q = 0; q < D; ++q) Tm[p][q] = E.Rm[(size_t)e * D * D + p * D + q];
        Tm[p][D] = E.t[(size_t)e * D + p];
        Tm[D][p] = 0.0;
        om[p] = w * E.kappa[e];
      }
      Tm[D][D] = 1.0;
      om[D] = w * E.tau[e];
#pragma unroll
      for (int p = 0; p < B; ++p) {
#pragma unroll
        for (int q = 0; q < B; ++q) {
          double v;
          if (kind == 0) {
            v = 0.0;
#pragma unroll
            for (int kk = 0; kk < B; ++kk) v = fma(Tm[p][kk] * om[kk], Tm[q][kk], v);
          } else if (kind == 1) {
            v = (p == q) ? om[p] : 0.0;
          } else if (kind == 2) {
            v = -Tm[p][q] * om[q];
          } else {
            v = -om[p] * Tm[q][p];
          }
          acc[p * B + q] += v;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < BB; ++q) vals[(size_t)s * BB + q] = base[(size_t)s * BB + q] + sign * acc[q];
  }
}
