// spmm_lab.hip -- stand-alone experiment harness for the Q*X block-SpMM core (D = 3, R = 5).
// Builds a 3-D lattice block-CSR (same pattern statistics as the 100k-pose benchmark grid), runs
// several kernel variants, checks them against variant 0 and prints the average launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/spmm_lab.hip -o tools/spmm_lab && tools/spmm_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define HC(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int D = 3, R = 5, B = 4, T = 20, BB = 16;

struct TileIter {
  int first, last, step;
};
__device__ __forceinline__ TileIter tile_iter(int ntiles) {
  TileIter it;
  const int G = gridDim.x;
  if (G < 16 || ntiles < 16) {
    it.first = blockIdx.x; it.last = ntiles; it.step = G;
    return it;
  }
  const int x = blockIdx.x & 7, lb = blockIdx.x >> 3;
  const int nbx = (G - x + 7) >> 3;
  const int lo = (int)(((long long)ntiles * x) >> 3), hi = (int)(((long long)ntiles * (x + 1)) >> 3);
  it.first = lo + lb; it.last = hi; it.step = nbx;
  return it;
}

// ---------------- V0: current product kernel (lane = (row, column c); 16 rows per wave)
__global__ __launch_bounds__(256) void v0(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    if (i < n) {
      double acc[R] = {0, 0, 0, 0, 0};
      const int t0 = rowptr[i], t1 = rowptr[i + 1];
      for (int t = t0; t < t1; ++t) {
        const int j = colidx[t];
        const double* __restrict__ q = vals + (size_t)t * BB + c * B;
        const double* __restrict__ x = V + (size_t)j * T;
        double qk[B];
#pragma unroll
        for (int k = 0; k < B; ++k) qk[k] = q[k];
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = fma(x[k * R + a], qk[k], acc[a]);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
  }
}

// ---------------- V1: as V0, explicit double2 loads, column indices of the row preloaded, 2-block unroll
typedef double double2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_tile(const double* __restrict__ p, double (&x)[T]) {
  const double2_t* __restrict__ p2 = reinterpret_cast<const double2_t*>(p);
#pragma unroll
  for (int k = 0; k < T / 2; ++k) {
    const double2_t v = p2[k];
    x[2 * k] = v.x;
    x[2 * k + 1] = v.y;
  }
}
__device__ __forceinline__ void load_qrow(const double* __restrict__ p, double (&q)[B]) {
  const double2_t* __restrict__ p2 = reinterpret_cast<const double2_t*>(p);
  const double2_t a = p2[0], b = p2[1];
  q[0] = a.x; q[1] = a.y; q[2] = b.x; q[3] = b.y;
}
__device__ __forceinline__ void fma_tile(const double (&x)[T], const double (&q)[B], double (&acc)[R]) {
#pragma unroll
  for (int k = 0; k < B; ++k)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] = fma(x[k * R + a], q[k], acc[a]);
}

template <int MAXGRID_UNUSED>
__global__ __launch_bounds__(256) void v1(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    if (i < n) {
      double acc[R] = {0, 0, 0, 0, 0};
      const int t0 = rowptr[i], t1 = rowptr[i + 1];
      int t = t0;
      for (; t + 1 < t1; t += 2) {
        const int j0 = colidx[t], j1 = colidx[t + 1];
        double q0[B], q1[B], x0[T], x1[T];
        load_qrow(vals + (size_t)t * BB + c * B, q0);
        load_qrow(vals + (size_t)(t + 1) * BB + c * B, q1);
        load_tile(V + (size_t)j0 * T, x0);
        load_tile(V + (size_t)j1 * T, x1);
        fma_tile(x0, q0, acc);
        fma_tile(x1, q1, acc);
      }
      if (t < t1) {
        const int j0 = colidx[t];
        double q0[B], x0[T];
        load_qrow(vals + (size_t)t * BB + c * B, q0);
        load_tile(V + (size_t)j0 * T, x0);
        fma_tile(x0, q0, acc);
      }
      double2_t* o = reinterpret_cast<double2_t*>(OUT + (size_t)i * T + c * R);
      // 40-byte columns are only 8-byte aligned for odd c: scalar stores
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
      (void)o;
    }
  }
}

// ---------------- V2: thread per row (one lane = one block row); Q block and X tile private loads
__global__ __launch_bounds__(256) void v2(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double acc[T];
#pragma unroll
    for (int e = 0; e < T; ++e) acc[e] = 0.0;
    const int t0 = rowptr[i], t1 = rowptr[i + 1];
    for (int t = t0; t < t1; ++t) {
      const int j = colidx[t];
      double x[T], q[BB];
      load_tile(V + (size_t)j * T, x);
      const double2_t* __restrict__ q2 = reinterpret_cast<const double2_t*>(vals + (size_t)t * BB);
#pragma unroll
      for (int k = 0; k < BB / 2; ++k) {
        const double2_t v = q2[k];
        q[2 * k] = v.x; q[2 * k + 1] = v.y;
      }
#pragma unroll
      for (int c = 0; c < B; ++c)
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[c * R + a] = fma(x[k * R + a], q[c * B + k], acc[c * R + a]);
    }
    double2_t* o = reinterpret_cast<double2_t*>(OUT + (size_t)i * T);
#pragma unroll
    for (int k = 0; k < T / 2; ++k) {
      double2_t v; v.x = acc[2 * k]; v.y = acc[2 * k + 1];
      o[k] = v;
    }
  }
}

// ---------------- V3: 2 lanes per row (lane handles columns {2h, 2h+1}); 32 rows per wave
__global__ __launch_bounds__(256) void v3(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 1, h = l & 1;
  const int ntiles = (n + 127) / 128;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 128 + wave * 32 + g;
    if (i < n) {
      double acc0[R] = {0, 0, 0, 0, 0}, acc1[R] = {0, 0, 0, 0, 0};
      const int t0 = rowptr[i], t1 = rowptr[i + 1];
      for (int t = t0; t < t1; ++t) {
        const int j = colidx[t];
        double x[T], q0[B], q1[B];
        load_qrow(vals + (size_t)t * BB + (2 * h) * B, q0);
        load_qrow(vals + (size_t)t * BB + (2 * h + 1) * B, q1);
        load_tile(V + (size_t)j * T, x);
        fma_tile(x, q0, acc0);
        fma_tile(x, q1, acc1);
      }
      double2_t* o = reinterpret_cast<double2_t*>(OUT + (size_t)i * T + 2 * h * R);  // 80-byte offset: 16-B aligned
      double2_t v;
      v.x = acc0[0]; v.y = acc0[1]; o[0] = v;
      v.x = acc0[2]; v.y = acc0[3]; o[1] = v;
      v.x = acc0[4]; v.y = acc1[0]; o[2] = v;
      v.x = acc1[1]; v.y = acc1[2]; o[3] = v;
      v.x = acc1[3]; v.y = acc1[4]; o[4] = v;
    }
  }
}

// ---------------- V4: LDS-staged.  A workgroup owns P consecutive rows; their Q blocks are one contiguous
// span which is streamed to LDS with 16-byte lane-linear loads; the X tiles named by colidx are gathered
// piece-parallel (one 16-byte piece per lane per instruction) into LDS; compute reads LDS only.
template <int P, int MAXB>
__global__ __launch_bounds__(256) void v4(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* qs = reinterpret_cast<double*>(smem);                 // MAXB * 16 doubles
  double* xs = qs + (size_t)MAXB * BB;                          // MAXB * 20 doubles
  int* js = reinterpret_cast<int*>(xs + (size_t)MAXB * T);      // MAXB ints
  const int ntiles = (n + P - 1) / P;
  const TileIter ti = tile_iter(ntiles);
  const int tid = threadIdx.x;
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int r0 = tile * P, r1 = min(r0 + P, n);
    const int tb = rowptr[r0], te = rowptr[r1];
    const int nb = te - tb;  // assumed <= MAXB (checked on the host)
    __syncthreads();
    for (int k = tid; k < nb; k += 256) js[k] = colidx[tb + k];
    {  // Q span: nb*128 bytes contiguous, 16 B per lane
      const double2_t* __restrict__ src = reinterpret_cast<const double2_t*>(vals + (size_t)tb * BB);
      double2_t* dst = reinterpret_cast<double2_t*>(qs);
      for (int k = tid; k < nb * (BB / 2); k += 256) dst[k] = src[k];
    }
    __syncthreads();
    {  // X tiles: nb*10 pieces of 16 B
      double2_t* dst = reinterpret_cast<double2_t*>(xs);
      for (int k = tid; k < nb * (T / 2); k += 256) {
        const int b = k / (T / 2), pc = k - b * (T / 2);
        dst[k] = reinterpret_cast<const double2_t*>(V + (size_t)js[b] * T)[pc];
      }
    }
    __syncthreads();
    // compute: lane = (row, column c); 64 rows per pass of 256 threads
    for (int rr = tid >> 2; rr < r1 - r0; rr += 64) {
      const int i = r0 + rr, c = tid & 3;
      double acc[R] = {0, 0, 0, 0, 0};
      const int t0 = rowptr[i] - tb, t1 = rowptr[i + 1] - tb;
      for (int t = t0; t < t1; ++t) {
        const double* q = qs + (size_t)t * BB + c * B;
        const double* x = xs + (size_t)t * T;
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = fma(x[k * R + a], q[k], acc[a]);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
  }
}

// ---------------- V5: wave-private LDS staging of the X tiles only (no block barrier): each wave owns 16 rows,
// gathers their X tiles piece-parallel into its LDS slot, then computes with Q read straight from global.
template <int MAXB_W>
__global__ __launch_bounds__(256) void v5(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  __shared__ __attribute__((aligned(16))) double xs_all[4][MAXB_W * T];
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  double* xs = xs_all[wave];
  const int nwt = (n + 15) / 16;  // wave tiles
  const int ntiles = (nwt + 3) / 4;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int r0 = (tile * 4 + wave) * 16;
    if (r0 >= n) continue;
    const int r1 = min(r0 + 16, n);
    const int tb = rowptr[r0], te = rowptr[r1];
    const int nb = te - tb;
    {
      double2_t* dst = reinterpret_cast<double2_t*>(xs);
      for (int k = l; k < nb * (T / 2); k += 64) {
        const int b = k / (T / 2), pc = k - b * (T / 2);
        const int j = colidx[tb + b];
        dst[k] = reinterpret_cast<const double2_t*>(V + (size_t)j * T)[pc];
      }
    }
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt/lgkmcnt drain (wave-private LDS: no barrier needed)
    __builtin_amdgcn_wave_barrier();
    const int i = r0 + g;
    if (i < n) {
      double acc[R] = {0, 0, 0, 0, 0};
      const int t0 = rowptr[i], t1 = rowptr[i + 1];
      for (int t = t0; t < t1; ++t) {
        double q[B];
        load_qrow(vals + (size_t)t * BB + c * B, q);
        const double* x = xs + (size_t)(t - tb) * T;
#pragma unroll
        for (int k = 0; k < B; ++k)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = fma(x[k * R + a], q[k], acc[a]);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// ---------------- V6: SPLIT lanes-groups per row: lane = (row g, slice s, column c); slice s handles blocks
// t0+s, t0+s+SPLIT, ...; partial columns combined with SPLIT-1 xor-shuffles.  Shortens the dependent
// index->tile load chain per wave from deg to deg/SPLIT.
template <int SPLIT>
__global__ __launch_bounds__(256) void v6(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  constexpr int LPR = 4 * SPLIT, RPW = 64 / LPR, RPB = RPW * 4;
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l / LPR, s = (l / 4) % SPLIT, c = l & 3;
  const int ntiles = (n + RPB - 1) / RPB;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * RPB + wave * RPW + g;
    double acc[R] = {0, 0, 0, 0, 0};
    if (i < n) {
      const int t0 = rowptr[i], t1 = rowptr[i + 1];
      for (int t = t0 + s; t < t1; t += SPLIT) {
        const int j = colidx[t];
        double q[B], x[T];
        load_qrow(vals + (size_t)t * BB + c * B, q);
        load_tile(V + (size_t)j * T, x);
        fma_tile(x, q, acc);
      }
    }
#pragma unroll
    for (int o = 4; o < LPR; o <<= 1)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] += __shfl_xor(acc[a], o);
    if (i < n && s == 0) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
  }
}

// ---------------- V7: V0 with the row's column indices preloaded by the quad (2 per lane, rows up to 8 blocks;
// longer rows fall back to the plain loop for the tail)
__global__ __launch_bounds__(256) void v7(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    const bool ok = i < n;
    const int t0 = ok ? rowptr[i] : 0, t1 = ok ? rowptr[i + 1] : 0;
    const int deg = t1 - t0;
    const int ja = (c < deg) ? colidx[t0 + c] : 0;
    const int jb = (c + 4 < deg) ? colidx[t0 + c + 4] : 0;
    double acc[R] = {0, 0, 0, 0, 0};
    int maxdeg = deg;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
    const int lim = maxdeg < 8 ? maxdeg : 8;
    for (int k = 0; k < lim; ++k) {
      const int j = __shfl((k < 4) ? ja : jb, (l & ~3) | (k & 3));
      if (k < deg) {
        double q[B], x[T];
        load_qrow(vals + (size_t)(t0 + k) * BB + c * B, q);
        load_tile(V + (size_t)j * T, x);
        fma_tile(x, q, acc);
      }
    }
    for (int t = t0 + 8; t < t1; ++t) {
      const int j = colidx[t];
      double q[B], x[T];
      load_qrow(vals + (size_t)t * BB + c * B, q);
      load_tile(V + (size_t)j * T, x);
      fma_tile(x, q, acc);
    }
    if (ok) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
  }
}

// ---------------- V8: block-parallel.  A wave owns RW consecutive rows = one contiguous span of blocks; each
// quad of lanes takes ONE block per pass (16 blocks per wave-pass: Q rows coalesced, 16 independent tile
// gathers in flight), writes its partial column to a wave-private LDS slot; afterwards lane (row, c) sums
// the partials of its row (segmented reduce through LDS, fixed order => deterministic).
template <int RW, int MAXB_W>
__global__ __launch_bounds__(256) void v8(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                          const double* __restrict__ vals, const double* __restrict__ V,
                                          double* __restrict__ OUT, int n) {
  __shared__ double ps_all[4][MAXB_W * T];
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  double* ps = ps_all[wave];
  const int nwt = (n + RW - 1) / RW;
  const int ntiles = (nwt + 3) / 4;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int r0 = (tile * 4 + wave) * RW;
    if (r0 < n) {
      const int r1 = min(r0 + RW, n);
      const int tb = rowptr[r0], te = rowptr[r1];
      for (int base = tb; base < te; base += MAXB_W) {
        const int lim = min(te, base + MAXB_W);
#pragma unroll 2
        for (int t = base + g; t < lim; t += 16) {
          const int j = colidx[t];
          double q[B], x[T], acc[R] = {0, 0, 0, 0, 0};
          load_qrow(vals + (size_t)t * BB + c * B, q);
          load_tile(V + (size_t)j * T, x);
          fma_tile(x, q, acc);
          double* dst = ps + (size_t)(t - base) * T + c * R;
#pragma unroll
          for (int a = 0; a < R; ++a) dst[a] = acc[a];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        // segmented reduce: RW rows, 4 lanes each -> RW*4 lanes busy (RW = 16 => all 64)
        for (int rr = g; rr < r1 - r0; rr += 16) {
          const int i = r0 + rr;
          const int a0 = max(rowptr[i], base), a1 = min(rowptr[i + 1], lim);
          double acc[R] = {0, 0, 0, 0, 0};
          for (int t = a0; t < a1; ++t) {
            const double* src = ps + (size_t)(t - base) * T + c * R;
#pragma unroll
            for (int a = 0; a < R; ++a) acc[a] += src[a];
          }
          if (a1 > a0) {
            double* o = OUT + (size_t)i * T + c * R;
            if (base == tb || rowptr[i] >= base) {
#pragma unroll
              for (int a = 0; a < R; ++a) o[a] = acc[a];
            } else {
#pragma unroll
              for (int a = 0; a < R; ++a) o[a] += acc[a];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ---------------- V9: symmetric storage.  Only blocks (i, j >= i) are stored; row i additionally walks a list of
// references (j < i, slot of block (j, i)) and uses those blocks TRANSPOSED (column c instead of row c).  Every
// block is touched twice; the bet is that the second touch is an L2 hit, so HBM sees ~half of Q.
__global__ __launch_bounds__(256) void v9(const int* __restrict__ rpu, const int* __restrict__ ciu,
                                          const double* __restrict__ valsu, const int* __restrict__ rpl,
                                          const int* __restrict__ cil, const int* __restrict__ slotl,
                                          const double* __restrict__ V, double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    if (i < n) {
      double acc[R] = {0, 0, 0, 0, 0};
      for (int t = rpu[i]; t < rpu[i + 1]; ++t) {
        const int j = ciu[t];
        double q[B], x[T];
        load_qrow(valsu + (size_t)t * BB + c * B, q);
        load_tile(V + (size_t)j * T, x);
        fma_tile(x, q, acc);
      }
      for (int t = rpl[i]; t < rpl[i + 1]; ++t) {
        const int j = cil[t];
        const double* __restrict__ blk = valsu + (size_t)slotl[t] * BB;
        double q[B], x[T];
#pragma unroll
        for (int k = 0; k < B; ++k) q[k] = blk[k * B + c];
        load_tile(V + (size_t)j * T, x);
        fma_tile(x, q, acc);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = acc[a];
    }
  }
}

// ---------------- V11: outer-product accumulation.  Lane c of a quad loads only column c of the gathered tile
// (40 B) and column c of the Q block (32 B, blocks stored transposed), accumulates the 5x4 partial
// P_c[a][c'] = X_j[a][c] * Q_ij[c'][c]; ONE quad reduction per row (not per block) yields OUT_i.  Load
// instructions per 16 blocks: 2 (Q) + 5 x 8-byte (tile column) instead of 2 + 10 x 16-byte.
template <bool PRELOAD>
__global__ __launch_bounds__(256) void v11(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                           const double* __restrict__ valsT, const double* __restrict__ V,
                                           double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    const bool ok = i < n;
    double acc[B][R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = 0.0;
    const int t0 = ok ? rowptr[i] : 0, t1 = ok ? rowptr[i + 1] : 0;
    if (PRELOAD) {
      const int deg = t1 - t0;
      const int ja = (c < deg) ? colidx[t0 + c] : 0;
      const int jb = (c + 4 < deg) ? colidx[t0 + c + 4] : 0;
      int maxdeg = deg;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
      const int lim = maxdeg < 8 ? maxdeg : 8;
      for (int k = 0; k < lim; ++k) {
        const int j = __shfl((k < 4) ? ja : jb, (l & ~3) | (k & 3));
        if (k < deg) {
          double q[B], xc[R];
          load_qrow(valsT + (size_t)(t0 + k) * BB + c * B, q);
#pragma unroll
          for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
          for (int cc = 0; cc < B; ++cc)
#pragma unroll
            for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
        }
      }
      for (int t = t0 + 8; t < t1; ++t) {
        const int j = colidx[t];
        double q[B], xc[R];
        load_qrow(valsT + (size_t)t * BB + c * B, q);
#pragma unroll
        for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
      }
    } else {
      for (int t = t0; t < t1; ++t) {
        const int j = colidx[t];
        double q[B], xc[R];
        load_qrow(valsT + (size_t)t * BB + c * B, q);
#pragma unroll
        for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
      }
    }
    // quad reduction: every lane gets the full sums, lane c keeps column c
    double out[R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) {
        double v = acc[cc][a];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (cc == c) out[a] = v;
      }
    if (ok) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = out[a];
    }
  }
}

// ---------------- V12: V11 (outer-product, index preload) on symmetric storage: upper blocks stored transposed
// (column c contiguous); lower references use the stored block of (j, i) through its rows (strided 8-byte loads).
__global__ __launch_bounds__(256) void v12(const int* __restrict__ rpu, const int* __restrict__ ciu,
                                           const double* __restrict__ valsuT, const int* __restrict__ rpl,
                                           const int* __restrict__ cil, const int* __restrict__ slotl,
                                           const double* __restrict__ V, double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    const bool ok = i < n;
    double acc[B][R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = 0.0;
    const int u0 = ok ? rpu[i] : 0, u1 = ok ? rpu[i + 1] : 0, l0 = ok ? rpl[i] : 0, l1 = ok ? rpl[i + 1] : 0;
    const int du = u1 - u0, dl = l1 - l0;
    const int ju = (c < du) ? ciu[u0 + c] : 0;
    const int jl = (c < dl) ? cil[l0 + c] : 0;
    const int sl = (c < dl) ? slotl[l0 + c] : 0;
    int mu = du, ml = dl;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mu = max(mu, __shfl_xor(mu, o)); ml = max(ml, __shfl_xor(ml, o)); }
    const int lu = mu < 4 ? mu : 4, ll = ml < 4 ? ml : 4;
    for (int k = 0; k < lu; ++k) {
      const int j = __shfl(ju, (l & ~3) | k);
      if (k < du) {
        double q[B], xc[R];
        load_qrow(valsuT + (size_t)(u0 + k) * BB + c * B, q);
#pragma unroll
        for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
      }
    }
    for (int t = u0 + 4; t < u1; ++t) {
      const int j = ciu[t];
      double q[B], xc[R];
      load_qrow(valsuT + (size_t)t * BB + c * B, q);
#pragma unroll
      for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
      for (int cc = 0; cc < B; ++cc)
#pragma unroll
        for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
    }
    for (int k = 0; k < ll; ++k) {
      const int j = __shfl(jl, (l & ~3) | k);
      const int sb = __shfl(sl, (l & ~3) | k);
      if (k < dl) {
        const double* __restrict__ blk = valsuT + (size_t)sb * BB;  // transposed storage of Q[j,i]: blk[p*4+q] = Q[j,i][q][p]
        double q[B], xc[R];
        // need column c of Q[i,j] = row c of Q[j,i] = blk[p*4 + c], p = 0..3
#pragma unroll
        for (int pp = 0; pp < B; ++pp) q[pp] = blk[pp * B + c];
#pragma unroll
        for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
      }
    }
    for (int t = l0 + 4; t < l1; ++t) {
      const int j = cil[t];
      const double* __restrict__ blk = valsuT + (size_t)slotl[t] * BB;
      double q[B], xc[R];
#pragma unroll
      for (int pp = 0; pp < B; ++pp) q[pp] = blk[pp * B + c];
#pragma unroll
      for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
      for (int cc = 0; cc < B; ++cc)
#pragma unroll
        for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
    }
    double out[R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) {
        double v = acc[cc][a];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (cc == c) out[a] = v;
      }
    if (ok) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = out[a];
    }
  }
}

// ---------------- V13: V11 + batched issue.  The loads of NB consecutive blocks of the row (column indices are
// preloaded) are all issued before the first FMA, so a row costs ceil(deg/NB) memory round trips instead of deg.
template <int NB>
__global__ __launch_bounds__(256) void v13(const int* __restrict__ rowptr, const int* __restrict__ colidx,
                                           const double* __restrict__ valsT, const double* __restrict__ V,
                                           double* __restrict__ OUT, int n) {
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6, g = l >> 2, c = l & 3;
  const int ntiles = (n + 63) / 64;
  const TileIter ti = tile_iter(ntiles);
  for (int tile = ti.first; tile < ti.last; tile += ti.step) {
    const int i = tile * 64 + wave * 16 + g;
    const bool ok = i < n;
    double acc[B][R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = 0.0;
    const int t0 = ok ? rowptr[i] : 0, t1 = ok ? rowptr[i + 1] : 0;
    const int deg = t1 - t0;
    const int ja = (c < deg) ? colidx[t0 + c] : 0;
    const int jb = (c + 4 < deg) ? colidx[t0 + c + 4] : 0;
    int maxdeg = deg;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
    const int lim = maxdeg < 8 ? maxdeg : 8;
    for (int k0 = 0; k0 < lim; k0 += NB) {
      double q[NB][B], xc[NB][R];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int k = k0 + u;
        const int j = __shfl((k < 4) ? ja : jb, (l & ~3) | (k & 3));
        const bool on = k < deg;
        const size_t tq = on ? (size_t)(t0 + k) : (size_t)0;
        const size_t jj = on ? (size_t)j : (size_t)0;
        load_qrow(valsT + tq * BB + c * B, q[u]);
#pragma unroll
        for (int a = 0; a < R; ++a) xc[u][a] = V[jj * T + c * R + a];
        if (!on) {
#pragma unroll
          for (int cc = 0; cc < B; ++cc) q[u][cc] = 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[u][a], q[u][cc], acc[cc][a]);
    }
    for (int t = t0 + 8; t < t1; ++t) {
      const int j = colidx[t];
      double q[B], xc[R];
      load_qrow(valsT + (size_t)t * BB + c * B, q);
#pragma unroll
      for (int a = 0; a < R; ++a) xc[a] = V[(size_t)j * T + c * R + a];
#pragma unroll
      for (int cc = 0; cc < B; ++cc)
#pragma unroll
        for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
    }
    double out[R];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) {
        double v = acc[cc][a];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (cc == c) out[a] = v;
      }
    if (ok) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)i * T + c * R + a] = out[a];
    }
  }
}

__global__ __launch_bounds__(256) void k_scrub(double* __restrict__ x, size_t n) {
  // read-only: a scrub that wrote would leave dirty lines whose write-back the timed kernel then pays for
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == 1.2345e300) x[0] = acc;
}

int main(int argc, char** argv) {
  const int nx = 50, ny = 50, nz = 40, n = nx * ny * nz;
  std::vector<int> rowptr(n + 1, 0), colidx;
  auto id = [&](int x, int y, int z) { return x + nx * (y + ny * z); };
  for (int z = 0; z < nz; ++z)
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x) {
        std::vector<int> nb;
        nb.push_back(id(x, y, z));
        if (x > 0) nb.push_back(id(x - 1, y, z));
        if (x < nx - 1) nb.push_back(id(x + 1, y, z));
        if (y > 0) nb.push_back(id(x, y - 1, z));
        if (y < ny - 1) nb.push_back(id(x, y + 1, z));
        if (z > 0) nb.push_back(id(x, y, z - 1));
        if (z < nz - 1) nb.push_back(id(x, y, z + 1));
        std::sort(nb.begin(), nb.end());
        for (int j : nb) colidx.push_back(j);
        rowptr[id(x, y, z) + 1] = (int)colidx.size();
      }
  const int nnzb = (int)colidx.size();
  std::vector<double> vals((size_t)nnzb * BB), Vh((size_t)n * T);
  srand(1);
  for (auto& v : vals) v = (rand() / (double)RAND_MAX) - 0.5;
  for (auto& v : Vh) v = (rand() / (double)RAND_MAX) - 0.5;
  // make Q symmetric: block (j,i) = block (i,j)^T, diagonal blocks symmetric
  {
    auto slot = [&](int i, int j) { for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) if (colidx[t] == j) return t; return -1; };
    for (int i = 0; i < n; ++i)
      for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
        const int j = colidx[t];
        if (j > i) { const int u = slot(j, i); for (int p = 0; p < B; ++p) for (int q = 0; q < B; ++q) vals[(size_t)u * BB + q * B + p] = vals[(size_t)t * BB + p * B + q]; }
        if (j == i) for (int p = 0; p < B; ++p) for (int q = p + 1; q < B; ++q) vals[(size_t)t * BB + q * B + p] = vals[(size_t)t * BB + p * B + q];
      }
  }
  const double bytes = (double)nnzb * (8 * BB + 4) + 4.0 * (n + 1) + 16.0 * R * B * n;
  printf("n=%d nnzb=%d algorithmic bytes=%.1f MB\n", n, nnzb, bytes / 1e6);

  int *d_rp, *d_ci;
  double *d_vals, *d_V, *d_O, *d_ref;
  HC(hipMalloc(&d_rp, sizeof(int) * (n + 1)));
  HC(hipMalloc(&d_ci, sizeof(int) * nnzb));
  HC(hipMalloc(&d_vals, sizeof(double) * vals.size()));
  HC(hipMalloc(&d_V, sizeof(double) * Vh.size()));
  HC(hipMalloc(&d_O, sizeof(double) * Vh.size()));
  HC(hipMalloc(&d_ref, sizeof(double) * Vh.size()));
  HC(hipMemcpy(d_rp, rowptr.data(), sizeof(int) * (n + 1), hipMemcpyHostToDevice));
  HC(hipMemcpy(d_ci, colidx.data(), sizeof(int) * nnzb, hipMemcpyHostToDevice));
  HC(hipMemcpy(d_vals, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice));
  HC(hipMemcpy(d_V, Vh.data(), sizeof(double) * Vh.size(), hipMemcpyHostToDevice));

  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  std::vector<double> ref(Vh.size()), out(Vh.size());
  // "cold" mode: a 640 MB scrub between launches evicts the operands from the 256 MB Infinity Cache and every
  // launch is timed on its own (warm mode = back-to-back launches, operands stay on die)
  const bool cold = argc > 1 && std::string(argv[1]) == "cold";
  double* d_scrub = nullptr;
  const size_t scrub_n = (640ull << 20) / 8;
  if (cold) HC(hipMalloc(&d_scrub, scrub_n * 8));
  if (cold) HC(hipMemset(d_scrub, 0, scrub_n * 8));
  auto run = [&](const char* name, auto launch, bool is_ref) {
    HC(hipMemset(d_O, 0, sizeof(double) * Vh.size()));
    for (int i = 0; i < 5; ++i) launch();
    float ms;
    const int reps = cold ? 20 : 100;
    if (cold) {
      ms = 0.f;
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(k_scrub, dim3(4096), dim3(256), 0, 0, d_scrub, scrub_n);
        HC(hipEventRecord(e0));
        launch();
        HC(hipEventRecord(e1));
        HC(hipEventSynchronize(e1));
        float t;
        HC(hipEventElapsedTime(&t, e0, e1));
        ms += t;
      }
    } else {
      HC(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      HC(hipEventRecord(e1));
      HC(hipEventSynchronize(e1));
      HC(hipEventElapsedTime(&ms, e0, e1));
    }
    HC(hipGetLastError());
    HC(hipMemcpy(out.data(), d_O, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
    double err = 0;
    if (is_ref) ref = out;
    for (size_t k = 0; k < out.size(); ++k) err = std::max(err, std::fabs(out[k] - ref[k]));
    const double us = ms * 1e3 / reps;
    printf("%-34s %8.2f us  %7.1f GB/s  %5.1f%% of 8TB/s  maxerr %.2e\n", name, us, bytes / us / 1e3,
           bytes / us / 1e3 / 80.0, err);
  };
  const int nt64 = (n + 63) / 64;
  for (int grid : {1024, 2048, nt64}) {
    char nm[64];
    snprintf(nm, 64, "v0 lane=(row,col) grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL(v0, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, grid == 1024);
    snprintf(nm, 64, "v1 +double2,unroll2 grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL(v1<0>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  }
  for (int grid : {256, 391, 1024})
    run(grid == 256 ? "v2 thread-per-row grid=256" : (grid == 391 ? "v2 thread-per-row grid=391" : "v2 thread-per-row grid=1024"),
        [&] { hipLaunchKernelGGL(v2, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  for (int grid : {782, 1024})
    run(grid == 782 ? "v3 2 lanes/row grid=782" : "v3 2 lanes/row grid=1024",
        [&] { hipLaunchKernelGGL(v3, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  {
    constexpr int P = 64, MAXB = 64 * 7;
    const size_t lds = (size_t)MAXB * (BB + T) * 8 + MAXB * 4;
    HC(hipFuncSetAttribute(reinterpret_cast<const void*>(&v4<P, MAXB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int grid : {512, 768, 1024, nt64}) {
      char nm[64];
      snprintf(nm, 64, "v4 LDS-staged P=64 grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL((v4<P, MAXB>), dim3(grid), dim3(256), lds, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
    }
  }
  {
    constexpr int P = 32, MAXB = 32 * 7;
    const size_t lds = (size_t)MAXB * (BB + T) * 8 + MAXB * 4;
    const int nt = (n + P - 1) / P;
    for (int grid : {1024, 2048, nt}) {
      char nm[64];
      snprintf(nm, 64, "v4 LDS-staged P=32 grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL((v4<P, MAXB>), dim3(grid), dim3(256), lds, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
    }
  }
  for (int grid : {1024, 1563, 2048}) {
    char nm[64];
    snprintf(nm, 64, "v5 wave-LDS X gather grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL((v5<16 * 7>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  }
  for (int split : {2, 4}) {
    for (int mult : {1, 2}) {
      const int rpb = (64 / (4 * split)) * 4;
      const int nt = (n + rpb - 1) / rpb;
      const int grid = std::min(nt, 1024 * mult);
      char nm[64];
      snprintf(nm, 64, "v6 split=%d grid=%d", split, grid);
      if (split == 2) run(nm, [&] { hipLaunchKernelGGL((v6<2>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
      else run(nm, [&] { hipLaunchKernelGGL((v6<4>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
    }
  }
  for (int grid : {1024, 1563, 2048}) {
    char nm[64];
    snprintf(nm, 64, "v7 idx-preload grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL(v7, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  }
  for (int grid : {512, 1024, 1563}) {
    char nm[64];
    snprintf(nm, 64, "v8 block-par RW=16 MAXB=112 grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL((v8<16, 112>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
    snprintf(nm, 64, "v8 block-par RW=16 MAXB=64 grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL((v8<16, 64>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  }
  for (int grid : {512, 782}) {
    char nm[64];
    snprintf(nm, 64, "v8 block-par RW=32 MAXB=224 grid=%d", grid);
    run(nm, [&] { hipLaunchKernelGGL((v8<32, 224>), dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vals, d_V, d_O, n); }, false);
  }
  {  // symmetric storage arrays
    std::vector<int> rpu(n + 1, 0), ciu, rpl(n + 1, 0), cil, slotl;
    std::vector<double> valsu;
    std::vector<int> slot_of_full(nnzb, -1);
    for (int i = 0; i < n; ++i) {
      for (int t = rowptr[i]; t < rowptr[i + 1]; ++t)
        if (colidx[t] >= i) { slot_of_full[t] = (int)ciu.size(); ciu.push_back(colidx[t]); for (int q = 0; q < BB; ++q) valsu.push_back(vals[(size_t)t * BB + q]); }
      rpu[i + 1] = (int)ciu.size();
    }
    for (int i = 0; i < n; ++i) {
      for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
        const int j = colidx[t];
        if (j < i) { int u = -1; for (int tt = rowptr[j]; tt < rowptr[j + 1]; ++tt) if (colidx[tt] == i) u = slot_of_full[tt]; cil.push_back(j); slotl.push_back(u); }
      }
      rpl[i + 1] = (int)cil.size();
    }
    int *d_rpu, *d_ciu, *d_rpl, *d_cil, *d_sl; double* d_vu;
    HC(hipMalloc(&d_rpu, 4 * (n + 1))); HC(hipMalloc(&d_ciu, 4 * ciu.size())); HC(hipMalloc(&d_rpl, 4 * (n + 1)));
    HC(hipMalloc(&d_cil, 4 * cil.size())); HC(hipMalloc(&d_sl, 4 * slotl.size())); HC(hipMalloc(&d_vu, 8 * valsu.size()));
    HC(hipMemcpy(d_rpu, rpu.data(), 4 * (n + 1), hipMemcpyHostToDevice)); HC(hipMemcpy(d_ciu, ciu.data(), 4 * ciu.size(), hipMemcpyHostToDevice));
    HC(hipMemcpy(d_rpl, rpl.data(), 4 * (n + 1), hipMemcpyHostToDevice)); HC(hipMemcpy(d_cil, cil.data(), 4 * cil.size(), hipMemcpyHostToDevice));
    HC(hipMemcpy(d_sl, slotl.data(), 4 * slotl.size(), hipMemcpyHostToDevice)); HC(hipMemcpy(d_vu, valsu.data(), 8 * valsu.size(), hipMemcpyHostToDevice));
    printf("symmetric storage: %zu upper blocks (%.1f MB) + %zu lower refs\n", ciu.size(), valsu.size() * 8 / 1e6, cil.size());
    for (int grid : {1024, 1563, 2048}) {
      char nm[64];
      snprintf(nm, 64, "v9 symmetric storage grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v9, dim3(grid), dim3(256), 0, 0, d_rpu, d_ciu, d_vu, d_rpl, d_cil, d_sl, d_V, d_O, n); }, false);
    }
    {
      std::vector<double> vuT(valsu.size());
      for (size_t t = 0; t < ciu.size(); ++t) for (int p = 0; p < B; ++p) for (int q = 0; q < B; ++q) vuT[t * BB + q * B + p] = valsu[t * BB + p * B + q];
      double* d_vuT; HC(hipMalloc(&d_vuT, 8 * vuT.size())); HC(hipMemcpy(d_vuT, vuT.data(), 8 * vuT.size(), hipMemcpyHostToDevice));
      for (int grid : {1024, 1563, 2048}) {
        char nm[64];
        snprintf(nm, 64, "v12 outer-product symmetric grid=%d", grid);
        run(nm, [&] { hipLaunchKernelGGL(v12, dim3(grid), dim3(256), 0, 0, d_rpu, d_ciu, d_vuT, d_rpl, d_cil, d_sl, d_V, d_O, n); }, false);
      }
    }
  }
  {  // transposed blocks for v11
    std::vector<double> vT(vals.size());
    for (int t = 0; t < nnzb; ++t) for (int p = 0; p < B; ++p) for (int q = 0; q < B; ++q) vT[(size_t)t * BB + q * B + p] = vals[(size_t)t * BB + p * B + q];
    double* d_vT; HC(hipMalloc(&d_vT, 8 * vT.size())); HC(hipMemcpy(d_vT, vT.data(), 8 * vT.size(), hipMemcpyHostToDevice));
    for (int grid : {1024, 1563, 2048}) {
      char nm[64];
      snprintf(nm, 64, "v11 outer-product grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v11<false>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vT, d_V, d_O, n); }, false);
      snprintf(nm, 64, "v11 outer-product+preload grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v11<true>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vT, d_V, d_O, n); }, false);
      snprintf(nm, 64, "v13 batched x2 grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v13<2>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vT, d_V, d_O, n); }, false);
      snprintf(nm, 64, "v13 batched x4 grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v13<4>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vT, d_V, d_O, n); }, false);
      snprintf(nm, 64, "v13 batched x8 grid=%d", grid);
      run(nm, [&] { hipLaunchKernelGGL(v13<8>, dim3(grid), dim3(256), 0, 0, d_rp, d_ci, d_vT, d_V, d_O, n); }, false);
    }
  }
  return 0;
}
