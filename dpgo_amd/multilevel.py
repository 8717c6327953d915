"""Host-side setup of the optional two-level (aggregation multigrid) preconditioner.

The reference preconditions tCG with an exact CHOLMOD solve of Q + 0.1 I (src/PoseGraph.cpp:598-613,
src/QuadraticProblem.cpp:56-69), built on the host once per Q.  This is the analogous host step for the device's
`precond = "multilevel"`: aggregates of k consecutive poses, prolongation blocks = relative poses composed along the
odometry chain (read off Q's own blocks), dense inverse of the Galerkin coarse operator.  The per-iteration work
(smoothing, restriction, coarse solve, prolongation) runs on the device.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

MAX_COARSE_UNKNOWNS = 3200  # dense inverse <= 82 MB in fp64: stays in the 256 MB Infinity Cache


def default_aggregate_size(n: int, b: int, max_coarse: int = MAX_COARSE_UNKNOWNS) -> int:
    """Smallest power of two >= 4 such that the coarse operator has <= max_coarse unknowns."""
    k = 4
    while ((n + k - 1) // k) * b > max_coarse:
        k *= 2
    # aggregates larger than 16 poses straddle the workgroup tiles of the fused cycle (16 poses in 3-D): prefer
    # k = 16 with a larger coarse operator (<= 8192 unknowns, 268 MB in fp32) over the 9-launch fallback
    if k > 16 and ((n + 15) // 16) * b <= 8192:
        k = 16
    return k


def prolongation_blocks(rowptr, colidx, vals, d: int, k: int) -> np.ndarray:
    """Pb[i] = G(root -> i)^T, root = (i // k) k.  For an edge i -> i+1 the block Q_{i,i+1} = -T Om =
    -[w kappa R, w tau t; 0, w tau] (src/DPGO_utils.cpp:307-329) gives back T = [R t; 0 1]; where the chain is broken
    (no such block) it restarts at the identity."""
    n, b = len(rowptr) - 1, d + 1
    Pb = np.zeros((n, b, b))
    G = np.eye(b)
    for i in range(n):
        if i % k == 0:
            G = np.eye(b)
        else:
            blk = None
            for t in range(rowptr[i - 1], rowptr[i]):
                if colidx[t] == i:
                    blk = vals[t]
            ok = blk is not None and -blk[d, d] > 0
            wk = np.linalg.norm(blk[:d, 0]) if ok else 0.0
            if ok and wk > 0:
                T = np.eye(b)
                T[:d, :d] = -blk[:d, :d] / wk
                T[:d, d] = -blk[:d, d] / (-blk[d, d])
                G = G @ T
            else:
                G = np.eye(b)
        Pb[i] = G.T
    return Pb


def build(rowptr, colidx, vals, d: int, shift: float = 1e-1, k: int = 0):
    """Returns (k, P_blocks [n, b, b], AcInv [N, N]) for Q given as block-CSR (vals [nnzb, b, b])."""
    n, b = len(rowptr) - 1, d + 1
    k = int(k) if k else default_aggregate_size(n, b)
    Pb = prolongation_blocks(rowptr, colidx, vals, d, k)
    nc = (n + k - 1) // k
    rows = np.broadcast_to(np.arange(n)[:, None, None] * b + np.arange(b)[None, :, None], (n, b, b))
    cols = np.broadcast_to((np.arange(n) // k)[:, None, None] * b + np.arange(b)[None, None, :], (n, b, b))
    P = sp.csr_matrix((Pb.ravel(), (rows.ravel(), cols.ravel())), shape=(n * b, nc * b))
    A = sp.bsr_matrix((vals, colidx, rowptr), shape=(n * b, n * b)).tocsr() + shift * sp.identity(n * b, format="csr")
    Ac = (P.T @ A @ P).toarray()
    Ac = 0.5 * (Ac + Ac.T)
    return k, np.ascontiguousarray(Pb), np.ascontiguousarray(np.linalg.inv(Ac))
