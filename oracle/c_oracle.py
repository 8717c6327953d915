"""ctypes binding of oracle/dpgo_oracle_c.c (plain-C restatement of the local solve).

TEST INFRASTRUCTURE, NOT PRODUCT -- see the header of dpgo_oracle_c.c.  Built by `make -C oracle`
(__graft_entry__.build() does it); `load()` builds it on demand when gcc is available."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libdpgo_oracle_c.so")


class CParams(C.Structure):
    _fields_ = [("method", C.c_int), ("gradnorm_tol", C.c_double), ("RGD_stepsize", C.c_double),
                ("RGD_use_preconditioner", C.c_int), ("RTR_iterations", C.c_int), ("RTR_tCG_iterations", C.c_int),
                ("RTR_initial_radius", C.c_double), ("precond", C.c_int), ("precond_shift", C.c_double),
                ("accept_tiny_decrease", C.c_int), ("hess_recurrence", C.c_int)]


class CResult(C.Structure):
    _fields_ = [("success", C.c_int), ("fInit", C.c_double), ("gradNormInit", C.c_double), ("fOpt", C.c_double),
                ("gradNormOpt", C.c_double), ("tCGStatus", C.c_int), ("tcg_iterations", C.c_int),
                ("rtr_iterations", C.c_int), ("n_spmm", C.c_int)]


_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "dpgo_oracle_c.c")
        if not os.path.exists(SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(SO)):
            subprocess.check_call(["make", "-C", HERE, "-s"])
        _lib = C.CDLL(SO)
        P, I, D = C.c_void_p, C.c_int, C.c_double
        _lib.dpgo_c_spmm.argtypes = [I, I, I, P, P, P, P, P, I]
        _lib.dpgo_c_eval.argtypes = [I, I, I, P, P, P, P, P, P, D, C.POINTER(D), C.POINTER(D), P, P, P]
        _lib.dpgo_c_optimize.argtypes = [I, I, I, P, P, P, P, C.POINTER(CParams), P, P, C.POINTER(CResult)]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bsr(Q):
    return (np.ascontiguousarray(Q.rowptr, dtype=np.int32), np.ascontiguousarray(Q.colidx, dtype=np.int32),
            np.ascontiguousarray(Q.vals, dtype=np.float64))


def spmm(Q, X, reps: int = 1):
    """X Q for tiles X [n, d+1, r]; Q = oracle BSR."""
    n, b, r = X.shape
    rp, ci, v = _bsr(Q)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    out = np.empty_like(Xc)
    assert load().dpgo_c_spmm(n, b - 1, r, _p(rp), _p(ci), _p(v), _p(Xc), _p(out), reps) == 0
    return out


def evaluate(Q, G, X, V=None, shift: float = 0.1):
    """(f, |rgrad|, rgrad, Hess[V] or None, Precond[V] or None) at X (block-Jacobi preconditioner)."""
    n, b, r = X.shape
    rp, ci, v = _bsr(Q)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    Gc = None if G is None else np.ascontiguousarray(G, dtype=np.float64)
    Vc = None if V is None else np.ascontiguousarray(V, dtype=np.float64)
    f, gn = C.c_double(0.0), C.c_double(0.0)
    RG = np.empty_like(Xc)
    HV = None if V is None else np.empty_like(Xc)
    PV = None if V is None else np.empty_like(Xc)
    rc = load().dpgo_c_eval(n, b - 1, r, _p(rp), _p(ci), _p(v), _p(Gc), _p(Xc), _p(Vc), shift, C.byref(f), C.byref(gn),
                            _p(RG), _p(HV), _p(PV))
    assert rc == 0, rc
    return f.value, gn.value, RG, HV, PV


def optimize(Q, G, X0, gradnorm_tol=1e-2, RTR_iterations=3, RTR_tCG_iterations=50, RTR_initial_radius=100.0,
             method="RTR", RGD_stepsize=1e-3, RGD_use_preconditioner=True, precond="jacobi", shift=0.1,
             accept_tiny_decrease=True, hess_recurrence=False, tiled_sums=False):
    """QuadraticOptimizer::optimize restated in C.  Returns (Xopt tiles, CResult).  tiled_sums: every full-vector sum
    is formed tile-wise (64 poses per partial, partials in order) -- a second summation order of the same arithmetic."""
    n, b, r = X0.shape
    rp, ci, v = _bsr(Q)
    Xc = np.ascontiguousarray(X0, dtype=np.float64)
    Gc = None if G is None else np.ascontiguousarray(G, dtype=np.float64)
    prm = CParams(0 if method == "RTR" else 1, gradnorm_tol, RGD_stepsize, int(RGD_use_preconditioner), RTR_iterations,
                  RTR_tCG_iterations, RTR_initial_radius, {"none": 0, "jacobi": 1}[precond], shift,
                  int(accept_tiny_decrease), int(bool(hess_recurrence)) | (2 if tiled_sums else 0))
    res = CResult()
    out = np.empty_like(Xc)
    rc = load().dpgo_c_optimize(n, b - 1, r, _p(rp), _p(ci), _p(v), _p(Gc), C.byref(prm), _p(Xc), _p(out), C.byref(res))
    assert rc == 0, rc
    return out, res
