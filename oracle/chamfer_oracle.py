"""ctypes/numpy front end of oracle/chamfer_oracle.c.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def chamfer_forward(xyz1, xyz2):
    """xyz1 [B,n,3], xyz2 [B,m,3] -> dist1[B,n], dist2[B,m] f32, idx1, idx2 i32."""
    a, b = _f(xyz1), _f(xyz2)
    B, n, _ = a.shape
    m = b.shape[1]
    d1 = np.empty((B, n), np.float32); i1 = np.empty((B, n), np.int32)
    d2 = np.empty((B, m), np.float32); i2 = np.empty((B, m), np.int32)
    lib().chamfer_oracle_forward(B, n, _p(a), m, _p(b), _p(d1), _p(i1), _p(d2), _p(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, gd1, gd2, idx1, idx2):
    a, b = _f(xyz1), _f(xyz2)
    B, n, _ = a.shape
    m = b.shape[1]
    g1 = np.empty_like(a); g2 = np.empty_like(b)
    i1 = np.ascontiguousarray(idx1, np.int32); i2 = np.ascontiguousarray(idx2, np.int32)
    lib().chamfer_oracle_backward(B, n, _p(a), m, _p(b), _p(_f(gd1)), _p(i1), _p(_f(gd2)), _p(i2), _p(g1), _p(g2))
    return g1, g2


def emd_approxmatch(xyz1, xyz2):
    """oracle/emd_oracle.c: xyz1 [B,n,3], xyz2 [B,m,3] -> match [B,m,n]."""
    a, b = _f(xyz1), _f(xyz2)
    B, n, _ = a.shape
    m = b.shape[1]
    match = np.empty((B, m, n), np.float32)
    lib().emd_oracle_approxmatch(B, n, m, _p(a), _p(b), _p(match))
    return match


def emd_matchcost(xyz1, xyz2, match):
    a, b, mt = _f(xyz1), _f(xyz2), _f(match)
    B, n, _ = a.shape
    cost = np.empty(B, np.float32)
    lib().emd_oracle_matchcost(B, n, b.shape[1], _p(a), _p(b), _p(mt), _p(cost))
    return cost


def emd_matchcost_backward(grad_cost, xyz1, xyz2, match):
    a, b, mt, gc = _f(xyz1), _f(xyz2), _f(match), _f(grad_cost)
    B, n, _ = a.shape
    g1, g2 = np.empty_like(a), np.empty_like(b)
    lib().emd_oracle_matchcost_backward(B, n, b.shape[1], _p(gc), _p(a), _p(b), _p(mt), _p(g1), _p(g2))
    return g1, g2
