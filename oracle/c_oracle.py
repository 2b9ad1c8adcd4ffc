"""ctypes binding of oracle/gate_route.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build() -> str:
    src = os.path.join(_HERE, "gate_route.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.m3o_gate_fwd.restype = ctypes.c_int
        _lib.m3o_route_build.restype = ctypes.c_int
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def gate_fwd(x, w, k, bias=None, noise=None, std=0.0, dense=True):
    """x [T,D] float32, w [D,E] float32 -> dict(idx i64[T,k], score, top_logits[T,min(k+1,E)],
    clean, noisy, gates)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    T, D = x.shape
    E = w.shape[1]
    kp = min(k + 1, E)
    idx = np.empty((T, k), np.int64)
    score = np.empty((T, k), np.float32)
    top = np.empty((T, kp), np.float32)
    clean = np.empty((T, E), np.float32) if dense else None
    noisy = np.empty((T, E), np.float32) if dense else None
    gates = np.empty((T, E), np.float32) if dense else None
    bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    noise = None if noise is None else np.ascontiguousarray(noise, dtype=np.float32)
    f = ctypes.c_float
    rc = lib().m3o_gate_fwd(_p(x, f), ctypes.c_int64(T), ctypes.c_int(D), ctypes.c_int64(D),
                            _p(w, f), ctypes.c_int(E), _p(bias, f), _p(noise, f), ctypes.c_float(std),
                            ctypes.c_int(k), _p(idx, ctypes.c_int64), _p(score, f), _p(top, f),
                            _p(clean, f), _p(noisy, f), _p(gates, f))
    if rc != 0:
        raise RuntimeError(f"m3o_gate_fwd failed: {rc}")
    return dict(idx=idx, score=score, top_logits=top, clean=clean, noisy=noisy, gates=gates)


def route_build(idx, E):
    flat = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
    n = flat.size
    counts = np.empty(E, np.int64)
    offsets = np.empty(E + 1, np.int64)
    pos = np.empty(n, np.int64)
    ros = np.empty(n, np.int64)
    i64 = ctypes.c_int64
    rc = lib().m3o_route_build(_p(flat, i64), i64(n), ctypes.c_int(E), _p(counts, i64), _p(offsets, i64),
                               _p(pos, i64), _p(ros, i64))
    if rc != 0:
        raise RuntimeError(f"m3o_route_build failed: {rc}")
    return counts, offsets, pos, ros
