"""ctypes front-end of oracle/libfpq_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfpq_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _tab(t):
    return np.ascontiguousarray(t.numpy().astype(np.float32))


def nearest(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    xa = np.ascontiguousarray(x.numpy().astype(np.float32).reshape(-1))
    z = np.empty_like(xa)
    t = _tab(table)
    lib().fpq_oracle_nearest_f32(_p(xa), _p(t), _p(z), ctypes.c_int64(xa.size), ctypes.c_int(t.size))
    return torch.from_numpy(z).reshape(x.shape)


def rows(x: torch.Tensor, table: torch.Tensor, cols: int, out_f16: bool = False) -> torch.Tensor:
    t = _tab(table)
    n_rows = x.numel() // cols
    if x.dtype == torch.float16:
        xa = np.ascontiguousarray(x.contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1))
        out = np.empty_like(xa)
        rc = lib().fpq_oracle_rows_f16(_p(xa), _p(out), ctypes.c_int64(n_rows), ctypes.c_int64(cols), _p(t),
                                       ctypes.c_int(t.size))
        assert rc == 0
        return torch.from_numpy(out.view(np.int16)).view(torch.float16).reshape(x.shape)
    xa = np.ascontiguousarray(x.contiguous().numpy().reshape(-1))
    if out_f16:
        out = np.empty(xa.size, dtype=np.uint16)
        rc = lib().fpq_oracle_rows_f32(_p(xa), None, _p(out), ctypes.c_int64(n_rows), ctypes.c_int64(cols), _p(t),
                                       ctypes.c_int(t.size))
        assert rc == 0
        return torch.from_numpy(out.view(np.int16)).view(torch.float16).reshape(x.shape)
    out = np.empty_like(xa)
    rc = lib().fpq_oracle_rows_f32(_p(xa), _p(out), None, ctypes.c_int64(n_rows), ctypes.c_int64(cols), _p(t),
                                   ctypes.c_int(t.size))
    assert rc == 0
    return torch.from_numpy(out).reshape(x.shape)


def rows_dual(x: torch.Tensor, tneg: torch.Tensor, tpos: torch.Tensor, cols: int) -> torch.Tensor:
    a, b = _tab(tneg), _tab(tpos)
    n_rows = x.numel() // cols
    if x.dtype == torch.float16:
        xa = np.ascontiguousarray(x.contiguous().view(torch.int16).numpy().view(np.uint16).reshape(-1))
        out = np.empty_like(xa)
        rc = lib().fpq_oracle_rows_dual_f16(_p(xa), _p(out), ctypes.c_int64(n_rows), ctypes.c_int64(cols), _p(a),
                                            ctypes.c_int(a.size), _p(b), ctypes.c_int(b.size))
        assert rc == 0
        return torch.from_numpy(out.view(np.int16)).view(torch.float16).reshape(x.shape)
    xa = np.ascontiguousarray(x.contiguous().numpy().reshape(-1))
    out = np.empty_like(xa)
    rc = lib().fpq_oracle_rows_dual_f32(_p(xa), _p(out), ctypes.c_int64(n_rows), ctypes.c_int64(cols), _p(a),
                                        ctypes.c_int(a.size), _p(b), ctypes.c_int(b.size))
    assert rc == 0
    return torch.from_numpy(out).reshape(x.shape)


def h2f(h: torch.Tensor) -> torch.Tensor:
    ha = np.ascontiguousarray(h.view(torch.int16).numpy().view(np.uint16).reshape(-1))
    f = np.empty(ha.size, dtype=np.float32)
    lib().fpq_oracle_h2f(_p(ha), _p(f), ctypes.c_int64(ha.size))
    return torch.from_numpy(f)


def f2h(f: torch.Tensor) -> torch.Tensor:
    fa = np.ascontiguousarray(f.numpy().astype(np.float32).reshape(-1))
    h = np.empty(fa.size, dtype=np.uint16)
    lib().fpq_oracle_f2h(_p(fa), _p(h), ctypes.c_int64(fa.size))
    return torch.from_numpy(h.view(np.int16)).view(torch.float16)
