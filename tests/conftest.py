import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(GOLDEN)


def from_bits(arr: np.ndarray) -> torch.Tensor:
    """uint16/uint32/uint64 bit patterns -> fp16/fp32/fp64 tensor."""
    if arr.dtype == np.uint16:
        return torch.from_numpy(arr.view(np.int16).copy()).view(torch.float16)
    if arr.dtype == np.uint32:
        return torch.from_numpy(arr.view(np.int32).copy()).view(torch.float32)
    if arr.dtype == np.uint64:
        return torch.from_numpy(arr.view(np.int64).copy()).view(torch.float64)
    raise TypeError(arr.dtype)


def assert_bits_equal(got: torch.Tensor, want: torch.Tensor, what: str = ""):
    """Bit-exact comparison where every NaN is treated as one value (NaN payloads
    are not a contract of the reference) but +0.0 and -0.0 differ."""
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} != {tuple(want.shape)}"
    g = got.detach().cpu().contiguous()
    w = want.detach().cpu().contiguous()
    gn, wn = torch.isnan(g), torch.isnan(w)
    assert torch.equal(gn, wn), f"{what}: NaN pattern differs ({int(gn.sum())} vs {int(wn.sum())})"
    itype = {torch.float16: torch.int16, torch.float32: torch.int32, torch.float64: torch.int64}[g.dtype]
    gi = torch.where(gn, torch.zeros_like(g), g).view(itype)
    wi = torch.where(wn, torch.zeros_like(w), w).view(itype)
    bad = gi != wi
    if bad.any():
        k = int(bad.reshape(-1).nonzero()[0])
        raise AssertionError(
            f"{what}: {int(bad.sum())} of {bad.numel()} elements differ; first at flat index {k}: "
            f"got {g.reshape(-1)[k].item()!r} want {w.reshape(-1)[k].item()!r}")


@pytest.fixture
def lib_options():
    """`lib_options(name, value)` sets an experiment switch of libfpq_hip.so (fpq_set_option, include/fpq.h) and the
    fixture puts every touched switch back when the test ends.  value None = the library's built-in choice."""
    from fpqvar_amd import _lib
    saved = {}

    def set_option(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)

    yield set_option
    for name, value in saved.items():
        _lib.set_option(name, value)
