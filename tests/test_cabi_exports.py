"""CPU-only: the C-ABI library loads and exports every symbol include/fpq.h declares;
argument validation that needs no GPU; the host mirror refuses CPU tensors."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FPQ_VERSION = int(re.search(r"#define FPQ_VERSION (\d+)", open(os.path.join(ROOT, "include", "fpq.h")).read()).group(1))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from fpqvar_amd import _lib
    return _lib.lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fpq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fpq_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fpq.h but not exported by libfpq_hip.so"
    from fpqvar_amd import _lib
    assert set(_lib._SIGS) == set(names), "ctypes signatures out of sync with include/fpq.h"


def test_version_and_errors(lib):
    assert lib.fpq_version() == FPQ_VERSION
    assert lib.fpq_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert lib.fpq_strerror(code)
    # argument validation happens before anything touches a device
    assert lib.fpq_quant_rows(None, None, -1, 128, 0, 0, 0, None) == -1
    assert lib.fpq_quant_rows(None, None, 4, 128, 99, 0, 0, None) == -4
    assert lib.fpq_quant_rows(None, None, 4, 128, 5, 0, 0, None) == -4      # half table as symmetric table
    assert lib.fpq_quant_rows(None, None, 4, 128, 0, 2, 0, None) == -2      # f64 not supported here
    assert lib.fpq_quant_rows(None, None, 4, 128, 0, 0, 0, None) == -1      # null pointers
    assert lib.fpq_quant_rows(None, None, 0, 128, 0, 0, 0, None) == 0       # empty input is fine
    assert lib.fpq_quant_rows_dual(None, None, 4, 128, 0, 6, 0, 0, None, 1.0, None, None) == -4
    assert lib.fpq_quant_nearest(None, None, None, 4, 300, 1, None) == -3
    assert lib.fpq_quant_nearest(None, None, None, 4, 15, 0, None) == -2
    assert lib.fpq_quant_nearest(None, None, None, 0, 15, 1, None) == 0
    assert lib.fpq_quant_rows_codes(None, None, None, 4, 128, 3, 0, 1, None) == -3   # FP6 codes cannot be nibble-packed
    assert lib.fpq_quant_tensor_argmin(None, None, None, None, 4, 0, 1, None) == -1  # no scale / workspace
    assert lib.fpq_quant_tensor_argmin(None, None, None, None, 4, 5, 1, None) == -4  # half table
    assert lib.fpq_quant_tensor_argmin(None, None, None, None, 4, 0, 2, None) == -2  # f64


def test_tables_match_oracle(lib):
    from fpqvar_amd import _lib, quant_utils as qu
    from oracle import fpq_oracle as orc
    for name in _lib.TABLE_IDS:
        assert torch.equal(_lib.table_values(name), orc.TABLES[name]), name
    assert torch.equal(qu.fp4_e2m1_grid, orc.TABLES["e2m1"])
    assert torch.equal(qu.fp4_e1m2_grid, orc.TABLES["e1m2"])
    assert torch.equal(qu.fp4_e3m0_grid, orc.TABLES["e3m0"])
    assert torch.equal(qu.fp6_e2m3_grid, orc.TABLES["e2m3"])
    assert torch.equal(qu.fp6_e3m2_grid, orc.TABLES["e3m2"])
    assert torch.equal(qu.int_neg_grid, orc.TABLES["int_neg"])
    assert torch.equal(qu.e2m3_pos_grid, orc.TABLES["e2m3_pos"])
    assert _lib.TABLE_IDS == orc.TABLE_IDS


def test_no_cpu_fallback():
    import quant_cuda
    from fpqvar_amd import quant_utils as qu
    x = torch.randn(4, 128)
    with pytest.raises(RuntimeError, match="GPU"):
        qu.fp_quant_e2_per_group_cuda(x.half(), 4, 128)
    with pytest.raises(RuntimeError, match="GPU"):
        quant_cuda.quant(x.view(-1), qu.fp4_e2m1_grid)
    with pytest.raises(AssertionError):
        qu.fp_quant_e2_per_group_cuda(x.half(), 8, 128)


def test_product_does_not_import_oracle():
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); import fpqvar_amd, quant_cuda; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imports oracle'")
    subprocess.run([sys.executable, "-c", code % ROOT], check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fpqvar_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("oracle/", "").replace(
                    "the oracle", ""), f"{f} mentions the oracle module"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: without libfpq_hip.so the loader raises, it does not degrade."""
    from fpqvar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libfpq_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().fpq_version() == FPQ_VERSION


def test_option_table_is_the_only_reader_of_the_environment(lib):
    """include/fpq.h: no entry point calls getenv().  (a) In the sources, getenv appears in ONE place, the initialiser of
    the option table; (b) in the shipped library, the only code that calls getenv@plt is that initialiser (the static
    constructor of fpq_kernels.hip) - checked in the disassembly of the host code."""
    import subprocess
    csrc = os.path.join(ROOT, "fpqvar_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            for i, line in enumerate(open(os.path.join(csrc, f)), 1):
                code = line.split("//")[0]
                if "getenv" in code:
                    hits.append((f, i))
    assert len(hits) == 1 and hits[0][0] == "fpq_kernels.hip", hits
    so = os.path.join(ROOT, "fpqvar_amd", "libfpq_hip.so")
    dis = subprocess.run(["objdump", "-d", "--no-show-raw-insn", so], capture_output=True, text=True, check=True).stdout
    callers, cur = set(), None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
        elif "<getenv@plt>" in line and cur != "getenv@plt":
            callers.add(cur)
    assert callers, "expected the option initialiser to call getenv"
    assert all(("GLOBAL__sub_I" in c or "FpqOptionInit" in c or c.startswith(".plt")) for c in callers), callers


def test_options_roundtrip_and_env_is_read_once(lib, monkeypatch):
    from fpqvar_amd import _lib
    names = _lib.option_names()
    assert "FPQ_NO_HW4" in names and "FPQ_GEMM_CFG" in names and len(names) == len(set(names)) >= 19
    assert lib.fpq_option_name(len(names)) is None and lib.fpq_option_name(-1) is None
    for n in names:
        before = _lib.get_option(n)
        with _lib.option(n, 7):
            assert _lib.get_option(n) == 7
        assert _lib.get_option(n) == before
    assert lib.fpq_set_option(b"FPQ_NO_SUCH_SWITCH", 1) == -1 and lib.fpq_set_option(None, 1) == -1
    assert lib.fpq_get_option(b"FPQ_NO_HW4", None) == -1
    # the environment was read when the library was loaded; a later change is not seen
    before = _lib.get_option("FPQ_NO_HW6")
    monkeypatch.setenv("FPQ_NO_HW6", "1" if not before else "0")
    assert _lib.get_option("FPQ_NO_HW6") == before
    # ... and a fresh process does see it, with the documented parsing (flags: 1 unless "0"; numbers: atoi; empty = unset)
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from fpqvar_amd import _lib; "
            "print(_lib.get_option('FPQ_NO_HW4'), _lib.get_option('FPQ_NO_HW6'), _lib.get_option('FPQ_GEMM_CFG'), "
            "_lib.get_option('FPQ_NO_FAST32'), _lib.get_option('FPQ_ADALN_ROWS'))" % ROOT)
    env = dict(os.environ, FPQ_NO_HW4="yes", FPQ_NO_HW6="0", FPQ_GEMM_CFG="20", FPQ_NO_FAST32="", FPQ_ADALN_ROWS="12")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "0", "20", "None", "12"], out


def test_header_is_plain_c(tmp_path):
    """include/fpq.h is the boundary a cgo / JNI / ctypes binding compiles against: it must be valid C99 on its own (no torch, no
    HIP headers), and the structs the Python layer mirrors must have the sizes ctypes gives them."""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from fpqvar_amd import _lib
    src = tmp_path / "hdr.c"
    src.write_text('#include <stdio.h>\n#include "fpq.h"\nint main(void) { printf("%zu %zu %zu %d\\n", sizeof(fpq_gemm_epilogue_t), '
                   'sizeof(fpq_gemm_split_t), sizeof(fpq_segment_t), FPQ_VERSION); return 0; }\n')
    exe = tmp_path / "hdr"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    import ctypes
    assert int(out[0]) == ctypes.sizeof(_lib.GemmEpilogue) and int(out[1]) == ctypes.sizeof(_lib.GemmSplit) and int(out[2]) == ctypes.sizeof(_lib.Segment)
    assert int(out[3]) == FPQ_VERSION
