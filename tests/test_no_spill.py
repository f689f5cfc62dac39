"""No hot kernel may spill: the register and scratch figures of every kernel in libfpq_hip.so, read from the code
objects' own metadata (llvm-readelf --notes on the gfx950 ELFs bundled in the .so's .hip_fatbin section).  Runs on CPU:
the library is cross-compiled by __graft_entry__.build().  A spill shows as .vgpr_spill_count / .sgpr_spill_count > 0 or
a non-zero .private_segment_fixed_size (scratch); round 2's rotate_quant_mfma_kernel<half, EMIT=0, SMOOTH=1> had 4
spilled registers = 20 bytes of scratch."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fpqvar_amd", "libfpq_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# the kernels of the hot path: quantizers, producers, KV step, code emitters
HOT = re.compile(r"rows16_lut|groups32|rows32|rotate_quant_mfma|adaln_mfma|adaln_rq16|kv16_step|rows16_codes|codes128|decode128|gemm_fp4_glds|gemm_fp6_rows|gemm_fp8_rows")


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def kernel_metadata(tmp_path):
    """[(kernel name, {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size})]"""
    objcopy, bundler, readelf = (_tool(t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    if not all((objcopy, bundler, readelf)):
        pytest.skip("LLVM binutils of the ROCm toolchain not found")
    fat = str(tmp_path / "fatbin")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    assert starts, "no offload bundle in .hip_fatbin"
    out = []
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(data)
        bundle, elf = str(tmp_path / f"bundle{k}"), str(tmp_path / f"co{k}.elf")
        open(bundle, "wb").write(data[a:b])
        subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}",
                        f"--output={elf}"], check=True)
        notes = subprocess.run([readelf, "--notes", elf], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip().strip("'\"")
            if key == "agpr_count" and (cur is None or "agpr_count" in cur):   # first key of a kernel's record
                cur = {}
                out.append(cur)
            if cur is not None:
                cur[key] = val
    recs = [(r.get("name", "?"), r) for r in out if "name" in r]
    assert recs, "no kernel records in the notes"
    return recs


@pytest.mark.skipif(not os.path.exists(LIB), reason="libfpq_hip.so not built")
def test_hot_kernels_do_not_spill(tmp_path):
    recs = kernel_metadata(tmp_path)
    hot = [(n, r) for n, r in recs if HOT.search(n)]
    assert len(hot) > 50, f"only {len(hot)} hot kernels found among {len(recs)}"
    bad = []
    for n, r in hot:
        spill = int(r.get("vgpr_spill_count", 0)) + int(r.get("sgpr_spill_count", 0))
        scratch = int(r.get("private_segment_fixed_size", 0))
        if spill or scratch:
            bad.append(f"{n[:160]}: {r.get('vgpr_spill_count')} VGPRs / {r.get('sgpr_spill_count')} SGPRs spilled, {scratch} B of scratch")
    assert not bad, "spilling kernels:\n" + "\n".join(bad)


@pytest.mark.skipif(not os.path.exists(LIB), reason="libfpq_hip.so not built")
def test_occupancy_relevant_register_budgets(tmp_path):
    """The register budgets the designs count on: the adaLN producer's 32 KiB form (five workgroups
    per CU) needs <= 96 registers and exactly 32768 bytes of LDS; the persistent rotate kernel 6 wavefronts per SIMD
    (<= 80 registers)."""
    recs = kernel_metadata(tmp_path)

    def find(pred):
        got = [(n, r) for n, r in recs if pred(n)]
        assert got, "kernel not found"
        return got
    # (mangled names: <half or float modulation, MAXC 4, values, no emit, per group, fp16 rows, HW4, TIGHT, 4 wavefronts>)
    for n, r in find(lambda n: re.search(r"adaln_mfma_kernelI(DF16_|f)Li4ELb0ELb0ELb0ELb0ELb1ELb1ELi4E", n)):
        assert int(r["vgpr_count"]) <= 96 and int(r["group_segment_fixed_size"]) <= 32768, (n, r["vgpr_count"], r["group_segment_fixed_size"])
    # <fp16 input, EMIT any, SMOOTH = 0, ...>
    for n, r in find(lambda n: re.search(r"rotate_quant_mfma_kernelIDF16_Lb[01]ELb0E", n)):
        assert int(r["vgpr_count"]) <= 80, (n, r["vgpr_count"])


@pytest.mark.skipif(not os.path.exists(LIB), reason="libfpq_hip.so not built")
def test_fc1_tail_register_budgets(tmp_path):
    """The fc1 GEMM with GELU + dual quantizer in its epilogue (round 5) keeps the plain GEMM's occupancy: 256 x 128 tiles two
    workgroups per CU (<= 256 registers per lane), 128 x 128 and 64 x 128 tiles three (<= 168), no scratch."""
    recs = kernel_metadata(tmp_path)
    fc1 = [(n, r) for n, r in recs if "gemm_fp4_glds_kernel" in n and "GemmFc1" in n]
    assert len(fc1) == 6, [n for n, _ in fc1]                      # {fp16, fp32 weight scales} x {64, 128, 256 rows}
    for n, r in fc1:
        limit = 256 if "Li8ELi4E" in n else 168
        assert int(r["vgpr_count"]) + int(r.get("agpr_count", 0)) <= limit and int(r.get("private_segment_fixed_size", 0)) == 0, (n, r["vgpr_count"])


def _code_objects(tmp_path):
    """the gfx950 code objects inside the library's fat binary (paths)"""
    objcopy, bundler = _tool("llvm-objcopy"), _tool("clang-offload-bundler")
    if not all((objcopy, bundler, _tool("llvm-objdump"))):
        pytest.skip("LLVM binutils of the ROCm toolchain not found")
    fat = str(tmp_path / "fatbin_isa")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    out = []
    for k, a in enumerate(starts):
        b = starts[k + 1] if k + 1 < len(starts) else len(data)
        bundle, elf = str(tmp_path / f"isa_bundle{k}"), str(tmp_path / f"isa_co{k}.elf")
        open(bundle, "wb").write(data[a:b])
        subprocess.run([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={bundle}", f"--output={elf}"], check=True)
        out.append(elf)
    return out


@pytest.mark.skipif(not os.path.exists(LIB), reason="libfpq_hip.so not built")
def test_gemm_main_loops_carry_no_compiler_vmem_waits(tmp_path):
    """The LDS-DMA GEMMs wait for a stage with ONE explicit `s_waitcnt vmcnt(0)` in front of the loop's barrier; the loads are
    assembly the compiler does not see.  If its scoreboard believes that other loads are still in flight (the prologue's scale
    loads behind a branch it cannot rule out: round 5), it inserts vmcnt waits INSIDE the loop - each one waits for the stage just
    requested and the pipeline is gone (8 % on the FP4 GEMM, silently).  Between the loop's barrier and its back edge there must
    be none."""
    checked = 0
    for elf in _code_objects(tmp_path):
        txt = subprocess.run([_tool("llvm-objdump"), "-d", elf], capture_output=True, text=True, check=True).stdout.splitlines()
        heads = [i for i, l in enumerate(txt) if re.search(r"<.*(gemm_fp4_glds_kernel|gemm_fp6_rows_kernel|gemm_fp8_rows_kernel).*>:$", l)]
        for h in heads:
            ops = []                                     # (address, text)
            for l in txt[h + 1:]:
                m = re.match(r"\t(.*?)\s*// ([0-9A-Fa-f]+):", l)
                if m:
                    ops.append((int(m.group(2), 16), " ".join(m.group(1).split())))
                    if ops[-1][1].startswith("s_endpgm"):
                        break
            mf = [i for i, (_, o) in enumerate(ops) if o.startswith("v_mfma")]
            assert mf, txt[h]
            # the main loop: closed by the first branch behind the last MFMA that jumps back in front of the first one
            loop = None
            for i in range(mf[-1], len(ops)):
                addr, o = ops[i]
                if o.startswith("s_cbranch"):
                    off = int(o.split()[1])
                    target = addr + 4 + 4 * (off - 65536 if off >= 32768 else off)
                    if target <= ops[mf[0]][0]:
                        loop = [o2 for a2, o2 in ops if target <= a2 <= addr]
                        break
            between = [o for _, o in ops[mf[0]:mf[-1] + 1] if o.startswith("s_waitcnt") and "vmcnt" in o]
            assert not between, (txt[h][:160], between)                      # never among the MFMAs
            if "gemm_fp4_glds_kernel" in txt[h]:                             # (the row-scaled kernels' loops are not contiguous in the image)
                assert loop, txt[h]
                waits = [o for o in loop if o.startswith("s_waitcnt") and "vmcnt" in o]
                assert waits == ["s_waitcnt vmcnt(0)"], (txt[h][:160], waits)    # the explicit one in front of the barrier, nothing else
            checked += 1
    assert checked >= 20, checked
