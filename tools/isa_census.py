#!/usr/bin/env python3
"""Instruction census of one kernel from a `hipcc -save-temps` assembly file.

    python tools/isa_census.py <file.s> <substring of the mangled kernel name> [--phases]

Prints, per basic block (label), the number of instructions by class (VALU / SALU / LDS / VMEM / MFMA / other), the
kernel's register and scratch figures from its metadata, and - with --phases, for a build made with -DFPQ_ISA_CENSUS,
whose sources carry `; PHASE <name>` markers behind scheduling barriers - the same split per phase inside each block.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "MFMA"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "WAIT"
    if op.startswith("s_"):
        return "SALU"
    return "other"


# Vector-pipe cycles per wave-instruction at saturation (4 wavefronts per SIMD), measured by tools/probe/valu_issue_cost.hip
# on MI355X (profiles/r03_valu_issue_cost.txt).  What the census' "pipe" column adds up: a kernel bound by vector issue
# takes about this many cycles per SIMD, not "instructions x 4".
def valu_cost(op):
    if op.startswith(("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_cvt_scalef32_pk_fp4", "v_cvt_scalef32_pk_fp6", "v_cvt_scalef32_pk32")):
        return 8.3
    if op.startswith(("v_permlane", "v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")):
        return 8.5
    if op.startswith("v_cndmask"):
        return 4.3          # (the probe's vcc form read 23 cycles: not understood, not used here)
    if op.startswith("v_pk_") and any(t in op for t in ("_f16", "add_u16", "add_i16")):
        return 4.85
    if "_dpp" in op or "_sdwa" in op or op.startswith(("v_pk_", "v_fma_mix_f32", "v_max", "v_min", "v_med3", "v_dot2", "v_perm_b32", "v_bfe_",
                                                      "v_lshl_or", "v_lshl_add", "v_add3", "v_and_or", "v_or3", "v_xad", "v_cvt_", "v_alignbit",
                                                      "v_mad_", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_readfirstlane", "v_cmp", "v_mbcnt", "v_bcnt",
                                                      "v_ashrrev_i64", "v_lshlrev_b64", "v_lshl_add_u64", "v_mov_b64")):
        return 4.4
    return 2.5


def kernel_body(lines, needle):
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") or ": " in l:
            m = re.match(r"^(_Z\S+):", l)
            if m and needle in m.group(1):
                start, name = i, m.group(1)
                break
    if start is None:
        raise SystemExit(f"no kernel label containing {needle!r}")
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return name, lines[start + 1:end]


def main():
    path, needle = sys.argv[1], sys.argv[2]
    phases = "--phases" in sys.argv
    lines = open(path, errors="replace").read().split("\n")
    name, body = kernel_body(lines, needle)
    print("kernel:", name)
    meta = {}
    for l in lines:
        m = re.match(r"\s*\.set " + re.escape(name) + r"\.(\w+), (\S+)", l)
        if m:
            meta[m.group(1)] = m.group(2)
    for l in lines:
        pass
    print("metadata:", {k: meta[k] for k in ("num_vgpr", "num_agpr", "numbered_sgpr", "private_seg_size") if k in meta})
    blocks = collections.OrderedDict()
    cur, phase = "entry", "-"
    nsplit = collections.Counter()
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^; PHASE (\S+)", s)
        if m:
            phase = m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        key = (cur, phase if phases else "-")
        blocks.setdefault(key, collections.Counter())[classify(op)] += 1
        blocks[key]["op:" + op] += 1
        if classify(op) == "VALU":
            blocks[key]["pipe"] += valu_cost(op)
        elif classify(op) == "MFMA":
            blocks[key]["pipe"] += 8.0      # an MFMA holds the SIMD's vector issue for 8 of its 16 cycles
        if op.startswith(("s_cbranch", "s_branch")):      # the fall-through part is a block of its own
            nsplit[cur.split("+")[0]] += 1
            cur = cur.split("+")[0] + "+" + str(nsplit[cur.split("+")[0]])
    tot = collections.Counter()
    for (blk, ph), c in blocks.items():
        n = sum(v for k, v in c.items() if not k.startswith("op:"))
        if n < 8 and not phases:
            continue
        print(f"{blk:14s} {ph:14s} " + " ".join(f"{k}={c[k]}" for k in ("VALU", "MFMA", "LDS", "VMEM", "SALU", "WAIT") if c[k]) +
              (f"  pipe~{c['pipe']:.0f}cyc" if c["pipe"] else ""))
        for k, v in c.items():
            tot[k] += v
    if phases:
        # the row loop is unrolled twice (two register sets alternate): per-row figures = half of the kernel's totals
        per = collections.OrderedDict()
        for (blk, ph), c in blocks.items():
            d = per.setdefault(ph, collections.Counter())
            for k, v in c.items():
                d[k] += v
        print("\nper phase, whole kernel (row-loop phases appear twice: two register sets):")
        for ph, c in per.items():
            print(f"  {ph:24s} " + " ".join(f"{k}={c[k]}" for k in ("VALU", "MFMA", "LDS", "VMEM", "SALU", "WAIT") if c[k]) +
                  (f"  pipe~{c['pipe']:.0f}cyc" if c["pipe"] else ""))
    if "--ops" in sys.argv:
        want = [a for a in sys.argv[3:] if a.startswith(".LBB") or a.startswith("phase=")]
        agg = collections.Counter()
        for (blk, ph), c in blocks.items():
            if want and blk not in want and ("phase=" + ph) not in want:
                continue
            for k, v in c.items():
                if k.startswith("op:"):
                    agg[k[3:]] += v
        for k, v in agg.most_common():
            print(f"   {v:5d} {k}")


if __name__ == "__main__":
    main()
