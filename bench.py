#!/usr/bin/env python3
"""bench.py - per-group FP4 (E2M1) fake-quant throughput on MI355X.

Metric (BASELINE.json): Gelements/s + achieved HBM GB/s, per-group FP4 quant of an
fp16 [65536 x 1920] activation tensor, group 128.  One "step" = one pass of the
fused kernel over that tensor (one call of fp_quant_e2_per_group_cuda's HIP
replacement through the C ABI), input and output resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE is set), or
`python bench.py --gpus N` starts them itself - as a CHILD process (torch.distributed.run), before this process
has touched a GPU - relays the child's output and exits with its return code.  A rank whose WORLD_SIZE differs
from --gpus refuses to run.  The path is embarrassingly parallel over 128-element groups, so each rank quantizes
its own [65536 x 1920] shards with no data-path collective ("scaling": "weak"); the only collectives of the timed
region are the barrier and the MAX over ranks of the elapsed time.

Prints ONE JSON line (rank 0).  Exit code 0 only when every part ran; a collective that blocks in the secondary
(weight calibration) measurement is reported in the line AND by exit code 3.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS, COLS, GROUP = 65536, 1920, 128
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STREAM_CEILING_GBS = 6290.0     # same guide: 6.29 TB/s measured for a float4 copy (79 % of the spec figure)
BYTES_PER_ELEM = 4              # fp16 read + fp16 write
NBUF = 4                        # rotating buffer pairs, see GpuPlatform.hot_path
WATCHDOG_S = 240.0
CALIB_DEPTH = 30                # VAR-d30 (BASELINE.json config 4)
SEARCH_BLOCKS = 30              # the format search walks the 30 blocks of VAR-d30 (search/search_fp6_format.py:558)
# (model, path) of the `generation` records: BASELINE config 3's workload and config 5's metric (fpqvar_amd/var_block.py)
GENERATION_PLAN = [(m, p) for m in ("d30-256", "d36-512") for p in ("R", "F", "Q")]
EXIT_COLLECTIVE_TIMEOUT = 3


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip", default="", help="comma-separated secondary legs to leave out (profiling aid): "
                                               "calibration,generation,format_search,other_kernels,steps")
    return ap.parse_args(argv)


class Stage:
    """What the run is doing right now (for the watchdog): assigning stage[0] restarts its clock."""

    def __init__(self):
        self.name, self.t0, self.done, self.limit = "starting", time.monotonic(), threading.Event(), WATCHDOG_S

    def __setitem__(self, _, name):
        self.name, self.t0, self.limit = name, time.monotonic(), WATCHDOG_S

    def set(self, name, limit):
        """A stage whose honest duration grows with the arguments (warm-up, the timed region) brings its own limit."""
        self.name, self.t0, self.limit = name, time.monotonic(), max(WATCHDOG_S, limit)

    def __getitem__(self, _):
        return self.name

    def age(self):
        return time.monotonic() - self.t0


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args, script, argv):
    """`bench.py --gpus N` without a launcher: start N ranks as a child process and hand its exit code back.
    Nothing here touches the GPU (no HIP call, no torch.cuda.is_available()), and the child is a fresh process,
    never an exec of this one."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    # relay the child's output line by line; a rank that gave up on a blocked collective says so in its JSON line and
    # leaves with EXIT_COLLECTIVE_TIMEOUT, which torch.distributed.run flattens to 1: restore it here
    timed_out = False
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        if '"error": "timeout after' in line:
            timed_out = True
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    return EXIT_COLLECTIVE_TIMEOUT if (rc != 0 and timed_out) else rc


# ------------------------------------------------------------------------------------------------ platform seam
class GpuPlatform:
    """Everything bench.py needs from the machine: device, collective backend, synchronisation, the hot-path step and
    its timer.  tests/test_bench_contract.py swaps in a CPU / gloo stand-in to rehearse the control flow (launcher,
    rank checks, collectives, JSON line) without a GPU; the numbers then mean nothing and say so in `data`."""
    backend = "nccl"          # = RCCL on ROCm
    data = "synthetic"

    def __init__(self, local_rank):
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        index = local_rank
        if os.environ.get("FPQ_BENCH_SHARE_GPU"):
            # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks: the ranks share the devices there are,
            # the collectives run over gloo (RCCL refuses two ranks on one device).  Exercises every GPU-side step of the
            # multi-rank line - slabs, in-place gather on device memory, codes exchange; its numbers are no scaling curve.
            index = local_rank % torch.cuda.device_count()
            self.backend = "gloo"
            self.data = "synthetic; REHEARSAL: the ranks share one GPU, collectives over gloo - not a scaling measurement"
        torch.cuda.set_device(index)
        self.torch = torch
        self.dev = torch.device("cuda", index)

    def init_dist(self, dist):
        if self.backend == "nccl":
            dist.init_process_group(self.backend, device_id=self.dev)
        else:
            dist.init_process_group(self.backend)

    def synchronize(self):
        self.torch.cuda.synchronize()

    def empty_cache(self):
        self.torch.cuda.empty_cache()

    def hot_path(self, rank):
        """NBUF distinct input/output pairs (NBUF x 503 MB) used round-robin: every step streams its tensor from HBM
        and back; nothing is re-served by the 256 MiB Infinity Cache, as it would be if one 252 MB input were
        quantized over and over (that variant runs ~12 % faster with cached loads and is NOT what is reported)."""
        torch = self.torch
        from fpqvar_amd import _lib
        lib = _lib.lib()
        xs, outs = [], []
        for b in range(NBUF):
            torch.manual_seed(rank * NBUF + b)   # buffer 0 of rank 0 = seed 0 = the SURVEY.md 8d primary input
            xs.append(torch.randn(ROWS, COLS, device=self.dev).half())
            outs.append(torch.empty(ROWS, COLS, device=self.dev, dtype=torch.float16))
        self._keep = (xs, outs)
        n_rows = xs[0].numel() // GROUP
        self.stream = torch.cuda.current_stream(self.dev)
        sp = self.stream.cuda_stream
        ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in zip(xs, outs)]
        counter = [0]

        def step():
            xp, op = ptrs[counter[0] % NBUF]
            counter[0] += 1
            st = lib.fpq_quant_rows(xp, op, n_rows, GROUP, _lib.TABLE_IDS["e2m1"], _lib.F16, _lib.F16, sp)
            if st != 0:
                _lib.check(st, "fpq_quant_rows")

        return step, xs[0].numel(), xs[0]

    def release_hot_path(self):
        self._keep = None
        self.empty_cache()

    def generation_replica(self, rank):
        """This rank's replica of the model-shaped generation batches: one record per entry of GENERATION_PLAN."""
        from fpqvar_amd import var_block
        recs = []
        for model in dict.fromkeys(m for m, _ in GENERATION_PLAN):
            recs += var_block.generation_record((model,), [p for m, p in GENERATION_PLAN if m == model], device=self.dev, seed=rank)
        return recs

    def search_evaluator(self, blocks):
        """evaluate(b) for the block-sharded format search: the FP6 2 x 2 search of one d30 mat_qkv layer
        (search/search_fp6_format.py:589-608) over its 100 dumped samples [2, pn^2, C] (13600 rows); synthetic samples and
        weights seeded by the block index, resident before the clock starts (the reference loads them from disk)."""
        torch = self.torch
        from fpqvar_amd import format_search as fs
        pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
        data = {}
        for b in blocks:
            g = torch.Generator(device=self.dev).manual_seed(7000 + b)
            xs = [torch.randn(2, pns[j % 10] ** 2, COLS, device=self.dev, generator=g).half() for j in range(100)]
            data[b] = (xs, (torch.randn(3 * COLS, COLS, device=self.dev, generator=g) * 0.02).half())

        def evaluate(b):
            wf, af, losses = fs.search_layer(*data[b], fs.FP6_FORMATS)
            return wf, af, losses[(wf, af)]

        return evaluate

    def timer(self):
        """HIP events on the stream the kernel is launched on."""
        torch = self.torch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = self.stream
        return (lambda: ev0.record(stream)), (lambda: ev1.record(stream)), (lambda: ev0.elapsed_time(ev1))


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(sample_rows: int = 8192, budget_s: float = 28.0):
    """The reference's pure-torch CPU path (tr/quant_utils.py:209-230,298-310: abs -> max -> div -> |x - grid| ->
    argmin -> gather -> mul), restated in oracle/fpq_oracle.py and validated against the reference's own output on the
    golden vectors, timed on this box's host cores on a bounded sample of the metric's workload, in fp32 AND fp16
    (SURVEY.md 8d).  Per dtype: one warm-up run, then the minimum of >= 5 runs (more while the time budget lasts).
    torch's intra-op pool is tried at all cores and at 32 threads (oversubscribing a [N,15] elementwise graph with
    hundreds of threads is slower, not faster); the better one is reported with the thread count actually used."""
    import torch
    from oracle import fpq_oracle as orc
    g = torch.Generator().manual_seed(0)
    x32 = torch.randn(sample_rows, COLS, generator=g)
    all_cores = os.cpu_count() or 1
    t_start = time.perf_counter()

    def measure(x, threads, min_runs, deadline):
        torch.set_num_threads(threads)
        orc.per_group_argmin_sem(x.clone(), "e2m1", GROUP, clamp3=False)       # warm-up, discarded
        best, runs = float("inf"), 0
        while runs < min_runs or (runs < 12 and time.perf_counter() < deadline):
            xi = x.clone()
            t0 = time.perf_counter()
            orc.per_group_argmin_sem(xi, "e2m1", GROUP, clamp3=False)
            best = min(best, time.perf_counter() - t0)
            runs += 1
        return best, runs

    out = {}
    thread_opts = sorted({all_cores, min(32, all_cores)}, reverse=True)
    for name, x in (("f32", x32), ("f16", x32.half())):
        # pick the thread count on 2 quick runs each, then >= 5 timed runs at the better one
        trial = {t: measure(x, t, 2, 0.0)[0] for t in thread_opts}
        threads = min(trial, key=trial.get)
        share = budget_s * (0.5 if name == "f32" else 1.0)
        best, runs = measure(x, threads, 5, t_start + share)
        best = min(best, trial[threads])
        out[name] = {"Gelem_s": round(x.numel() / best / 1e9, 5), "cores": threads, "runs": runs + 2}
    res = {"value": out["f32"]["Gelem_s"], "unit": "Gelem/s", "cores": out["f32"]["cores"], "kind": "port",
           "sample": f"[{sample_rows}x{COLS}] g={GROUP} E2M1 of the metric's randn input, torch-op restatement of the "
                     f"reference CPU path (argmin over a [N,15] distance tensor); value = fp32 (the dtype the reference "
                     f"uses this path on); warm-up + min of >= 5 runs per dtype at the better of {thread_opts} threads",
           "f32": out["f32"], "f16": out["f16"]}
    try:   # third figure: the scalar C restatement of the kernel-semantics path, one core
        from oracle import c_oracle as co
        xh = x32[:1024].half()
        co.rows(xh, orc.TABLES["e2m1"], GROUP)
        t0 = time.perf_counter()
        co.rows(xh, orc.TABLES["e2m1"], GROUP)
        dt = time.perf_counter() - t0
        res["c_port_1core_gelems"] = round(xh.numel() / dt / 1e9, 5)
    except Exception:
        pass
    return res


def unfused_gpu_sequence(x, steps=3):
    """The reference's GPU op sequence (tr/quant_utils.py:313-330) as torch-ROCm ops
    around the L0 scan kernel: the 'before' picture on the same GPU."""
    import torch
    from fpqvar_amd import ops
    tab = torch.tensor([-6.0, -4.0, -3.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])

    def once():
        grid = tab.to(x.device)
        xs = x.reshape(-1, GROUP)
        scale = xs.abs().max(dim=-1, keepdim=True)[0] / grid.abs().max()
        xn = (xs / scale).view(-1).to(torch.float32)
        q = ops.quant_nearest(xn, grid.type_as(xn))
        torch.zeros_like(xn)                       # the reference's never-used `idx` output
        return (q.view(xs.shape) * scale).view(x.shape).to(x.dtype)

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(x.numel() / dt / 1e9, 3), "unit": "Gelem/s", "ms": round(dt * 1e3, 4)}


class Timing:
    """What `timed` measured: mean ms per launch over every timed launch, the best steady burst, the launch count."""

    def __init__(self, mean, min_burst, launches):
        self.mean, self.min_burst, self.launches = mean, min_burst, launches


def other_kernels(dev):
    """Secondary measurements on the same GPU (not the headline metric): the other kernels of the path at the
    BASELINE shapes, each with its own byte denominator (DESIGN.md section 4 quotes THESE figures)."""
    import torch
    from fpqvar_amd import ops, rotation as rot
    out = {}

    burst_env = os.environ.get("FPQ_BENCH_BURST")   # "launches,lead": protocol experiments (profiles/r03_burst_lead.txt)
    d_iters, d_lead = (int(v) for v in burst_env.split(",")) if burst_env else (100, 10)

    def timed(fn, iters=d_iters, max_bursts=40, lead=d_lead):
        """Steady-state time per call: bursts of `iters` calls (HIP events around each burst) until three consecutive
        bursts agree within 2 % - a kernel's first hundred launches after a change of workload run up to 25 % slow on
        this chip while the clocks settle (profiles/r02_ab_adaln_variants.txt) - then the minimum of those three and two
        more.  Every burst starts with `lead` untimed calls on the same stream: after the idle gap of a synchronize the
        core clock needs about a millisecond of load to come back (adaLN producer: first launch 126 us against 85 for
        the ones behind it; per-call averages of 88.9 / 86.9 / 85.9 / 85.2 us for bursts of 20 / 40+4 / 100+10 / 200+20
        launches on one box, profiles/r03_burst_lead.txt) - the kernels near the vector-issue limit feel it, the
        streaming ones do not; in a model these kernels follow each other without a gap.  Default: 100 timed launches
        behind 10 untimed ones."""
        def burst():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(lead):
                fn()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters

        every, last = [], []
        for _ in range(max_bursts):
            every.append(burst())
            last = every[-3:]
            if len(last) == 3 and max(last) <= 1.02 * min(last):
                break
        # two more bursts once steady: a plateau 5 - 10 % above the kernel's usual time was seen to satisfy the 2 % rule
        # for three bursts and then give way (profiles/r02_bench_repeats.txt, first line)
        every += [burst(), burst()]
        # ONE protocol per figure: `mean` = the average launch duration over EVERY timed launch of this
        # call - the settling bursts included, i.e. what a rocprofv3 --kernel-trace average of the same run sees - is what
        # the rates and fractions below are computed from, as the headline's are; `min_burst` = the best burst of the steady
        # tail (the figure rounds 1 - 4 printed) rides along.
        return Timing(sum(every) / len(every), min(every[-5:]), len(every) * iters)

    def hbm(name, t, nbytes):
        out[name] = {"ms": round(t.mean, 4), "min_burst_ms": round(t.min_burst, 4), "launches": t.launches,
                     "GBps": round(nbytes / t.mean / 1e6, 1), "frac_of_8TBps": round(nbytes / t.mean / 1e6 / HBM_PEAK_GBS, 3)}

    def guarded(name, fn):
        try:
            fn()
        except Exception as e:   # one broken secondary kernel must not hide the others
            out[name] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    g = torch.Generator(device=dev).manual_seed(1)
    k = [0]

    def nxt(lst):
        k[0] += 1
        return lst[k[0] % len(lst)]

    def fc2_inputs():
        return [torch.nn.functional.gelu(torch.randn(ROWS, 4 * COLS, device=dev, generator=g), approximate="tanh").half()
                for _ in range(2)]

    def dual_fp4():
        hs = fc2_inputs()
        hbm("dual_fc2_e1m2neg_e2m1pos_fp16_65536x7680",
            timed(lambda: ops.quant_rows_dual(nxt(hs), "e1m2_neg", "e2m1_pos", GROUP, 1.0)), hs[0].numel() * 4)

    def gelu_dual():
        # fc2.act_quant(act(y)) in one pass over the fc1 output (tr/basic_var.py:120-121, tr/quant_utils.py:991): 2 B read + 2 B written
        ys = [(torch.randn(ROWS, 4 * COLS, device=dev, generator=g) * 1.5).half() for _ in range(2)]
        hbm("gelu_dual_fc2_one_pass_fp16_65536x7680", timed(lambda: ops.gelu_quant_rows_dual(nxt(ys))), ys[0].numel() * 4)

    def dual_fp6():
        hs = fc2_inputs()
        n = hs[0].numel()
        hbm("dual_fc2_intneg_e2m3pos_per_group_fp16_65536x7680",
            timed(lambda: ops.quant_rows_dual(nxt(hs), "int_neg", "e2m3_pos", GROUP, None)), n * 4)
        hbm("dual_fc2_intneg_e2m3pos_per_token_fp16_65536x7680",
            timed(lambda: ops.quant_rows_dual(nxt(hs), "int_neg", "e2m3_pos", 4 * COLS, None)), n * 4)

    def act16():
        xs = [torch.randn(ROWS, COLS, device=dev, generator=g).half() for _ in range(3)]
        n = xs[0].numel()
        hbm("fused_rotate_quant_e2m1_fp16_65536x1920", timed(lambda: rot.rotate_quant(nxt(xs), "e2m1")), n * 4)
        hbm("fp6_e2m3_per_token_fp16_65536x1920", timed(lambda: ops.quant_rows(nxt(xs), "e2m3", COLS, torch.float16)), n * 4)
        # the complete producer (LayerNorm + AdaLN modulate + smooth + rotate + quant): B = 100 conditioned rows of a
        # d30 batch, L = 655 tokens each (= 65500 rows)
        B, L = 100, 655
        xa = [xs[i][:B * L].view(B, L, COLS) for i in range(3)]
        scale = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
        shift = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
        s = torch.rand(COLS, device=dev, generator=g) + 0.5
        hbm("adaln_rotate_quant_e2m1_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant(nxt(xa), scale, shift, "e2m1", smooth=s)), B * L * COLS * 4)
        # the same producer on fp32 rows - the dtype of the residual stream in the reference's autocast run
        # (tr/var.py:209, tr/basic_var.py:264,267): 4 B read + 2 B written per element
        del xs
        xf = [xa[i].float() for i in range(2)]
        del xa
        hbm("adaln_rotate_quant_e2m1_fp32rows_65500x1920",
            timed(lambda: rot.adaln_rotate_quant(nxt(xf), scale, shift, "e2m1", smooth=s)), B * L * COLS * 6)

    def operands():
        # The same producers emitting what the matrix-core GEMMs consume instead of fake-quantized values:
        # FP4: packed E2M1 codes + one fp16 scale per group of 128 = 0.516 B written per element; per token: E4M3 bytes
        # (1 B) or dense 6-bit codes (0.75 B) + one fp16 scale per row.  Denominator = bytes read + bytes written.
        xs = [torch.randn(ROWS, COLS, device=dev, generator=g).half() for _ in range(3)]
        n = xs[0].numel()
        wr4 = 0.5 + 2.0 / GROUP
        hbm("rotate_quant_codes_mx_fp16_65536x1920", timed(lambda: rot.rotate_quant_mx(nxt(xs))), n * (2 + wr4))
        B, L = 100, 655
        na = B * L * COLS
        xa = [xs[i][:B * L].view(B, L, COLS) for i in range(3)]
        scale = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
        shift = (torch.randn(B, 1, COLS, device=dev, generator=g) * 0.3).half()
        s = torch.rand(COLS, device=dev, generator=g) + 0.5
        hbm("adaln_rotate_quant_codes_mx_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_mx(nxt(xa), scale, shift, smooth=s)), na * (2 + wr4))
        hbm("adaln_rotate_quant_codes_mx_kmajor_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_mx(nxt(xa), scale, shift, smooth=s, kmajor=True)), na * (2 + wr4))
        hbm("adaln_rotate_quant_token_codes_fp6_kmajor_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_token(nxt(xa), scale, shift, "e2m3", smooth=s, emit="fp6", kmajor=True)), na * (2 + 0.75) + B * L * 2)
        hbm("adaln_rotate_quant_token_codes_fp8_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_token(nxt(xa), scale, shift, "e2m3", smooth=s, emit="fp8")), na * (2 + 1) + B * L * 2)
        hbm("adaln_rotate_quant_token_codes_fp6_fp16_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_token(nxt(xa), scale, shift, "e2m3", smooth=s, emit="fp6")), na * (2 + 0.75) + B * L * 2)
        del xs
        xf = [xa[i].float() for i in range(2)]
        del xa
        hbm("adaln_rotate_quant_codes_mx_fp32rows_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_mx(nxt(xf), scale, shift, smooth=s)), na * (4 + wr4))
        hbm("adaln_rotate_quant_token_codes_fp8_fp32rows_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_token(nxt(xf), scale, shift, "e2m3", smooth=s, emit="fp8")), na * (4 + 1) + B * L * 2)
        hbm("adaln_rotate_quant_token_codes_fp6_fp32rows_65500x1920",
            timed(lambda: rot.adaln_rotate_quant_token(nxt(xf), scale, shift, "e2m3", smooth=s, emit="fp6")), na * (4 + 0.75) + B * L * 2)

    def weights():
        ws = [torch.randn(ROWS // 2, COLS, device=dev, generator=g) * 0.02 for _ in range(3)]
        n = ws[0].numel()
        hbm("weights_e2m1_per_group_fp32_to_fp32_32768x1920", timed(lambda: ops.quant_rows(nxt(ws), "e2m1", GROUP)), n * 8)
        hbm("weights_e2m1_per_group_fp32_to_fp16_32768x1920",
            timed(lambda: ops.quant_rows(nxt(ws), "e2m1", GROUP, torch.float16)), n * 6)
        hbm("weights_e2m3_per_channel_fp32_to_fp16_32768x1920",
            timed(lambda: ops.quant_rows(nxt(ws), "e2m3", COLS, torch.float16)), n * 6)

    def consumers():
        # the consumers on the other side of the quantizers: matrix-core kernels, TFLOP/s
        from fpqvar_amd import gemm
        x = torch.randn(ROWS, COLS, device=dev, generator=g).half()
        a = gemm.quantize_mx(x)
        w = gemm.quantize_mx(torch.randn(3 * COLS, COLS, device=dev, generator=g) * 0.02)
        def flops(tag, t):
            out[tag] = {"ms": round(t.mean, 4), "min_burst_ms": round(t.min_burst, 4), "launches": t.launches,
                        "TFLOPs": round(2.0 * ROWS * COLS * 3 * COLS / t.mean / 1e9, 1),
                        "TFLOPs_min_burst": round(2.0 * ROWS * COLS * 3 * COLS / t.min_burst / 1e9, 1)}

        flops("gemm_fp4_w4a4_mat_qkv_65536x1920x5760", timed(lambda: gemm.linear_fp4(*a, *w)))
        # the same codes as k-major operand images (include/fpq.h): every LDS-DMA piece of the GEMM is 1 KiB contiguous
        ak = (gemm.to_kmajor(a[0], 4), gemm.to_kmajor_scales(a[1]))
        wk = (gemm.to_kmajor(w[0], 4, dealt=True), gemm.to_kmajor_scales(w[1], weight_side=True))
        flops("gemm_fp4_w4a4_mat_qkv_65536x1920x5760_kmajor", timed(lambda: gemm.linear_fp4(*ak, *wk)))
        del a, w, ak, wk
        # the W6A6 pair (per token x per channel): the 6-bit packed form and the E4M3-byte form of the same instruction
        wf = torch.randn(3 * COLS, COLS, device=dev, generator=g) * 0.02
        for tag, quant, lin in (("gemm_fp6_w6a6_mat_qkv_65536x1920x5760", gemm.quantize_fp6, gemm.linear_fp6),
                                ("gemm_fp8_rows_mat_qkv_65536x1920x5760", gemm.quantize_fp8, gemm.linear_fp8)):
            a, w = quant(x), quant(wf)
            flops(tag, timed(lambda: lin(*a, *w)))
            if "fp6" in tag:
                ak, wk = (gemm.to_kmajor(a[0], 6), a[1]), (gemm.to_kmajor(w[0], 6, dealt=True), w[1])
                flops(tag + "_kmajor", timed(lambda: lin(*ak, *wk)))
                del ak, wk
            del a, w

    def configs():
        """One driver-written number per BASELINE.json configuration that is not the metric's own (config 3 at the metric
        shape is everything above; its real row counts and config 5's are config3_steps / config5_steps of the line)."""
        from fpqvar_amd import calibrate as cal, format_search as fs, quant_utils as qu
        # config 1: single [4096 x 1024] fp32 tensor, per-tensor E2M1 through the pure-torch (argmin) semantics
        # (search/baseline/plot_weight_distribution_for_motivation.py:286-297): two launches (maxima, then scale + lookup),
        # 4 B read twice + 4 B written per element
        t1 = [torch.randn(4096, 1024, device=dev, generator=g) for _ in range(3)]
        hbm("config1_fp_quant_e2_per_tensor_fp32_4096x1024", timed(lambda: qu.fp_quant_e2_per_tensor(nxt(t1))), t1[0].numel() * 12)
        out["config1_fp_quant_e2_per_tensor_fp32_4096x1024"]["bytes_per_element"] = "4 (absmax pass) + 4 + 4"
        # config 2: the 16 mat_qkv weights of VAR-d16, [3072 x 1024] fp32, per group of 128 E2M1 -> fp32 as from_float does
        # (tr/quant_utils.py:828-837): 16 calls, and ONE launch over a segment table
        ws = {f"blocks.{b}.attn.mat_qkv": torch.randn(3072, 1024, device=dev, generator=g) * 0.02 for b in range(16)}
        n2 = sum(w.numel() for w in ws.values())
        hbm("config2_d16_mat_qkv_16_calls_fp32_to_fp32", timed(lambda: [qu.fp_quant_e2_per_group_cuda(w, 4, 128) for w in ws.values()],
                                                                iters=20, lead=2), n2 * 8)
        shard = cal.LocalShard(ws, out_dtype=torch.float32)
        hbm("config2_d16_mat_qkv_one_segment_launch_fp32_to_fp32", timed(shard.quantize, iters=50, lead=5), n2 * 8)
        del shard, ws
        # config 4: the format search of one d30 mat_qkv layer over its 100 dumped samples (13600 rows), batched
        # (search/search_fp6_format.py:589-608: FP6 2 x 2; search_fp4_format.py:782-821: FP4 3 x 3); ms per layer incl. its one read-back
        pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
        xs = [torch.randn(2, pns[j % 10] ** 2, COLS, device=dev, generator=g).half() for j in range(100)]
        w = (torch.randn(3 * COLS, COLS, device=dev, generator=g) * 0.02).half()
        for label, formats in (("fp6_2x2", fs.FP6_FORMATS), ("fp4_3x3", fs.FP4_FORMATS)):
            fs.search_layer(xs, w, formats)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                best = fs.search_layer(xs, w, formats)
            ms = (time.perf_counter() - t0) / 5 * 1e3
            out[f"config4_format_search_d30_mat_qkv_{label}_100_samples"] = {"ms_per_layer": round(ms, 3), "winner": list(best[:2]),
                                                                             "rows": 13600, "clock": "host wall clock incl. the read-back"}
        del xs, w
        # config 5: VAR-d36 512 x 512, the widths of its activations at the 44800 rows of one batch (B = 20 conditioned rows,
        # 2240 tokens): producers on the fp32 residual stream (C = 2304), dual format on the hidden width 9216
        B5, L5, C5 = 20, 2240, 2304
        x5 = [torch.randn(B5, L5, C5, device=dev, generator=g) for _ in range(2)]
        sc5 = (torch.randn(B5, 1, C5, device=dev, generator=g) * 0.3).half()
        sh5 = (torch.randn(B5, 1, C5, device=dev, generator=g) * 0.3).half()
        s5 = torch.rand(C5, device=dev, generator=g) + 0.5
        n5 = B5 * L5 * C5
        hbm("config5_adaln_rotate_quant_e2m1_fp32rows_44800x2304",
            timed(lambda: rot.adaln_rotate_quant(nxt(x5), sc5, sh5, "e2m1", smooth=s5)), n5 * 6)
        hbm("config5_adaln_rotate_quant_codes_mx_fp32rows_44800x2304",
            timed(lambda: rot.adaln_rotate_quant_mx(nxt(x5), sc5, sh5, smooth=s5)), n5 * (4 + 0.5 + 2.0 / GROUP))
        x5h = [t.half() for t in x5]
        del x5
        hbm("config5_adaln_rotate_quant_e2m1_fp16rows_44800x2304",
            timed(lambda: rot.adaln_rotate_quant(nxt(x5h), sc5, sh5, "e2m1", smooth=s5)), n5 * 4)
        a5 = [t.view(B5 * L5, C5) for t in x5h]
        hbm("config5_act_quant_e2m1_g128_fp16_44800x2304", timed(lambda: ops.quant_rows(nxt(a5), "e2m1", GROUP)), n5 * 4)
        del x5h, a5
        h5 = [torch.nn.functional.gelu(torch.randn(B5 * L5, 4 * C5, device=dev, generator=g), approximate="tanh").half() for _ in range(2)]
        hbm("config5_dual_fc2_e1m2neg_e2m1pos_fp16_44800x9216",
            timed(lambda: ops.quant_rows_dual(nxt(h5), "e1m2_neg", "e2m1_pos", GROUP, 1.0)), h5[0].numel() * 4)

    groups = [("baseline_configs_1_2_4_5", configs), ("dual_fc2_e1m2neg_e2m1pos_fp16_65536x7680", dual_fp4), ("gelu_dual_fc2_one_pass_fp16_65536x7680", gelu_dual), ("dual_fc2_intneg_e2m3pos_fp16_65536x7680", dual_fp6),
              ("activations_fp16_65536x1920", act16), ("operand_emitting_producers_65536x1920", operands),
              ("weights_fp32_32768x1920", weights), ("gemm_fp4_w4a4_mat_qkv_65536x1920x5760", consumers)]
    only = os.environ.get("FPQ_BENCH_GROUPS")   # profiling aid: a comma-separated subset of the group functions' names
    for name, fn in groups:
        if not only or fn.__name__ in only.split(","):
            guarded(name, fn)
    return out


# ------------------------------------------------------------------------------------------------ configs 3 and 5
STEP_MODELS = {
    # tr/var.py:175 - ten scale steps over patch_nums, B images with CFG = 2 B conditioned rows per token
    "d30": dict(C=1920, B=100, pn=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16),
                what="VAR-d30 256x256, 50 images with CFG (evaluate_fp_quant_transform_rotate.py:187-199)"),
    "d36-512": dict(C=2304, B=20, pn=(1, 2, 3, 4, 6, 9, 13, 18, 24, 32),
                    what="VAR-d36 512x512, 10 images with CFG (evaluate_fp_quant_transform_rotate_512x512.py:54,62,192-200)"),
}
STEP_POOL_BYTES = 3 << 29        # 1.5 GiB per side: six times the 256 MiB Infinity Cache
STEP_FLUSH_BYTES = 1 << 29


def generation_steps(dev, model="d30", rows_dtype="fp32", mode="rotating", replays=7, only_ops=None):
    """BASELINE configs 3 / 5 at the row counts they actually run (tr/var.py:175: rows = 2 B pn^2): the four quantizer
    calls of one W4A4 AdaLN block (tr/basic_var.py:263,266 -> two adaLN producers on the residual stream; tr/quant_utils.py:765
    -> the proj input; :991 -> the dual-format fc2 input) at every one of the ten scale steps, through the C ABI on
    preallocated buffers, N calls of one kind captured in ONE hipGraph and replayed.
    mode "rotating": call i of a graph works on slice i of a 1.5 GiB input pool and a 1.5 GiB output pool, and a 512 MiB
    write between replays empties the caches - every call streams cold HBM, also at 100 rows.  mode "resident": every call
    on the same tensor (what r03's tool did; at small steps the Infinity Cache serves it, as it may in a model whose
    previous kernel has just written the tensor).  rows_dtype: dtype of the residual stream entering the adaLN producer
    (fp32 under the reference's autocast, tr/var.py:209); the other two inputs are fp16 Linear / GELU outputs.
    Per step and kernel: median and minimum us per call over `replays` replays, algorithmic bytes, fraction of 8 TB/s; per
    model: sum of bytes / sum of (2 adaLN + act + dual) medians = the time-weighted fraction."""
    import ctypes
    import statistics
    import torch
    from fpqvar_amd import _lib, rotation as rot
    lib = _lib.lib()
    m = STEP_MODELS[model]
    C, B, HID = m["C"], m["B"], 4 * m["C"]
    x32 = rows_dtype == "fp32"
    g = torch.Generator(device=dev).manual_seed(3)
    scale = (torch.randn(B, C, device=dev, generator=g) * 0.3).half()
    shift = (torch.randn(B, C, device=dev, generator=g) * 0.3).half()
    smooth = torch.rand(C, device=dev, generator=g) + 0.5
    mask = rot._mask_arg(None)
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    flush = torch.empty(STEP_FLUSH_BYTES, dtype=torch.uint8, device=dev)
    E2M1, NEG, POS = _lib.TABLE_IDS["e2m1"], _lib.TABLE_IDS["e1m2_neg"], _lib.TABLE_IDS["e2m1_pos"]

    def pool(kind):
        n = STEP_POOL_BYTES // (4 if kind == "f32" else 2)
        chunk = 1 << 26
        t = torch.empty(n, dtype=torch.float32 if kind == "f32" else torch.float16, device=dev)
        for o in range(0, n, chunk):
            r = torch.randn(min(chunk, n - o), device=dev, generator=g)
            t[o:o + r.numel()] = torch.nn.functional.gelu(r, approximate="tanh") if kind == "gelu" else r
        return t

    def graph_us(call, in_bytes, out_bytes, xin, xout):
        n_calls = 50 if in_bytes + out_bytes < (64 << 20) else 20
        ib, ob = (in_bytes + 255) & ~255, (out_bytes + 255) & ~255
        k = 1 if mode == "resident" else max(1, min(n_calls, STEP_POOL_BYTES // max(ib, ob)))
        xi, xo = xin.data_ptr(), xout.data_ptr()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            sp = _lib.stream_ptr(dev)
            call(xi, xo, sp)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                sp = _lib.stream_ptr(dev)
                for i in range(n_calls):
                    call(xi + (i % k) * ib, xo + (i % k) * ob, sp)
        torch.cuda.current_stream().wait_stream(s)
        for _ in range(3):
            gr.replay()
        ts = []
        for _ in range(replays):
            if mode != "resident":
                flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / n_calls)
        del gr
        return statistics.median(ts), min(ts), n_calls, k

    def ck(st, what):
        if st != 0:
            _lib.check(st, what)

    steps = [dict(pn=pn, rows=B * pn * pn) for pn in m["pn"]]
    xout = torch.empty(STEP_POOL_BYTES, dtype=torch.uint8, device=dev)
    plan = [("adaln", "f32" if x32 else "f16", C), ("act", "f16", C), ("dual", "gelu", HID)]
    for name, kind, cols in plan:
        if only_ops and name not in only_ops:
            continue
        xin = pool(kind)
        for stp in steps:
            rows, L = stp["rows"], stp["pn"] ** 2
            in_b = rows * cols * (4 if kind == "f32" else 2)
            out_b = rows * cols * 2
            if name == "adaln":
                def call(xi, xo, sp, rows=rows, L=L):
                    ck(lib.fpq_adaln_rotate_quant_rows(xi, xo, None, None, rows, C, _lib.F32 if x32 else _lib.F16,
                                                       scale.data_ptr(), shift.data_ptr(), _lib.F16, L, 1e-6,
                                                       smooth.data_ptr(), mask, E2M1, sp), "fpq_adaln_rotate_quant_rows")
            elif name == "act":
                def call(xi, xo, sp, rows=rows):
                    ck(lib.fpq_quant_rows(xi, xo, rows * (C // GROUP), GROUP, E2M1, _lib.F16, _lib.F16, sp), "fpq_quant_rows")
            else:
                def call(xi, xo, sp, rows=rows):
                    ck(lib.fpq_quant_rows_dual(xi, xo, rows * (HID // GROUP), GROUP, NEG, POS, _lib.F16, _lib.F16, None, 1.0,
                                               flag.data_ptr(), sp), "fpq_quant_rows_dual")
            med, best, n_calls, k = graph_us(call, in_b, out_b, xin, xout)
            stp[name] = {"us": round(med, 2), "min_us": round(best, 2), "bytes": in_b + out_b,
                         "frac_of_8TBps": round((in_b + out_b) / med / 1e3 / HBM_PEAK_GBS, 3), "calls_per_graph": n_calls,
                         "slices": k}
        del xin
        torch.cuda.empty_cache()
    assert not bool(flag.any()), "the dual quantizer's NaN scratch must be zero again"
    # What a dependent launch costs before it moves a byte, in THIS protocol (same graph machinery, cold caches): the smallest
    # launch of the path - one group of 128 - fifty times in one graph.  The bound below prices every launch of a block-step at
    # this floor plus its bytes at the chip's 1:1 stream ceiling (a 16-byte copy: 6.29 TB/s, MI355X_MICROARCH.md), so that
    # the "five dependent launches bound the step at 0.66" of DESIGN.md section 4c is a number the driver's run reproduces.
    floor_us = None
    try:
        tiny = torch.zeros(1 << 16, dtype=torch.float16, device=dev)

        def call(xi, xo, sp):
            ck(lib.fpq_quant_rows(xi, xo, 1, GROUP, E2M1, _lib.F16, _lib.F16, sp), "fpq_quant_rows")
        floor_us = graph_us(call, 256, 256, tiny, xout)[0]
        del tiny
    except Exception:
        pass
    weights = {"adaln": 2, "act": 1, "dual": 1}     # calls per block and step
    launches = {"adaln": 1, "act": 1, "dual": 2}    # launches per call (the dual quantizer's NaN fix-up is the second)
    have = [n for n, _, _ in plan if all(n in s for s in steps)]
    by_kernel, tot_t, tot_b = {}, 0.0, 0
    for n in have:
        t = sum(s[n]["us"] for s in steps)
        b = sum(s[n]["bytes"] for s in steps)
        by_kernel[n] = {"sum_us": round(t, 1), "bytes": b, "frac_of_8TBps": round(b / t / 1e3 / HBM_PEAK_GBS, 3)}
        # least-squares line us = fixed + rows * slope over the ten steps: what a call costs before it moves a byte, and the
        # rate it approaches (the time-weighted fraction above mixes the two)
        xs_, ys_ = [float(s["rows"]) for s in steps], [s[n]["us"] for s in steps]
        mx, my = sum(xs_) / len(xs_), sum(ys_) / len(ys_)
        slope = sum((x - mx) * (y - my) for x, y in zip(xs_, ys_)) / sum((x - mx) ** 2 for x in xs_)
        by_kernel[n]["fit"] = {"fixed_us_per_call": round(my - slope * mx, 2), "ns_per_row": round(slope * 1e3, 3),
                               "frac_of_8TBps_of_the_slope": round(steps[-1][n]["bytes"] / steps[-1]["rows"] / slope / 1e3 / HBM_PEAK_GBS, 3)}
        tot_t += weights[n] * t
        tot_b += weights[n] * b
    for s in steps:
        if len(have) == 3:
            t = sum(weights[n] * s[n]["us"] for n in have)
            b = sum(weights[n] * s[n]["bytes"] for n in have)
            s["block_us"] = round(t, 2)
            s["frac_of_8TBps"] = round(b / t / 1e3 / HBM_PEAK_GBS, 3)
    bound = None
    if floor_us and len(have) == 3:
        t_bound = sum(weights[n] * (launches[n] * floor_us + s[n]["bytes"] / (STREAM_CEILING_GBS * 1e3)) for s in steps for n in have)
        bound = {"launch_floor_us": round(floor_us, 2), "stream_ceiling_GBps": STREAM_CEILING_GBS,
                 "launches_per_block_and_step": sum(weights[n] * launches[n] for n in have),
                 "block_us_at_the_bound": round(t_bound, 1), "bound_frac_of_8TBps": round(tot_b / t_bound / 1e3 / HBM_PEAK_GBS, 4),
                 "what": "every launch of a block-step at the measured floor of a dependent launch in this protocol + its bytes at the "
                         "1:1 stream ceiling: the time-weighted fraction this launch sequence cannot exceed"}
    return {"model": model, "what": m["what"], "rows_dtype_of_the_residual_stream": rows_dtype, "mode": mode, "bound": bound,
            "clock": "hipGraph replay, HIP events around each replay, median over %d replays per (step, kernel)" % replays,
            "per_block_and_step": "2 x adaLN producer (values out) + 1 x E2M1 g=128 (proj input) + 1 x dual E1M2-/E2M1+ g=128 "
                                  "(fc2 input, default clipping strength: two launches)",
            "steps": steps, "by_kernel": by_kernel,
            "block_us_over_the_ten_steps": round(tot_t, 1), "bytes_per_block": tot_b,
            "time_weighted_frac_of_8TBps": round(tot_b / tot_t / 1e3 / HBM_PEAK_GBS, 4) if tot_t else None}


def steps_summary(full):
    """The compact form for the bench line (the full record goes to profiles/ through tools/bench_small_steps.py)."""
    out = {k: full[k] for k in ("model", "rows_dtype_of_the_residual_stream", "mode", "clock", "per_block_and_step",
                                "block_us_over_the_ten_steps", "bytes_per_block", "time_weighted_frac_of_8TBps")}
    if full.get("bound"):
        out["launch_floor_us"] = full["bound"]["launch_floor_us"]
        out["bound_frac"] = full["bound"]["bound_frac_of_8TBps"]
        out["bound"] = full["bound"]
    out["by_kernel"] = {n: v["frac_of_8TBps"] for n, v in full["by_kernel"].items()}
    out["fit_us_fixed_plus_ns_per_row"] = {n: v["fit"] for n, v in full["by_kernel"].items()}
    out["rows"] = [s["rows"] for s in full["steps"]]
    for n in full["by_kernel"]:
        out[n + "_us"] = [s[n]["us"] for s in full["steps"]]
    out["step_frac_of_8TBps"] = [s.get("frac_of_8TBps") for s in full["steps"]]
    return out


# ------------------------------------------------------------------------------------------------ config 4
def weight_calibration(plat, dist, world, rank, stage, depth=None, iters=10, warm=6):
    """BASELINE.json config 4: every Linear weight of VAR-d30 (1.327 G fp32 elements, synthetic randn*0.02) quantized
    per-group(128) E2M1 -> fp16, layers partitioned over the ranks (fpqvar_amd.calibrate.partition), each rank
    materialising and quantizing only its own share ("ms": no collective on the data path, max over ranks), and, at
    N > 1, the same followed by the ONE all-gather that leaves every rank with the whole quantized model
    ("ms_with_all_gather": calibrate.calibrate_sharded, fp16 exchange over RCCL).  Strong scaling (the model is fixed).
    Secondary measurement, never the headline value.  `stage[0]` names what is running (for the watchdog); every
    failure is reported as {"error": ...}, never swallowed."""
    import torch
    depth = CALIB_DEPTH if depth is None else depth
    dev = plat.dev
    dt, dt_g, dt_c = float("nan"), float("nan"), float("nan")
    total = 0
    err = None
    codes_err = None
    own, shapes, cal = {}, {}, None
    try:
        stage[0] = "local quantization"
        from fpqvar_amd import calibrate as cal
        shapes = cal.var_linear_shapes(depth)
        sizes = [(n, o * i) for n, (o, i) in shapes.items()]
        total = sum(sz for _, sz in sizes)
        mine = cal.partition(sizes, world)[rank]
        torch.manual_seed(1000 + rank)
        own = {n: torch.randn(*shapes[n], device=dev) * 0.02 for n in mine}
        local = cal.LocalShard(own, shapes)            # segment table + output slab, built once (not timed)
        for _ in range(warm):                          # the clocks take ~10 ms of this load to settle (1.29 - 1.56 ms
            local.quantize()                           # per launch over the first launches, profiles/r02_bench_kernel_stats.csv)
        plat.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            local.quantize()                           # ONE launch over all of this rank's layers
        plat.synchronize()
        dt = (time.perf_counter() - t0) / iters
        del local
    except Exception as e:
        err = f"local quantization: {e!r}"[:300]
    if dist is not None:   # every rank reaches these collectives, whatever happened above
        stage[0] = "all_reduce of the local status"
        flag = torch.tensor([0.0 if err else 1.0, dt if dt == dt else 1e30], device=dev, dtype=torch.float64)
        mn = flag.clone()
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        all_ok, dt = bool(mn[0].item() > 0.5), float(flag[1].item())
        if not all_ok and err is None:
            err = "local quantization failed on another rank"
        if all_ok:         # all ranks are healthy: time the gathered form (same code path on every rank)
            try:
                stage[0] = "sharded calibration + all_gather_into_tensor"
                plan = cal.ShardedCalibration(shapes, own, group=None)   # slab + segment table, built once (not timed)
                for _ in range(3):
                    plan.run()
                plat.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                for _ in range(iters):
                    plan.run()
                plat.synchronize()
                dist.barrier()
                t = torch.tensor([(time.perf_counter() - t0) / iters], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_g = float(t.item())
                del plan
            except Exception as e:
                err = f"gathered calibration (fp16 exchange): {e!r}"[:300]
            # the same with the packed exchange format (nibble codes + one fp32 scale per group: 0.53 B per element on the
            # wire instead of 2, decoded locally, bit-identical): a second curve for the same run.  Its own try block and
            # its own status exchange: a rank that failed above (or fails building the plan) must not leave the others
            # blocked in this phase's collectives.
            codes_run, cerr = None, None
            try:
                stage[0] = "codes calibration plan"
                if err is None:
                    codes_run = getattr(plat, "codes_calibration", None)
                    if codes_run is None:   # slabs + the two segment tables built once (not timed), as for the fp16 form
                        codes_plan = cal.ShardedCodesCalibration(own_all(shapes, own, dev), group=None)
                        codes_run = codes_plan.run
            except Exception as e:
                cerr = f"codes calibration plan: {e!r}"[:300]
            stage[0] = "all_reduce of the codes-plan status"
            okc = torch.tensor([0.0 if (err or cerr or codes_run is None) else 1.0], device=dev, dtype=torch.float64)
            dist.all_reduce(okc, op=dist.ReduceOp.MIN)
            if okc.item() > 0.5:
                try:
                    stage[0] = "sharded calibration, codes exchange + all_gather_into_tensor"
                    codes_run()
                    plat.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        codes_run()
                    plat.synchronize()
                    dist.barrier()
                    t = torch.tensor([(time.perf_counter() - t0) / 3], device=dev, dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt_c = float(t.item())
                except Exception as e:
                    cerr = f"codes exchange: {e!r}"[:300]
            if cerr and err is None:
                codes_err = cerr
            elif okc.item() <= 0.5 and err is None:   # this rank is healthy, another is not: say so instead of leaving the key out
                codes_err = "skipped: the fp16 exchange or the codes plan failed on another rank"
    stage[0] = "done"
    try:
        own.clear()
        plat.empty_cache()
    except Exception:
        pass
    if rank != 0:
        return None
    if err is not None or not (dt < 1e29) or total == 0:
        return {"error": err or "no timing"}
    res = {"workload": f"VAR-d{depth} all-Linear weights fp32 -> per-group(128) E2M1 -> fp16, layers sharded over the ranks, "
                       "one launch per rank over a segment table",
           "elements": total, "n_gpus": world, "ms": round(dt * 1e3, 3), "Gelem_s": round(total / dt / 1e9, 1),
           "GBps_at_6B_per_elem": round(total * 6 / dt / 1e9, 1),
           "frac_of_8TBps_per_gpu": round(total * 6 / dt / 1e9 / world / HBM_PEAK_GBS, 3),
           "scaling": "strong"}
    if dt_g == dt_g:
        res["exchange"] = "fp16 (the de-quantized weights, 2 B per element), in-place all_gather_into_tensor"
        res["ms_with_all_gather"] = round(dt_g * 1e3, 3)
        res["Gelem_s_with_all_gather"] = round(total / dt_g / 1e9, 1)
        res["gathered_bytes_per_rank"] = 2 * total
        ag = max(dt_g - dt, 1e-9)
        res["all_gather_ms"] = round(ag * 1e3, 3)                                  # = ms_with_all_gather - ms
        res["all_gather_GBps_per_rank"] = round(2 * total * (world - 1) / world / ag / 1e9, 1)   # bytes a rank receives / time
    if codes_err is not None:
        res["codes_exchange"] = {"error": codes_err}
    if dt_c == dt_c:
        res["codes_exchange"] = {"exchange": "nibble codes + fp32 group scales (0.53 B per element), decoded locally",
                                 "ms_with_all_gather": round(dt_c * 1e3, 3), "Gelem_s_with_all_gather": round(total / dt_c / 1e9, 1),
                                 "gathered_bytes_per_rank": int(total * (0.5 + 4.0 / GROUP)),
                                 "note": "one launch quantizes the rank's layers into its slot of a prebuilt codes slab, one "
                                         "launch decodes every layer of every rank after the gather (both inside this time)"}
    return res


# ------------------------------------------------------------------------------------------------ config 5 (and 3)
def generation(plat, dist, world, rank, stage):
    """BASELINE.json config 5's metric - sample throughput of the generation loop
    (evaluate_fp_quant_transform_rotate_512x512.py:196-214) - and config 3's workload (tr/var.py:175, tr/basic_var.py:263-267):
    the transformer part of one generation batch of VAR-d30 256x256 (50 images with CFG) and VAR-d36 512x512 (10 images),
    W4A4 + FP6 KV cache, random weights, one hipGraph per scale step, one eager warm-up batch + best of 3 replays
    (fpqvar_amd/var_block.py).  Paths: R = the reference's op sequence on this GPU, F = fused fake-quant launches around fp16
    Linears, Q = operands straight into the FP4 matrix cores; torch's OWN GEMMs (fc2 on every path, every Linear of R / F) with
    TunableOp selections recorded once for these shapes (var_block.tuned_torch_gemms; named in the records; the same on all
    three paths, no tuning at run time).  The loop is independent per (class, seed) - "replicas only",
    no collective in the model: every rank runs its own replica, the line carries the slowest rank's time
    and images of ALL ranks / that time (fpqvar_amd.generation.aggregate_throughput: one SUM and one MAX per record).
    Weak scaling.  Secondary measurement; every failure is reported in the record, never swallowed."""
    from fpqvar_amd import generation as gen
    stage.set("generation replicas", 900.0)
    try:
        recs = plat.generation_replica(rank)
        assert len(recs) == len(GENERATION_PLAN)
    except Exception as e:
        recs = [{"model": m, "path": p, "error": f"replica failed: {e!r}"[:200]} for m, p in GENERATION_PLAN]
    out = []
    for (model, path), rec in zip(GENERATION_PLAN, recs):       # every rank walks the same plan: same collectives
        ok = "error" not in rec and rec.get("ms_per_batch", 0) > 0
        imgs = rec.get("images_per_batch", 0) if ok else 0
        secs = rec["ms_per_batch"] / 1e3 if ok else 0.0
        total, rate = imgs, (imgs / secs if secs > 0 else 0.0)
        ranks_ok = 1 if ok else 0
        if dist is not None:
            stage[0] = f"generation: all_reduce for {model} {path}"
            total, rate = gen.aggregate_throughput(imgs, secs if ok else 0.0, device=plat.dev if plat.backend == "nccl" else None)
            t = plat.torch.tensor([float(ranks_ok)], dtype=plat.torch.float64, device=plat.dev if plat.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ranks_ok = int(t.item())
        line = {k: rec[k] for k in ("model", "path", "config", "what", "images_per_batch", "clock", "warmup_eager_ms", "torch_gemms", "operands", "kv_cache") if k in rec}
        line.update({"model": model, "path": path, "replicas": world, "replicas_ok": ranks_ok})
        if ranks_ok == world and rate > 0:
            line["ms_per_batch"] = round(total / rate * 1e3, 2)               # the slowest replica's batch (total / rate = its seconds)
            line["images_per_s"] = round(rate, 1)                             # whole job
        else:
            line["error"] = rec.get("error", "failed on another rank")
        out.append(line)
    return out if rank == 0 else None


# ------------------------------------------------------------------------------------------------ config 4, the search
def format_search_sharded(plat, dist, world, rank, stage, n_blocks=None):
    """BASELINE.json config 4's second half: the per-block format search (search/search_fp6_format.py:589-608: FP6 2 x 2 over
    {e2m3, e3m2} for weight and activation, argmin of the output MSE) of the 30 mat_qkv layers of VAR-d30, 100 samples
    (13600 rows) each, blocks dealt over the ranks (block b on rank b % N) and ONE all-gather of (loss, w_fmt, a_fmt)
    triples (fpqvar_amd.format_search.search_blocks_sharded).  "ms_local": the slowest rank's own blocks; "ms": the same
    plus the gather.  Strong scaling (30 blocks whatever N).  Samples and weights are resident before the clock starts."""
    from fpqvar_amd import format_search as fs
    n_blocks = SEARCH_BLOCKS if n_blocks is None else n_blocks
    err, res = None, None
    t_eval, dt = [0.0], float("nan")
    try:
        stage.set("format search: data", 600.0)
        mine = list(range(rank, n_blocks, world))
        evaluate = plat.search_evaluator(mine)
        for b in mine[:1]:
            evaluate(b)                                  # warm-up (kernel load, GEMM heuristics)

        def timed_eval(b):
            t0 = time.perf_counter()
            r = evaluate(b)                              # ends with the layer's one read-back: the host clock sees the GPU time
            t_eval[0] += time.perf_counter() - t0
            return r
    except Exception as e:
        err = f"format search set-up: {e!r}"[:300]
    if dist is not None:   # every rank reaches these collectives, whatever happened above
        stage[0] = "format search: all_reduce of the set-up status"
        okt = plat.torch.tensor([0.0 if err else 1.0], dtype=plat.torch.float64, device=plat.dev if plat.backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if okt.item() < 0.5 and err is None:
            err = "format search set-up failed on another rank"
    if err is None:
        try:
            stage[0] = "format search: sharded search + all_gather"
            plat.synchronize()
            if dist is not None:
                dist.barrier()
            t0 = time.perf_counter()
            res = fs.search_blocks_sharded(n_blocks, timed_eval, fs.FP6_FORMATS)
            plat.synchronize()
            dt = time.perf_counter() - t0
            if dist is not None:
                t = plat.torch.tensor([dt, t_eval[0]], dtype=plat.torch.float64, device=plat.dev if plat.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, t_eval[0] = float(t[0].item()), float(t[1].item())
        except Exception as e:
            err = f"sharded search: {e!r}"[:300]
    stage[0] = "done"
    plat.empty_cache()
    if rank != 0:
        return None
    if err is not None:
        return {"error": err}
    winners = {}
    for wf, af, _ in res:
        winners[f"{wf}/{af}"] = winners.get(f"{wf}/{af}", 0) + 1
    return {"workload": f"FP6 2x2 format search of the {n_blocks} mat_qkv layers of VAR-d30 ([5760x1920] fp16 weights, 100 samples = "
                        "13600 rows each, synthetic), blocks dealt over the ranks, one all-gather of (loss, w_fmt, a_fmt)",
            "n_gpus": world, "blocks": n_blocks, "blocks_on_the_busiest_rank": (n_blocks + world - 1) // world,
            "ms": round(dt * 1e3, 3), "ms_local": round(t_eval[0] * 1e3, 3), "all_gather_ms": round(max(dt - t_eval[0], 0.0) * 1e3, 3),
            "layers_per_s": round(n_blocks / dt, 1), "scaling": "strong", "winners": winners,
            "clock": "host wall clock (every layer ends with its one read-back), max over ranks"}


def own_all(shapes, own, dev):
    """calibrate_sharded wants every name (shapes are read from the tensors; only owned values are touched): layers of
    other ranks as zero-stride placeholders of the right shape - no memory, never read."""
    import torch
    z = torch.zeros((), device=dev)
    return {n: own[n] if n in own else z.expand(shapes[n]) for n in shapes}


def pmc_traffic():
    """HBM bytes per launch of the headline kernel.  NOT measured by this process (rocprofv3 counters cannot be read
    from inside the run): the figure of the committed rocprofv3 --pmc passes of this same command, with its source."""
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                with open(p) as f:
                    return json.load(f).get("traffic_bytes_per_launch"), f"profiles/{name}"
            except Exception:
                continue
    return None, None


def headline_kernel_label():
    """Which instantiation fpq_quant_rows picks for the metric shape: the library switch FPQ_NO_HW4 (asked of the library
    itself, fpq_get_option) keeps the bucket table; the vectors per lane are build-time constants of fpq_kernels.hip
    (FPQ_FAST16_HW4_U = 1, FPQ_FAST16_U = 2)."""
    from fpqvar_amd import _lib
    if _lib.get_option("FPQ_NO_HW4"):
        return "rows16_lut_subwave_kernel<16 lanes/group, U=2, bucket table in LDS (FPQ_NO_HW4)>"
    return "rows16_lut_subwave_kernel<16 lanes/group, U=1, HW4: E2M1 levels from the FP4 conversion hardware>"


def build_result(args, world, elems, elapsed, kernel_ms, calib, rccl_ranks, data):
    """The contract line (everything the timed region determines)."""
    ms_per_step = elapsed / args.steps * 1e3
    value = world * elems / (elapsed / args.steps) / 1e9
    achieved = elems * BYTES_PER_ELEM / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic()
    res = {
        "metric": "Gelements/s + achieved HBM GB/s, per-group FP4 quant [65536x1920,g=128]",
        "value": round(value, 3),
        "unit": "Gelem/s",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": data,
        "config": {"workload": "fp16 [65536x1920] randn, per-group(128) FP4 E2M1 fake-quant, fp16 out; "
                               f"{NBUF} distinct tensors per GPU used round-robin (cold HBM every step); "
                               "one shard of this shape per GPU, no data-path collective",
                   "rows": ROWS, "cols": COLS, "group": GROUP, "format": "fp_e2 (E2M1)",
                   "parallelism": f"shard{world}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": (f"{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                        "from an earlier run (gfx950 FETCH_SIZE x2 correction applied); a constant, not a "
                                        "measurement of this timed region") if traffic_src else None,
                     "kernel": headline_kernel_label(),
                     "kernel_ms": round(kernel_ms, 5),
                     "clocks": "two clocks in this line: roofline.achieved / frac use kernel_ms = HIP events on the launch stream "
                               "around the K timed launches / K (the kernel's average launch duration); value / ms_per_step use "
                               "the host's wall clock around the same K launches including the closing synchronize and, at N > 1, "
                               "the barriers (the contract's whole-job figure); the first is the larger by the host's share",
                     "algorithmic_bytes": elems * BYTES_PER_ELEM},
    }
    if calib is not None:
        res["weight_calibration"] = calib
    return res


# ------------------------------------------------------------------------------------------------ main
def main(argv=None, script=None, platform_factory=GpuPlatform):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    script = script or os.path.abspath(__file__)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, script, argv))     # parent: start the ranks, relay, hand the code back

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank "
                         f"run as a {args.gpus}-GPU figure")
    import torch
    # N > 1: every collective of this run - rendezvous, barriers, the MAX over ranks, the all-gather - sits under a
    # watchdog from here on.  `stage` names what is running and restarts the clock; a stage that does not finish within
    # WATCHDOG_S ends the process with EXIT_COLLECTIVE_TIMEOUT and the reason in rank 0's JSON line: a hang is never
    # reported as success.
    stage = Stage()
    partial = [None]        # a function building the line from what has been measured so far

    def watchdog():
        while not stage.done.wait(0.5):
            limit = stage.limit if rank == 0 else 1.5 * stage.limit + 5.0   # rank 0 reports first: the line is its to print
            if stage.age() > limit:
                err = {"error": f"timeout after {stage.limit:.0f} s in {stage[0]}"}
                if rank == 0:
                    line = partial[0](err) if partial[0] is not None else dict(
                        metric="Gelements/s + achieved HBM GB/s, per-group FP4 quant [65536x1920,g=128]", value=None,
                        n_gpus=world, **err)
                    print(json.dumps(line), flush=True)
                os._exit(EXIT_COLLECTIVE_TIMEOUT)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()
    stage[0] = "device initialisation"
    plat = platform_factory(local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        stage[0] = "init_process_group"
        plat.init_dist(dist)
        rccl_ranks = dist.get_world_size()
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: the process group has {rccl_ranks} ranks, WORLD_SIZE says {world}")

    # (a step of the hot path takes ~0.1 ms; the limits of these two stages grow with --warmup / --steps so that a long
    # honest run is not reported as a blocked collective)
    stage.set("warm-up", WATCHDOG_S + 0.01 * args.warmup)
    step, elems, x0 = plat.hot_path(rank)
    for _ in range(args.warmup):
        step()
    plat.synchronize()
    stage[0] = "barrier before the timed region"
    if dist is not None:
        dist.barrier()
    plat.synchronize()
    stage.set("timed region", WATCHDOG_S + 0.01 * args.steps)
    start, stop, elapsed_ms = plat.timer()
    t0 = time.perf_counter()
    start()                       # same stream the kernel is launched on
    for _ in range(args.steps):
        step()
    stop()
    plat.synchronize()
    stage[0] = "barrier after the timed region"
    if dist is not None:
        dist.barrier()
    plat.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = elapsed_ms() / args.steps                 # average launch duration, HIP events

    if dist is not None:
        stage[0] = "all_reduce(MAX) of the elapsed time"
        t = torch.tensor([elapsed], device=plat.dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def headline(calib):
        return build_result(args, world, elems, elapsed, kernel_ms, calib, rccl_ranks, plat.data)

    partial[0] = headline
    skip = set(filter(None, args.skip.split(",")))
    # The sharded calibration (with its all-gather at N > 1) is a secondary measurement taken AFTER the timed region.
    calib = None if "calibration" in skip else weight_calibration(plat, dist, world, rank, stage)
    if calib is not None and world > 1:
        calib["note"] = ("expected to be gather-bound: one GPU quantizes the whole model in ~1.3 ms, the all-gather moves 2 B per "
                         "element to every rank (DESIGN.md section 6); the parts of configs 4 / 5 that shard profitably are "
                         "format_search_sharded and generation")
    # ... and so are the two parts of configs 4 / 5 that shard: every rank takes part at every N, 1 included
    extra = {}

    def headline_with(calib_):
        line = headline(calib_)
        line.update(extra)
        return line

    partial[0] = headline_with
    release = getattr(plat, "release_hot_path", None)
    if "format_search" not in skip:
        res_fs = format_search_sharded(plat, dist, world, rank, stage)
        if res_fs is not None:
            extra["format_search_sharded"] = res_fs
    if "generation" not in skip:
        if world > 1 and release is not None:
            x0 = None
            release()                       # the generation batches want the memory of the headline's buffers at N > 1 too
        res_gen = generation(plat, dist, world, rank, stage)
        if res_gen is not None:
            extra["generation"] = res_gen
    stage.done.set()

    if rank == 0:
        res = headline_with(calib)
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["unfused_gpu"] = unfused_gpu_sequence(x0)
            except Exception as e:  # extra information only
                res["unfused_gpu"] = {"error": repr(e)[:200]}
            del x0
            plat.release_hot_path()
            if "other_kernels" not in skip:
                try:
                    res["other_kernels"] = other_kernels(plat.dev)
                except Exception as e:
                    res["other_kernels"] = {"error": repr(e)[:200]}
            # BASELINE configs 3 and 5 at their real row counts (ten scale steps), fp32 residual stream, cold inputs
            for key, model in (("config3_steps", "d30"), ("config5_steps", "d36-512")):
                if "steps" in skip:
                    break
                try:
                    res[key] = steps_summary(generation_steps(plat.dev, model, "fp32", "rotating"))
                except Exception as e:
                    res[key] = {"error": repr(e)[:200]}
            res["cpu_baseline"] = cpu_baseline()
        if world > 1:
            res["omitted_at_n_gt_1"] = ["cpu_baseline", "other_kernels", "unfused_gpu"]   # N = 1 lines carry them
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
