"""bench.py and __graft_entry__ keep the driver's contract: one JSON line with the agreed fields (GPU), and the
argument parser / constants are sane (CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_constants_match_baseline():
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert (bench.ROWS, bench.COLS, bench.GROUP) == (65536, 1920, 128)
    assert "65536" in base["metric"] and "1920" in base["metric"] and "128" in base["metric"]
    assert bench.BYTES_PER_ELEM == 4 and bench.HBM_PEAK_GBS == 8000.0


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
    assert "--offload-arch=gfx950" in g.HIP_FLAGS


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in res, key
    assert res["steps"] == 5 and res["warmup"] == 2 and res["n_gpus"] == 1 and res["vs_baseline"] is None
    assert res["unit"] == "Gelem/s" and res["dtype"] == "f16" and res["scaling"] == "weak" and "workload" in res["config"]
    rf = res["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.3 < rf["frac"] < 1.0, rf            # a 5-step run on a cold box is slower than the steady state, never absurd
    assert res["value"] > 400


@pytest.mark.gpu
def test_bench_two_ranks_share_the_gpu_over_gloo():
    """`bench.py --gpus 2` with FPQ_BENCH_SHARE_GPU=1: two ranks on the one GPU of a test box, collectives over gloo (RCCL
    refuses two ranks on a device).  Every GPU-side step of the multi-rank line runs for real - the shards, the barrier
    and the MAX over ranks, the sharded calibration's slab with its in-place all-gather on device memory, the codes
    exchange with its two segment launches; the line says that it is a rehearsal, its numbers are no scaling curve."""
    env = dict(os.environ, FPQ_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and "REHEARSAL" in res["data"] and res["scaling"] == "weak"
    cal = res["weight_calibration"]
    assert "error" not in cal, cal
    assert cal["n_gpus"] == 2 and cal["ms"] > 0 and cal["ms_with_all_gather"] > cal["ms"]
    assert cal["all_gather_ms"] > 0 and cal["gathered_bytes_per_rank"] == 2 * cal["elements"]
    cx = cal["codes_exchange"]
    assert cx["ms_with_all_gather"] > 0 and cx["gathered_bytes_per_rank"] < cal["gathered_bytes_per_rank"] // 3
    fsr = res["format_search_sharded"]
    assert "error" not in fsr and fsr["n_gpus"] == 2 and fsr["ms"] >= fsr["ms_local"] > 0 and sum(fsr["winners"].values()) == 30
    gen = res["generation"]
    assert len(gen) == 6 and all("error" not in g and g["replicas_ok"] == 2 and g["images_per_s"] > 0 for g in gen), gen
    by = {(g["model"], g["path"]): g["ms_per_batch"] for g in gen}
    # the fused paths beat the reference's op sequence (a factor of 5 - 7; Q against F is 1.3 and, with two ranks taking turns on
    # one GPU, not an order this rehearsal can assert)
    assert max(by[("d30-256", "Q")], by[("d30-256", "F")]) < by[("d30-256", "R")]
    assert res["omitted_at_n_gt_1"] == ["cpu_baseline", "other_kernels", "unfused_gpu"]


# ---- the N > 1 control flow, rehearsed on CPU: `bench.py --gpus 2` starts two ranks as a child process, the ranks
# meet over gloo, every rank reaches every collective, rank 0 prints ONE line with n_gpus == 2.  The platform
# (device, backend, the hot-path step) and the two HIP calibration classes are replaced by stand-ins in a launcher
# script of the test's own: this checks bench.py's launcher / rank checks / collectives / JSON line, not the kernels.
_REHEARSAL = '''
import os, sys, time
sys.path.insert(0, {root!r})
import torch
import bench
from fpqvar_amd import calibrate as cal
from oracle import fpq_oracle as orc


def _q(name, w):
    return orc.per_group_kernel_sem(w, "e2m1", 128).half()


class StubLocal:
    def __init__(self, weights, shapes=None, *a, **k):
        self.w = weights

    def quantize(self):
        self.out = {{n: _q(n, w) for n, w in self.w.items()}}


class StubSharded:
    def __init__(self, shapes, weights, group=None, *a, **k):
        self.shapes, self.own = shapes, weights

    def run(self):
        if os.environ.get("REHEARSAL_HANG") and int(os.environ.get("RANK", "0")) == 1:
            time.sleep(3600)                       # rank 1 never reaches the all-gather
        full = {{n: (self.own[n] if n in self.own else torch.full(self.shapes[n], float("nan"))) for n in self.shapes}}
        return cal.calibrate_sharded(full, quantize=_q, exchange="fp16")


class CpuPlatform:
    backend = "gloo"
    data = "REHEARSAL on CPU with stand-ins: the numbers mean nothing"

    def __init__(self, local_rank):
        self.torch, self.dev = torch, torch.device("cpu")

    def init_dist(self, dist):
        dist.init_process_group("gloo")

    def synchronize(self):
        pass

    def empty_cache(self):
        pass

    def hot_path(self, rank):
        return (lambda: time.sleep(0.001)), bench.ROWS * bench.COLS, torch.zeros(1)

    def release_hot_path(self):
        pass

    def generation_replica(self, rank):       # the model-shaped batches are HIP only: fixed stand-in times, slower on higher ranks
        return [dict(model=m, path=p, config="w4a4", images_per_batch=50 if m == "d30-256" else 10, ms_per_batch=10.0 + rank,
                     clock="stand-in") for m, p in bench.GENERATION_PLAN]

    def search_evaluator(self, blocks):       # ... and so is the search: a stand-in evaluation, the real sharding and all-gather
        def evaluate(b):
            time.sleep(0.002)
            return "fp6_e2m3", ("fp6_e3m2" if b % 2 else "fp6_e2m3"), float(b)
        return evaluate

    def codes_calibration(self):      # the packed-exchange variant is HIP only: the rehearsal meets in a collective of the same kind
        import torch.distributed as dist
        t = torch.zeros(dist.get_world_size(), 8)
        dist.all_gather_into_tensor(t.view(-1), torch.ones(8))

    def timer(self):
        t = [0.0, 0.0]
        return (lambda: t.__setitem__(0, time.perf_counter())), (lambda: t.__setitem__(1, time.perf_counter())), \\
               (lambda: (t[1] - t[0]) * 1e3)


cal.LocalShard, cal.ShardedCalibration = StubLocal, StubSharded
bench.CALIB_DEPTH = 1
bench.WATCHDOG_S = float(os.environ.get("REHEARSAL_WATCHDOG_S", "240"))
bench.main(script=os.path.abspath(__file__), platform_factory=CpuPlatform)
'''


def _rehearse(tmp_path, args, env_extra=None, timeout=300):
    script = tmp_path / "bench_rehearsal.py"
    script.write_text(_REHEARSAL.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(script), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                          env=env)


def test_bench_gpus2_launches_two_ranks_gloo(tmp_path):
    out = _rehearse(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["rccl_ranks"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    assert res["scaling"] == "weak" and res["config"]["parallelism"] == "shard2"
    assert "REHEARSAL" in res["data"]
    assert res["value"] > 0 and abs(res["roofline"]["frac"] - res["roofline"]["achieved"] / res["roofline"]["peak"]) < 1e-3
    assert (res["roofline"]["traffic"] is None) == (res["roofline"]["traffic_source"] is None)
    assert res["roofline"]["traffic_source"] is None or "profiles/" in res["roofline"]["traffic_source"]
    wc = res["weight_calibration"]
    assert "error" not in wc, wc
    assert wc["n_gpus"] == 2 and wc["scaling"] == "strong" and wc["elements"] == 12 * 64 * 64
    assert wc["ms"] > 0 and wc["ms_with_all_gather"] > 0 and wc["gathered_bytes_per_rank"] == 2 * wc["elements"]
    assert abs(wc["all_gather_ms"] - max(wc["ms_with_all_gather"] - wc["ms"], 1e-6)) < 5e-3 and wc["all_gather_GBps_per_rank"] >= 0
    assert "fp16" in wc["exchange"] and wc["codes_exchange"]["ms_with_all_gather"] > 0
    assert wc["codes_exchange"]["gathered_bytes_per_rank"] < wc["gathered_bytes_per_rank"]
    assert "cpu_baseline" not in res and "cpu_baseline" in res["omitted_at_n_gt_1"]           # N = 1 only, and the line says so
    assert "gather-bound" in wc["note"]
    # the two legs that shard: present at every N, aggregated over the ranks
    fsr = res["format_search_sharded"]
    assert "error" not in fsr, fsr
    assert fsr["n_gpus"] == 2 and fsr["blocks"] == 30 and fsr["blocks_on_the_busiest_rank"] == 15 and fsr["scaling"] == "strong"
    assert fsr["ms"] >= fsr["ms_local"] > 0 and fsr["winners"] == {"fp6_e2m3/fp6_e2m3": 15, "fp6_e2m3/fp6_e3m2": 15}
    gen = res["generation"]
    assert [(g["model"], g["path"]) for g in gen] == [(m, p) for m in ("d30-256", "d36-512") for p in ("R", "F", "Q")]
    for g in gen:
        assert "error" not in g and g["replicas"] == 2 and g["replicas_ok"] == 2, g
        assert abs(g["ms_per_batch"] - 11.0) < 1e-6                       # the slowest replica (rank 1: 10 + 1 ms)
        assert abs(g["images_per_s"] - 2 * g["images_per_batch"] / 0.011) < 0.1   # images of BOTH ranks / that time


def test_bench_gpus8_launches_eight_ranks_gloo(tmp_path):
    """The driver's largest case, rehearsed on CPU: eight ranks, more ranks than the rehearsal's model has layers (some own
    nothing), one line from rank 0."""
    out = _rehearse(tmp_path, ["--gpus", "8", "--steps", "2", "--warmup", "1"], timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["rccl_ranks"] == 8 and res["config"]["parallelism"] == "shard8"
    assert res["value"] > 0 and res["scaling"] == "weak"
    wc = res["weight_calibration"]
    assert "error" not in wc, wc
    assert wc["n_gpus"] == 8 and wc["elements"] == 12 * 64 * 64 and wc["ms_with_all_gather"] > 0
    fsr, gen = res["format_search_sharded"], res["generation"]
    assert fsr["n_gpus"] == 8 and fsr["blocks_on_the_busiest_rank"] == 4 and sum(fsr["winners"].values()) == 30
    assert len(gen) == 6 and all(g["replicas_ok"] == 8 and abs(g["ms_per_batch"] - 17.0) < 1e-6 for g in gen)


def test_bench_refuses_a_world_size_that_is_not_gpus(tmp_path):
    out = _rehearse(tmp_path, ["--gpus", "2", "--steps", "1", "--warmup", "0"],
                    {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and not out.stdout.strip()
    out = _rehearse(tmp_path, ["--gpus", "0"])
    assert out.returncode != 0


def test_bench_reports_a_blocked_collective_as_failure(tmp_path):
    """A rank that never reaches the all-gather: the line carries the error and the exit code is non-zero."""
    out = _rehearse(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "0"],
                    {"REHEARSAL_HANG": "1", "REHEARSAL_WATCHDOG_S": "8"}, timeout=240)
    assert out.returncode == 3                 # bench.EXIT_COLLECTIVE_TIMEOUT, restored by the launcher from the line
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads(lines[0])
    assert "timeout" in res["weight_calibration"]["error"] and "all_gather" in res["weight_calibration"]["error"]
    assert res["n_gpus"] == 2


@pytest.mark.gpu
def test_generation_steps_record():
    """bench.generation_steps (the bench line's config3_steps / config5_steps): ten steps, the three kernels, bytes and fractions
    that follow from the times, the dual quantizer's scratch back at zero, and the compact summary the line carries."""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    dev = torch.device("cuda:0")
    for model, rows0, cols in (("d30", 100, 1920), ("d36-512", 20, 2304)):
        full = bench.generation_steps(dev, model, "fp32", "rotating", replays=3)
        assert [s["rows"] for s in full["steps"]][0] == rows0 and len(full["steps"]) == 10
        last = full["steps"][-1]
        assert last["adaln"]["bytes"] == last["rows"] * cols * 6 and last["act"]["bytes"] == last["rows"] * cols * 4
        assert last["dual"]["bytes"] == last["rows"] * 4 * cols * 4
        for s in full["steps"]:
            for k in ("adaln", "act", "dual"):
                assert 0.0 < s[k]["us"] < 1e5 and 0.0 < s[k]["frac_of_8TBps"] < 1.0, (model, s["rows"], k, s[k])
                assert abs(s[k]["frac_of_8TBps"] - s[k]["bytes"] / s[k]["us"] / 1e3 / 8000.0) < 2e-3
        # (rates are the bench record's business - boxes of this pool differ by 10 - 15 %; here: sane and self-consistent)
        tw = full["time_weighted_frac_of_8TBps"]
        assert 0.0 < tw < 1.0 and abs(tw - full["bytes_per_block"] / full["block_us_over_the_ten_steps"] / 1e3 / 8000.0) < 1e-3
        bd = full["bound"]
        assert bd is not None and 0.2 < bd["launch_floor_us"] < 50 and bd["launches_per_block_and_step"] == 5
        assert 0.0 < tw <= bd["bound_frac_of_8TBps"] * 1.05 < 1.0, (tw, bd)     # the measured sequence cannot beat its own bound (5 % for noise)
        summ = bench.steps_summary(full)
        assert summ["launch_floor_us"] == bd["launch_floor_us"] and summ["bound_frac"] == bd["bound_frac_of_8TBps"]
        assert summ["time_weighted_frac_of_8TBps"] == tw and len(summ["adaln_us"]) == 10 and set(summ["by_kernel"]) == {"adaln", "act", "dual"}
        json.dumps(summ)
