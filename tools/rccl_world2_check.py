"""The sharded calibration over RCCL with world size 2 (fp16 and codes exchange, in-place all_gather_into_tensor), bit-equal to
per-layer launches - on a node with two GPUs.  On the one-GPU boxes this build had, both ranks land on device 0 and RCCL
refuses ("Duplicate GPU detected : rank 0 and rank 1 both on CUDA device", NCCL 2.26.6 - run of round 3): the N > 1 path is
covered by gloo ranks on CPU and by three-rank replays on one GPU (tests/test_gpu_configs.py) instead.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_world2_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import datetime
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
local = int(os.environ.get("LOCAL_RANK", rank))
DEV = local if torch.cuda.device_count() > local else 0
torch.cuda.set_device(DEV)
try:
    if os.environ.get("FPQ_CHECK_BACKEND", "nccl") == "gloo":   # two ranks on one GPU: gloo takes device tensors, RCCL refuses the duplicate device
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    else:
        dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=60), device_id=torch.device("cuda", DEV))
    slab = torch.full((dist.get_world_size(), 1024), float(rank + 1), device=f"cuda:{DEV}", dtype=torch.float16)
    dist.all_gather_into_tensor(slab.view(-1), slab[rank])
    torch.cuda.synchronize()
    print(f"rank {rank}: in-place all_gather_into_tensor ok: rows {slab[:, 0].tolist()}", flush=True)
    if os.environ.get("FPQ_CHECK_ONLY", "") in ("", "calibration"):
        from fpqvar_amd import calibrate as cal
        shapes = {f"l{i}": s for i, s in enumerate(((384, 128), (128, 512), (640, 256), (256, 128), (1024, 384)))}
        g = torch.Generator().manual_seed(3)
        w = {n: (torch.randn(*s, generator=g) * 0.02).to(f"cuda:{DEV}") for n, s in shapes.items()}
        from fpqvar_amd import ops
        want = {n: ops.quant_rows(w[n], "e2m1", 128, torch.float16) for n in shapes}
        got = cal.calibrate_sharded(w)
        ok16 = all(torch.equal(got[n].view(torch.int16), want[n].view(torch.int16)) for n in shapes)
        gotc = cal.calibrate_sharded(w, exchange="codes")
        okc = all(torch.equal(gotc[n].view(torch.int16), want[n].view(torch.int16)) for n in shapes)
        print(f"rank {rank}: sharded calibration over {dist.get_backend()}, world {dist.get_world_size()}: fp16 exchange bit-equal {ok16}, codes exchange bit-equal {okc}", flush=True)
    # the format search sharded by block (search/search_fp6_format.py's per-block loop): block b on rank b mod world, every
    # rank evaluates with the fused quantizers on ITS device tensors, one all-gather of (loss, formats) triples
    from fpqvar_amd import format_search as fs

    config4 = os.environ.get("FPQ_CHECK_SEARCH") == "config4"   # BASELINE config 4's size: a d30 mat_qkv layer, 100 dumped samples

    def layer(b):
        if config4:   # search/search_fp6_format.py:576-608: w [5760 x 1920], x_j [2, pn^2, 1920] over the ten scale steps
            gg = torch.Generator(device=f"cuda:{DEV}").manual_seed(400 + b)
            pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
            xs = [torch.randn(2, pns[j % 10] ** 2, 1920, device=f"cuda:{DEV}", generator=gg).half() for j in range(100)]
            wt = (torch.randn(5760, 1920, device=f"cuda:{DEV}", generator=gg) * 0.02).half()
            return xs, wt
        gg = torch.Generator().manual_seed(100 + b)
        xs = [torch.randn(2, 16 * (j + 1), 256, generator=gg).half().to(f"cuda:{DEV}") for j in range(3)]
        wt = (torch.randn(384, 256, generator=gg) * 0.05).half().to(f"cuda:{DEV}")
        return xs, wt

    def evaluate(b):
        xs, wt = layer(b)
        wf, af, losses = fs.search_layer(xs, wt, fs.FP6_FORMATS)
        return wf, af, losses[(wf, af)]
    n_blocks = 4 if config4 else 5
    got_s = fs.search_blocks_sharded(n_blocks, evaluate, fs.FP6_FORMATS)
    want_s = [evaluate(b) for b in range(n_blocks)]
    ok_s = all(g[0] == w_[0] and g[1] == w_[1] and abs(g[2] - w_[2]) <= 1e-6 * abs(w_[2]) for g, w_ in zip(got_s, want_s))
    print(f"rank {rank}: format search sharded over {dist.get_world_size()} ranks equals the single-process result: {ok_s}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(1)
