import sys, json
sys.path.insert(0, "/root/repo")
import torch, bench
o = bench.other_kernels(torch.device("cuda:0"))
print({k.split("_65")[0]: v.get("ms") for k, v in o.items() if "adaln" in k or "rotate" in k})
