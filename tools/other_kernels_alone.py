#!/usr/bin/env python3
"""bench.py's other_kernels alone in a fresh process (FPQ_BENCH_GROUPS=act16,operands,... selects groups,
FPQ_BENCH_BURST=timed,lead the burst protocol): prints the selected groups' records."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
o = bench.other_kernels(torch.device("cuda:0"))
print(json.dumps(o))
