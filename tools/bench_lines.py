#!/usr/bin/env python3
"""Print the headline fraction and the producers' times of bench.py lines: bench_lines.py <file.json> ..."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    o = d.get("other_kernels", {})
    print(f, d["roofline"]["frac"], d["roofline"]["kernel_ms"],
          {k.split("_65")[0]: v.get("ms") for k, v in o.items() if "adaln" in k or "rotate" in k})
