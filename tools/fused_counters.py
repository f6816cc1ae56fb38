#!/usr/bin/env python3
"""Driver for the SQ / TCP / TCC counter passes over the fused dequantize + linear kernels: runs each case `reps` times over a small pool of
distinct Q4_K weights.  `tools/fused_counters.sh` wraps it in rocprofv3 --pmc passes and prints the per-launch table.

    python tools/fused_counters.py "mfma:16@18432x3072@1,small@18432x3072@1,mfma:0@12288x3072@32"      (kernel spec @ rows x cols @ rows of x)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402
from tools.fused_sweep import kernel_fn, make_pool  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")
q = pkg.qtypes.Q[os.environ.get("GGQ_COUNTERS_QTYPE", "Q4_K")]
dtype = getattr(torch, os.environ.get("GGQ_COUNTERS_DTYPE", "bfloat16"))
pools = {}
for case in sys.argv[1].split(","):
    spec, shape, m = case.split("@")
    rows, cols = (int(v) for v in shape.split("x"))
    if (rows, cols) not in pools:
        pools[(rows, cols)] = make_pool(pkg, q, rows, cols, dev, min_bytes=150e6)[0]
    x = torch.randn(int(m), cols, device=dev, dtype=dtype) * 0.05
    fn = kernel_fn(pkg, spec, x)
    for w in pools[(rows, cols)]:
        fn(w)
    torch.cuda.synchronize()
