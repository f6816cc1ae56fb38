#!/usr/bin/env python3
"""Driver for the SQ / TCP / TCC counter passes over the shared-tile fused GEMM (ggq::linear_tile, csrc/ggq_gemm.hpp): Q4_K 12288x3072,
bf16, 4608 and 1024 rows of x; `tools/gemm_counters.sh` wraps it in rocprofv3 --pmc passes and prints the per-launch table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")
q = pkg.qtypes.Q[os.environ.get("GGQ_COUNTERS_QTYPE", "Q4_K")]
rows, cols = 12288, 3072
pool = [pkg.ops.GGMLTensor(pkg.synth.device_blocks(q, pkg.synth.n_blocks_for(q, rows * cols), dev, 10 + i), tensor_type=q, tensor_shape=(rows, cols)) for i in range(6)]
for m in (4608, 1024):
    x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16) * 0.05
    for w in pool:
        pkg.fused.linear_mfma(x, w, tile_rows=256)
    torch.cuda.synchronize()
