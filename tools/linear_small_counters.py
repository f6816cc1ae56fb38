#!/usr/bin/env python3
"""Driver for SQ counter passes over the fused 1-4-row linear (ggq::linear_small): Q4_K 18432x3072 and 9216x3072, m = 1 and 4, a rotating pool of
weights; `tools/linear_small_counters.sh` wraps it in rocprofv3 --pmc passes and prints the per-launch table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")
q = pkg.qtypes.Q.Q4_K
bs, ts = pkg.qtypes.block_geometry(q)
g = torch.Generator(device=dev).manual_seed(0)
for rows, cols in ((18432, 3072), (9216, 3072)):
    pool = []
    for i in range(16):
        data = torch.randint(0, 256, (rows * cols // bs, ts), dtype=torch.uint8, device=dev, generator=g)
        for off in pkg.qtypes.SCALE_FIELDS[q]:
            vals = (torch.rand(data.shape[0], device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
            data[:, off:off + 2] = vals.view(torch.uint8).reshape(-1, 2)
        pool.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
    for m in (1, 4):
        x = torch.randn(m, cols, device=dev, dtype=torch.bfloat16) * 0.05
        for w in pool:
            pkg.fused.linear_small(x, w)
        torch.cuda.synchronize()
    del pool
