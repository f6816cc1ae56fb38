#!/usr/bin/env python3
"""Fused dequantize + GEMM on the matrix cores (fused.linear_mfma) against the drop-in default (dequantize_tensor + F.linear =
our unpack kernel + hipBLASLt) and against F.linear on dense weights kept resident, per layer shape and number of rows of x.

    python tools/mfma_linear_bench.py [--qtype Q4_K] [--dtype bfloat16] [--m 8,32,128,512,4608] [--tiles 0,64,128,256]

Every timing rotates over a pool of distinct weights (dense bytes of the pool > the 256 MiB Infinity Cache) and is the HIP-event
time of `reps` back-to-back calls, i.e. it includes the host's issue rate, as a layer inside a model would see it.  One JSON line."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402

SHAPES = [(3072, 3072), (9216, 3072), (3072, 12288), (12288, 3072), (21504, 3072), (4096, 4096), (10240, 4096)]


def timed(fn, pool, reps):
    for w in pool[:2]:
        fn(w)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(pool[i % len(pool)])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3            # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qtype", default="Q4_K")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--m", default="8,32,64,128,256,512,1024,4608")
    ap.add_argument("--tiles", default="0,128,256")
    ap.add_argument("--shapes", default="")
    args = ap.parse_args()
    pkg = load_package()
    dev = torch.device("cuda:0")
    q = pkg.qtypes.Q[args.qtype]
    dtype = getattr(torch, args.dtype)
    bs, ts = pkg.qtypes.block_geometry(q)
    shapes = [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")] if args.shapes else SHAPES
    out = {"qtype": q.name, "dtype": args.dtype, "unit": "us per call", "rows": []}
    g = torch.Generator(device=dev).manual_seed(0)
    for rows, cols in shapes:
        n_pool = max(4, min(24, int(400e6 // (rows * cols * 2)) + 1))
        pool = []
        for i in range(n_pool):
            data = torch.randint(0, 256, (rows * cols // bs, ts), dtype=torch.uint8, device=dev, generator=g)
            for off in pkg.qtypes.SCALE_FIELDS[q]:
                vals = (torch.rand(data.shape[0], device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
                data[:, off:off + 2] = vals.view(torch.uint8).reshape(-1, 2)
            pool.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
        dense = [pkg.dequant.dequantize_tensor(w, dtype) for w in pool]
        for m in (int(v) for v in args.m.split(",")):
            x = torch.randn(m, cols, device=dev, dtype=dtype) * 0.05
            reps = max(20, min(200, int(2e11 // (2.0 * m * rows * cols)) + 1))
            row = {"weight": f"{rows}x{cols}", "m": m, "GFLOP": round(2.0 * m * rows * cols / 1e9, 2)}
            row["dequant+F.linear"] = round(timed(lambda w: torch.nn.functional.linear(x, pkg.dequant.dequantize_tensor(w, dtype)), pool, reps), 1)
            row["F.linear dense-resident"] = round(timed(lambda w: torch.nn.functional.linear(x, w), dense, reps), 1)
            best = None
            for t in (int(v) for v in args.tiles.split(",")):
                if t and t > 32 and t >= 2 * m and t != 32:
                    continue                                     # a tile more than twice m only wastes MFMAs
                us = round(timed(lambda w: pkg.fused.linear_mfma(x, w, tile_rows=t, auto_max_rows=None), pool, reps), 1)
                row[f"fused tile={t or 'auto'}"] = us
                best = us if best is None else min(best, us)
            row["fused_best_vs_default"] = round(row["dequant+F.linear"] / best, 2)
            row["fused_best_TFLOPs"] = round(2.0 * m * rows * cols / best / 1e6, 1)
            out["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
        del pool, dense
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
