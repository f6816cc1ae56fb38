#!/usr/bin/env python3
"""GPU time of the fused dequantize + linear kernels per layer shape, rows of x and kernel choice, replayed from a captured HIP graph
(no host issue rate in the figure), over a rotating pool of distinct weights whose PACKED bytes exceed the 256 MiB Infinity Cache.

    python tools/fused_sweep.py [--qtype Q4_K] [--dtype bfloat16] [--m 1,4,8,16,32,64] [--kernels small,mfma:0,mfma:16,mfma:32,mfma:64]
                                [--shapes 12288x3072,...] [--reps 5]

One JSON line per (shape, m) on stderr as it goes, the whole table on stdout.  `packed_GBps` = packed bytes of one weight / time:
the roofline these kernels are priced against is the packed-read rate (8 TB/s spec), DESIGN.md section 4d.

Run it over EVERY format before closing a round (lab/sessions/r6r.sh): the Q4_K-only sweeps of rounds 2-6 could not see Q3_K's 16-way LDS bank conflict, Q5_0's 4-way one
or Q8_0's register spills (EXPERIMENTS.md R6-9), and SQ_LDS_BANK_CONFLICT reads 0 on this pool."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402

SHAPES = [(12288, 3072), (18432, 3072), (9216, 3072), (3072, 3072), (3072, 12288), (21504, 3072), (3072, 15360)]


def make_pool(pkg, q, rows, cols, dev, min_bytes=420e6, seed=0):
    bs, ts = pkg.qtypes.block_geometry(q)
    packed = rows * cols // bs * ts
    n = max(4, min(64, int(min_bytes // packed) + 1))
    g = torch.Generator(device=dev).manual_seed(seed)
    pool = []
    for _ in range(n):
        data = torch.randint(0, 256, (rows * cols // bs, ts), dtype=torch.uint8, device=dev, generator=g)
        for off in pkg.qtypes.SCALE_FIELDS[q]:
            vals = (torch.rand(data.shape[0], device=dev, generator=g) * 1e-3 + 1e-4).to(torch.float16)
            data[:, off:off + 2] = vals.view(torch.uint8).reshape(-1, 2)
        pool.append(pkg.ops.GGMLTensor(data.reshape(-1), tensor_type=q, tensor_shape=(rows, cols)))
    return pool, packed


def graph_time_us(fn, pool, reps):
    """us per call: the calls over the whole pool captured once, replayed `reps` times between two events."""
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        keep = [fn(w) for w in pool]                       # warm-up: lazy module loads, allocator
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            keep = [fn(w) for w in pool]
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    del keep
    return a.elapsed_time(b) * 1e3 / (reps * len(pool))


def kernel_fn(pkg, spec, x):
    if spec == "small":
        return lambda w: pkg.fused.linear_small(x, w)
    if spec == "default":
        return lambda w: torch.nn.functional.linear(x, pkg.dequant.dequantize_tensor(w, x.dtype))
    kind, _, tile = spec.partition(":")
    assert kind == "mfma", spec
    return lambda w: pkg.fused.linear_mfma(x, w, tile_rows=int(tile or 0), auto_max_rows=None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qtype", default="Q4_K")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--m", default="1,4,8,16,32,64")
    ap.add_argument("--kernels", default="small,mfma:0,mfma:16")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    pkg = load_package()
    dev = torch.device("cuda:0")
    q = pkg.qtypes.Q[args.qtype]
    dtype = getattr(torch, args.dtype)
    shapes = [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")] if args.shapes else SHAPES
    out = {"qtype": q.name, "dtype": args.dtype, "unit": "us per call (graph replay)", "lib": os.environ.get("GGQ_HIP_LIB", "in-tree"),
           "env": {k: v for k, v in os.environ.items() if k.startswith("GGQ_")}, "rows": []}
    for rows, cols in shapes:
        pool, packed = make_pool(pkg, q, rows, cols, dev)
        for m in (int(v) for v in args.m.split(",")):
            x = torch.randn(m, cols, device=dev, dtype=dtype) * 0.05
            row = {"weight": f"{rows}x{cols}", "m": m, "packed_MB": round(packed / 1e6, 2)}
            for spec in args.kernels.split(","):
                if spec == "small" and m > pkg.fused.MAX_ROWS:
                    continue
                if spec == "mfma:16" and m > 32:
                    continue
                try:
                    us = graph_time_us(kernel_fn(pkg, spec, x), pool, args.reps)
                except pkg.dequant.GGQUnsupported as e:
                    row[spec] = f"declined: {e}"[:80]
                    continue
                row[spec] = round(us, 2)
                row[spec + " packed_GBps"] = round(packed / us / 1e3, 1)
            out["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
        del pool
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
