#!/usr/bin/env python3
"""GB/s of every format in every output dtype (and, with --arith, every arithmetic mode) on the bench pool
(64 x (3072x3072 + 3072x12288), one plan launch per measurement, HIP events) -- the table the per-mode team shapes in
csrc/ggq_capi.hip are chosen from.  Run it with two builds of the library alternately on the same box:

    GGQ_HIP_LIB=/path/to/variant.so python tools/mode_table.py ; python tools/mode_table.py

Prints one JSON object {format: {"f16->bf16": GB/s, ...}}."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ggq_pkg import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--formats", default="", help="comma-separated subset (default: all)")
    ap.add_argument("--arith", action="store_true", help="also the bf16 / fp32 arithmetic modes")
    ap.add_argument("--outs", default="f16,bf16,f32", help="comma-separated subset of the output dtypes")
    args = ap.parse_args()
    pkg = load_package()
    dev = torch.device("cuda:0")
    names = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
    want = [f for f in args.formats.split(",") if f]
    table = {}
    for q in pkg.qtypes.HIP_QTYPES:
        if want and q.name not in want:
            continue
        base = bench.build_pool(pkg, pkg.manifests.flux_linear_pool(q, args.pairs), dev, seed0=1000 * int(q))
        shapes = [tuple(o.shape) for o in base.outputs]
        row = {}
        for cd in ((torch.float16, torch.bfloat16, torch.float32) if args.arith else (torch.float16,)):
            for od in (torch.float16, torch.bfloat16, torch.float32):
                if names[od] not in args.outs.split(","):
                    continue
                p = pkg.grouped.DequantPlan([(d, q, sh) for d, sh in zip(base._keep, shapes)], out_dtype=od, dequant_dtype=cd)
                ms, _ = bench.timed_steps(p, args.steps, 3, dev, lambda: torch.cuda.synchronize(dev))
                row[f"{names[cd]}->{names[od]}"] = round(p.bytes / (ms / args.steps * 1e-3) / 1e9, 1)
                p.close()
                del p
                torch.cuda.empty_cache()
        table[q.name] = row
        base.close()
        del base
        torch.cuda.empty_cache()
    print(json.dumps(table))


if __name__ == "__main__":
    main()
