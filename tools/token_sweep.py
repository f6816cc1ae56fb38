#!/usr/bin/env python3
"""The table INTEGRATION.md section 4 quotes: an emulated denoising step (tools/flux_forward_emulation.py: every quantized linear of the model,
dequantize + F.linear + drop, back to back) at several token counts, for

    exact      install(exact=True) / GGQ_EXACT=1: unpack + F.linear everywhere (the default of rounds 1-4)
    default    install(): fused_small_m + fused_mfma (<= 256 rows) on top            (the default since round 5)
    dense      the same GEMMs on weights dequantized once and kept resident          (what `dense_cache_gb` buys, for its GB)

One box, one process per cell (a fresh allocator each), medians of --reps steps.  Prints ONE JSON object; commit it under profiles/.

    python tools/token_sweep.py [--model flux] [--tokens 64,256,512,1024,2304,4608] [--reps 7]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(model, tokens, reps, extra):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "flux_forward_emulation.py"), "--model", model, "--tokens", str(tokens), "--reps", str(reps)] + extra
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if proc.returncode:
        raise SystemExit(proc.stderr[-2000:])
    return json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="flux", choices=["flux", "sd35", "t5"])
    ap.add_argument("--tokens", default="64,256,512,1024,2304,4608")
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    rows = {}
    for t in [int(v) for v in args.tokens.split(",")]:
        exact = run(args.model, t, args.reps, [])
        fast = run(args.model, t, args.reps, ["--fused-small-m", "--fused-mfma", "256"])
        dense = sorted([exact["ms_per_step_dense_resident"], fast["ms_per_step_dense_resident"]])
        rows[str(t)] = {"exact_ms": exact["ms_per_step_dequant_on_the_fly"], "default_ms": fast["ms_per_step_dequant_on_the_fly"],
                        "dense_resident_ms": fast["ms_per_step_dense_resident"], "dense_resident_ms_both_runs": dense,
                        "default_minus_dense_ms": round(fast["ms_per_step_dequant_on_the_fly"] - fast["ms_per_step_dense_resident"], 2),
                        "exact_minus_dense_ms": round(exact["ms_per_step_dequant_on_the_fly"] - exact["ms_per_step_dense_resident"], 2)}
    import torch
    print(json.dumps({"workload": exact["workload"].split(",")[0] + f", {args.model}, bf16, medians of {args.reps} emulated steps per cell",
                      "columns": {"exact": "install(exact=True): unpack + F.linear everywhere", "default": "install(): fused_small_m + fused_mfma<=256 (round-5 default)",
                                  "dense_resident": "the same GEMMs on dense weights kept resident"},
                      "by_tokens": rows, "device": torch.cuda.get_device_name(0)}))


if __name__ == "__main__":
    main()
