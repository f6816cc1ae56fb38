#!/usr/bin/env python3
"""Which default?  Error of the two ways to run a quantized linear (reference ops.py:242-244), both against an fp64 evaluation on the
ORACLE's weights, on every linear shape of the FLUX.1-dev / SD3.5-large / T5-xxl manifests:

  (a) default path: dequantize_tensor (bit-exact weight) + ``F.linear`` (hipBLASLt / rocBLAS pick their own fp32 summation order);
  (b) fused path:   ``ggq_linear_small`` (1-4 rows) / ``ggq_linear_mfma`` (more rows) -- the same weights bit for bit, fp32 accumulation
      in the kernel's own order, ONE rounding to the output dtype.

Neither order is a contract of the reference: the reference calls ``F.linear`` and takes whatever the BLAS library does.  So the fused
path may be the default iff it is no further from the exact result than (a) is, and as deterministic.  Per case (shape, rows of x,
dtype) this prints: max and RMS error of (a) and of (b) relative to RMS(exact), the share of outputs on which the two agree bit for
bit, and whether a second run of each reproduces the first.  ``bench.py --workload fused-error`` wraps the same function; the committed
table is profiles/r05_fused_error.json.

    python tools/fused_error.py [--m 1,4,64,256] [--dtypes bf16,f16] [--models flux,sd35,t5] [--quick]
"""
import argparse
import json
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ggq_pkg import load_package  # noqa: E402

DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}


def linear_shapes(pkg, models=("flux", "sd35", "t5")):
    """Distinct (model, qtype, rows, cols) of the 2-D linear weights (embedding tables are not linears)."""
    M = pkg.manifests
    src = {"flux": M.flux_dev, "sd35": M.sd35_large, "t5": M.t5_xxl_encoder}
    seen, out = set(), []
    for model in models:
        for name, q, (rows, cols) in src[model]():
            if "token_embd" in name or (q, rows, cols) in seen:
                continue
            seen.add((q, rows, cols))
            out.append((model, re.sub(r"\.\d+\.", ".", name[:-len(".weight")] if name.endswith(".weight") else name), q, rows, cols))
    return out


def _stats(y, exact, scale):
    e = (y.double() - exact).abs()
    return float(e.max() / scale), float(e.pow(2).mean().sqrt() / scale)


def measure(pkg, device, ms=(1, 4, 64, 256), dtypes=("bf16", "f16"), models=("flux", "sd35", "t5"), shapes=None, seed=11, with_bias=True):
    """Returns {"cases": [...], "summary": {...}}.  Imports the oracle: measurement / test infrastructure only."""
    import oracle
    F = torch.nn.functional
    fused, dq, T = pkg.fused, pkg.dequant, pkg.ops.GGMLTensor
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    cases = []
    # which C entry point the policy really called: the two bindings are wrapped for the duration of the measurement
    if fused._small_call is None:
        fused._bind()
    real_small, real_mfma, real_ws, ran = fused._small_call, fused._mfma_call, fused._mfma_ws_call, []
    fused._small_call = lambda *a: (ran.append("ggq_linear_small"), real_small(*a))[1]
    fused._mfma_call = lambda *a: (ran.append("ggq_linear_mfma"), real_mfma(*a))[1]
    fused._mfma_ws_call = lambda *a: (ran.append("ggq_linear_mfma"), real_ws(*a))[1]          # (the same kernels with K also split across workgroups)
    try:
        return _measure(pkg, device, ms, dtypes, models, shapes, seed, with_bias, ran, T, F, fused, dq, oracle, gen, cases)
    finally:
        fused._small_call, fused._mfma_call, fused._mfma_ws_call = real_small, real_mfma, real_ws


def _measure(pkg, device, ms, dtypes, models, shapes, seed, with_bias, ran, T, F, fused, dq, oracle, gen, cases):
    for si, (model, layer, q, rows, cols) in enumerate(shapes or linear_shapes(pkg, models)):
        bs, ts = pkg.qtypes.block_geometry(q)
        packed = pkg.synth.device_blocks(q, pkg.synth.n_blocks_for(q, rows * cols), device, seed + si)
        w = T(packed.reshape(-1), tensor_type=q, tensor_shape=(rows, cols))
        host = packed.reshape(-1).cpu().numpy()
        w16 = torch.from_numpy(oracle.dequant_f16(q, host, simd=oracle.simd_available()).view(np.int16).reshape(rows, cols)).to(device).view(torch.float16)
        for dname in dtypes:
            dtype = DTYPES[dname]
            wd = w16.to(dtype)                                   # dequant.py:23, the reference's own cast of the oracle's fp16 weight
            w64 = wd.double()
            bias = (torch.randn(rows, device=device, generator=gen) * 0.02).to(dtype) if with_bias else None
            for m in ms:
                x = torch.randn(m, cols, device=device, generator=gen).to(dtype)
                exact = x.double() @ w64.T
                if bias is not None:
                    exact = exact + bias.double()
                scale = float(exact.pow(2).mean().sqrt())
                # (a) the default path, twice
                ya = [F.linear(x, dq.dequantize_tensor(w, dtype), bias) for _ in range(2)]
                # (b) the fused path, twice -- what install(fast) would run for this many rows; None when the kernel declines the shape
                ran.clear()
                try:
                    yb = [fused.linear_auto(x, w, bias) for _ in range(2)]     # the default's own policy (fused.py): GEMV at one row, the MFMA kernels above
                except dq.GGQUnsupported as e:
                    cases.append({"model": model, "layer": layer, "qtype": q.name, "rows": rows, "cols": cols, "m": m, "dtype": dname,
                                  "fused": None, "declined": str(e)[:80]})
                    continue
                a_max, a_rms = _stats(ya[0], exact, scale)
                b_max, b_rms = _stats(yb[0], exact, scale)
                cases.append({"model": model, "layer": layer, "qtype": q.name, "rows": rows, "cols": cols, "m": m, "dtype": dname,
                              "fused": ran[-1],
                              "default_max": a_max, "default_rms": a_rms, "fused_max": b_max, "fused_rms": b_rms,
                              "same_bits_share": float((ya[0] == yb[0]).double().mean()),
                              "default_deterministic": bool(torch.equal(ya[0], ya[1])), "fused_deterministic": bool(torch.equal(yb[0], yb[1]))})
        del w16, packed, w
        torch.cuda.empty_cache()
    done = [c for c in cases if c.get("fused")]
    eps = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}                 # one unit in the last place of the output dtype, relative
    summary = {
        "cases": len(cases), "fused_ran": len(done), "declined": len(cases) - len(done),
        "worst_rms_ratio_fused_over_default": max((c["fused_rms"] / c["default_rms"] for c in done), default=None),
        "worst_max_ratio_fused_over_default": max((c["fused_max"] / c["default_max"] for c in done), default=None),
        "worst_max_excess_in_output_ulps": max(((c["fused_max"] - c["default_max"]) / eps[c["dtype"]] for c in done), default=None),
        "fused_rms_not_above_default_in": sum(c["fused_rms"] <= c["default_rms"] * 1.0000001 for c in done),
        "fused_nondeterministic": sum(not c["fused_deterministic"] for c in done),
        "default_nondeterministic": sum(not c["default_deterministic"] for c in done),
        "min_same_bits_share": min((c["same_bits_share"] for c in done), default=None),
        "note": "errors relative to RMS(exact), exact = fp64 product on the oracle's weights cast to the dtype the reference's way; "
                "max excess in ulps: (fused_max - default_max) / (2^-8 bf16 | 2^-11 fp16) -- relative to RMS(exact), an output near that "
                "magnitude has a rounding step of about that size",
    }
    # the rule install()'s default stands on: fused RMS error within 2 % of the default path's in every case, its max error within one output
    # rounding step of the default path's, and it reproduces itself run to run
    summary["fused_no_worse"] = bool(done) and summary["worst_rms_ratio_fused_over_default"] <= 1.02 \
        and summary["worst_max_excess_in_output_ulps"] <= 1.0 and summary["fused_nondeterministic"] == 0
    return {"cases": cases, "summary": summary}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--m", default="1,4,64,256")
    ap.add_argument("--dtypes", default="bf16,f16")
    ap.add_argument("--models", default="flux,sd35,t5")
    ap.add_argument("--quick", action="store_true", help="three shapes only")
    args = ap.parse_args()
    pkg = load_package()
    dev = torch.device("cuda:0")
    shapes = None
    models = tuple(args.models.split(","))
    if args.quick:
        shapes = linear_shapes(pkg, models)[:3]
    out = measure(pkg, dev, tuple(int(v) for v in args.m.split(",")), tuple(args.dtypes.split(",")), models, shapes)
    out["device"] = torch.cuda.get_device_name(dev)
    out["torch"] = torch.__version__
    out["library_build"] = pkg._native.lib().ggq_build_id().decode()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
