"""The rule install()'s default stands on (Next #1): on FLUX.1-dev / SD3.5-large / T5-xxl linear shapes the fused dequantize + linear
kernels are NO FURTHER from an fp64 evaluation on the oracle's weights than the default path (bit-exact unpack + F.linear), and reproduce run to run.
The full table (every distinct shape x {1, 4, 64, 256} rows x {bf16, fp16}) is profiles/r05_fused_error.json, produced by the same function
(tools/fused_error.py measure()); here a sample of the shapes runs in every `pytest -m gpu`."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fused_error", os.path.join(ROOT, "tools", "fused_error.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fused_linears_are_no_further_from_fp64_than_unpack_plus_f_linear(pkg):
    T = _tool()
    every = T.linear_shapes(pkg)
    # one shape per (model, kernel-relevant geometry): FLUX modulation (18432x3072, the 1-row case), FLUX proj, FLUX mlp.2 (cols 12288: a row wider than
    # ggq_linear_small's LDS staging -- declined at 1 / 4 rows), SD3.5 fc2 (cols 9728), SD3.5 qkv (Q5_0, cols 2432 = 9.5 spans: the MFMA kernel's short
    # last span), T5 ffn_down (cols 10240)
    pick = [s for s in every if (s[3], s[4]) in ((18432, 3072), (3072, 3072), (3072, 12288), (2432, 9728), (7296, 2432), (4096, 10240))]
    assert len(pick) == 6
    out = T.measure(pkg, torch.device("cuda:0"), ms=(1, 4, 64, 256), dtypes=("bf16", "f16"), shapes=pick)
    s = out["summary"]
    # declined, all at 256 rows and in both dtypes: FLUX's modulation weight (18432 x 3072), FLUX mlp.2 (3072 x 12288) and T5's ffn_down (4096 x 10240) -- above 128 rows the auto
    # policy hands a call back to unpack + F.linear once rows of x * rows * columns exceeds fused.AUTO_MAX_MACS (8e9; 5e9 for Q5_0 / Q8_0 / Q3_K / Q6_K: SD3.5's 7296 x 2432
    # Q5_0 qkv at 256 rows is 4.5e9 and stays fused).
    # (FLUX mlp.2 at 1 / 4 rows -- a row wider than ggq_linear_small's LDS staging, declined in round 5 -- is served by the 16-row MFMA kernel since round 6.)
    assert s["cases"] == 48 and s["fused_ran"] == 42 and s["declined"] == 6
    assert s["fused_nondeterministic"] == 0
    assert s["worst_rms_ratio_fused_over_default"] <= 1.02, s
    assert s["worst_max_excess_in_output_ulps"] <= 1.0, s
    assert s["min_same_bits_share"] >= 0.98, s                                         # and in fact almost every output is the very same number
    assert s["fused_no_worse"] is True
    for c in out["cases"]:
        if c.get("fused"):
            # the default's policy (fused.linear_auto): one row -> the GEMV unless the weight is 16384+ rows tall or its rows do not fit the GEMV's LDS staging
            bs, ts = pkg.qtypes.block_geometry(pkg.qtypes.Q[c["qtype"]])
            small = c["m"] == 1 and c["rows"] < 16384 and c["cols"] // bs * ts + 15 <= 6 * 64 * 16      # (csrc/ggq_linear.hpp LIN_SLICE)
            assert c["fused"] == ("ggq_linear_small" if small else "ggq_linear_mfma"), (c["qtype"], c["rows"], c["cols"], c["m"], c["dtype"], c["fused"])


def test_the_committed_table_says_what_the_default_claims():
    """The newest profiles/rNN_fused_error.json is what install.DEFAULT_FAST cites: it must cover all three models and carry the verdict."""
    import glob
    import json
    d = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_fused_error.json")))[-1]))
    summary = d.get("summary") or d["config"]["summary"]                    # tools/fused_error.py's own output, or the line of `bench.py --workload fused-error`
    assert summary["fused_no_worse"] is True and summary["cases"] == 128
    assert {c["model"] for c in d["cases"]} == {"flux", "sd35", "t5"} and {c["m"] for c in d["cases"]} == {1, 4, 64, 256}
