#!/usr/bin/env python3
"""Which per-cell team-shape exceptions of csrc/ggq_capi.hip survive two boxes (Next #7).

    python tools/mode_prune.py gpurun_out/r6k gpurun_out/r6l        (one directory per lease: mode_{shipped,coopall,soloonly}_{1,2}.json from tools/mode_table.py --arith)

For every (format, arithmetic -> output) cell: GB/s of the all-workgroup-team build over the all-one-wave-team build, mean of the two alternations, per box.  A cell keeps (or gets)
an exception to the general rule only if the ratio is on the same side of 1 by >= 2 % on BOTH boxes; prints the table and the resulting exception list as JSON."""
import json
import os
import sys


def load(d):
    out = {}
    for v in ("shipped", "coopall", "soloonly"):
        runs = [json.load(open(os.path.join(d, f"mode_{v}_{i}.json"))) for i in (1, 2) if os.path.exists(os.path.join(d, f"mode_{v}_{i}.json"))]
        out[v] = {f: {c: sum(r[f][c] for r in runs) / len(runs) for c in runs[0][f]} for f in runs[0]}
        out[v + "_spread"] = max(abs(runs[0][f][c] / runs[-1][f][c] - 1) for f in runs[0] for c in runs[0][f]) if len(runs) > 1 else None
    return out


def general_rule(cell):
    ar, out = cell.split("->")
    return "coop" if (out == "f32" or ar == "f16") else "solo"


def main():
    boxes = [load(d) for d in sys.argv[1:]]
    table, keep = {}, {}
    for f in boxes[0]["coopall"]:
        for c in boxes[0]["coopall"][f]:
            ratios = [b["coopall"][f][c] / b["soloonly"][f][c] for b in boxes]
            ship = [b["shipped"][f][c] / max(b["coopall"][f][c], b["soloonly"][f][c]) for b in boxes]
            table[f"{f} {c}"] = {"coop_over_solo": [round(r, 3) for r in ratios], "shipped_over_best": [round(r, 3) for r in ship], "general_rule": general_rule(c)}
            if all(r >= 1.02 for r in ratios):
                want = "coop"
            elif all(r <= 0.98 for r in ratios):
                want = "solo"
            else:
                want = general_rule(c)
            table[f"{f} {c}"]["choice"] = want
            if want != general_rule(c):
                keep[f"{f} {c}"] = want
    print(json.dumps({"boxes": sys.argv[1:], "alternation_spread": [[b[v + "_spread"] for v in ("shipped", "coopall", "soloonly")] for b in boxes],
                      "cells": table, "exceptions_that_survive": keep}, indent=1))


if __name__ == "__main__":
    main()
