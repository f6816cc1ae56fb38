#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tests/microbench/pmc.sh (gpurun_out/pmc/p{1,2,3}) into the
table committed under profiles/: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes),
corrected as MI355X_MICROARCH.md prescribes and calibrated on the known-size streams of the same pass.

    python tools/pmc_summarize.py gpurun_out/pmc [--json profiles/pmc_traffic.json] > profiles/<name>.txt

--json also writes the table bench.py reports ``roofline.traffic`` from: corrected bytes per launch per (format[, mode]) plus
``_build_id`` = the identity of the kernel sources the counters were collected on (``_native.source_id()`` of THIS tree: the
harness compiles the same csrc/ggq_capi.hip the library is built from), which bench.py compares with the loaded library's
``ggq_build_id()`` before it reports a figure.
"""
import collections
import csv
import re
import sys


def counters(path, name):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def durations(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in d.items()}


TS = {"FmtQ4_K": 144 / 256, "FmtQ4_0": 18 / 32, "FmtQ6_K": 210 / 256, "FmtQ8_0": 34 / 32, "FmtQ5_0": 22 / 32, "FmtQ2_K": 84 / 256, "FmtQ3_K": 110 / 256}
ELEMENTS = 64 * (3072 * 3072 + 3072 * 12288)
DT = {"0": "f16", "1": "bf16", "2": "f32"}


def main(root, json_out=None):
    traffic = {}
    fetch = counters(f"{root}/p1/p_counter_collection.csv", "FETCH_SIZE")
    write = counters(f"{root}/p2/p_counter_collection.csv", "WRITE_SIZE")
    dur = durations(f"{root}/p1/p_kernel_trace.csv")
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), MI355X; bash tests/microbench/pmc.sh")
    print("# pools: 64 x (3072x3072 + 3072x12288) per format, shipped kernels via ggq_plan_launch; counter = mean over the launches of that kernel")
    print("# correction: read bytes = FETCH_SIZE * 2048 (gfx950 counts 16 B/lane streams at half rate; calibrated below on 1 GiB copies), write bytes = WRITE_SIZE * 1024")
    print(f"{'kernel (compute->out)':28s} {'FETCH_SIZE':>12s} {'WRITE_SIZE':>12s} {'read B':>14s} {'algorithmic':>14s} {'ratio':>7s} {'write B':>14s} {'algorithmic':>14s} {'ratio':>7s} {'avg us':>9s}")
    for k in fetch:
        m = re.search(r"dequant_many<ggq::(Fmt\w+), \d+, (\d), \w+, \w+, \d+, (\d), \w+>", k)
        if m:
            fmt, out, comp = m.groups()
            name = f"{fmt} {DT[comp]}->{DT[out]}"
            a_read = ELEMENTS * TS[fmt]
            a_write = ELEMENTS * (4 if out == "2" else 2)
        elif k.startswith(("k_copy16", "k_fill16")):
            name = k.split("(")[0]
            a_read = (1 << 30) if "copy" in name else 0
            a_write = 1 << 30
        else:
            continue
        rb, wb = fetch[k] * 2048, write.get(k, 0.0) * 1024
        if m:
            key = f"{fmt[3:]}:pairs64" + ("" if (comp, out) == ("0", "0") else f":{DT[comp]}->{DT[out]}")
            traffic[key] = int(round(rb + wb))
        rr = f"{rb / a_read:7.4f}" if a_read else "    nan"
        print(f"{name:28s} {fetch[k]:12.1f} {write.get(k, 0.0):12.1f} {rb:14.0f} {a_read:14.0f} {rr} {wb:14.0f} {a_write:14.0f} {wb / a_write:7.4f} {dur.get(k, 0.0):9.1f}")
    if json_out:
        import json
        import os
        import importlib
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        nat = importlib.import_module("comfyui-gguf_amd._native")
        traffic["_build_id"] = nat.source_id()
        traffic["_provenance"] = ("tools/pmc_summarize.py over bash tests/microbench/pmc.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                                  "(--kernel-trace only); FETCH_SIZE x 2048 (gfx950 counts a 16 B/lane stream at half rate: calibrated on a 1 GiB copy in the same pass), "
                                  "WRITE_SIZE x 1024; bytes per launch of the 64-pair FLUX pool, shipped kernels through ggq_plan_launch")
        with open(json_out, "w") as f:
            json.dump(traffic, f, indent=1)
    sq = f"{root}/p3/p_counter_collection.csv"
    names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
             "SQ_LDS_UNALIGNED_STALL", "GRBM_GUI_ACTIVE"]
    try:
        cols = {n: counters(sq, n) for n in names}
    except OSError:
        return
    print("\n# SQ pass (per launch; SQ_* cycle counters are quad-cycles summed over waves)")
    print(f"{'kernel (compute->out)':28s} " + " ".join(f"{n:>22s}" for n in names))
    for k in cols["SQ_WAVES"]:
        m = re.search(r"dequant_many<ggq::(Fmt\w+), \d+, (\d), \w+, \w+, \d+, (\d), \w+>", k)
        if not m:
            continue
        fmt, out, comp = m.groups()
        print(f"{fmt + ' ' + DT[comp] + '->' + DT[out]:28s} " + " ".join(f"{cols[n].get(k, float('nan')):22.4g}" for n in names))


def workloads(root, json_path):
    """gpurun_out/pmcw/<workload>-<COUNTER>/ (tools/pmc_workloads.sh) -> "<workload>:Q4_K_M" entries of pmc_traffic.json: per plan launch
    = the sum over the kernels of one step (one dequant_many per format present), mean over the steps of the run."""
    import json
    with open(json_path) as f:
        table = json.load(f)
    for w in ("flux", "sd35-t5"):
        total = 0.0
        for counter, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
            per_kernel = collections.defaultdict(list)
            for r in csv.DictReader(open(f"{root}/{w}-{counter}/p_counter_collection.csv")):
                if r["Counter_Name"] == counter and "dequant_many" in r["Kernel_Name"]:
                    per_kernel[r["Kernel_Name"]].append(float(r["Counter_Value"]))
            step = sum(sum(v) / len(v) for v in per_kernel.values())          # every kernel runs once per step
            print(f"{w:8s} {counter:10s} {len(per_kernel)} kernels/step  {step * scale / 1e6:12.2f} MB per step")
            total += step * scale
        table[f"{w}:Q4_K_M"] = int(round(total))
    with open(json_path, "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    if "--workloads" in args:
        i = args.index("--workloads")
        workloads(args[i + 1], args[i + 2])
        sys.exit(0)
    json_out = None
    if "--json" in args:
        i = args.index("--json")
        json_out = args[i + 1]
        del args[i:i + 2]
    main(args[0] if args else "gpurun_out/pmc", json_out)
