#!/usr/bin/env python3
"""Summarise the counter passes of tests/microbench/pmc3.sh (gpurun_out/pmc3/p*/) into one table: rows = counters, columns = the
shipped Q4_K / Q2_K kernels and the no-arithmetic streams (fill, copy, 9:32 mix), values = mean per launch; plus the kernel
durations of every pass.

    python tools/pmc3_summarize.py gpurun_out/pmc3 > profiles/<name>.txt
"""
import collections
import csv
import glob
import os
import re
import sys


def label(kernel):
    m = re.search(r"dequant_many<ggq::Fmt(\w+),", kernel)
    if m:
        return m.group(1)
    m = re.search(r"k_stream_x<(\d)>", kernel)
    if m:
        return {"0": "fill", "1": "copy", "2": "mix9:32"}[m.group(1)]
    return None


def main(root):
    cols = ["Q4_K", "Q2_K", "fill", "copy", "mix9:32"]
    table = collections.OrderedDict()
    durs = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "p*")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p)) or 0)):
        cc = os.path.join(d, "p_counter_collection.csv")
        kt = os.path.join(d, "p_kernel_trace.csv")
        if not os.path.isfile(cc):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            lab = label(r["Kernel_Name"])
            if lab:
                acc[(r["Counter_Name"], lab)].append(float(r["Counter_Value"]))
        for (name, lab), v in acc.items():
            table.setdefault(name, {})[lab] = sum(v) / len(v)
        if os.path.isfile(kt):
            for r in csv.DictReader(open(kt)):
                lab = label(r["Kernel_Name"])
                if lab:
                    durs[lab].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# rocprofv3 --pmc <set> --kernel-trace -- tests/microbench/ggq_microbench pmc3   (tests/microbench/pmc3.sh), MI355X")
    print("# Q4_K / Q2_K = the SHIPPED kernels on the bench pool (64 x (3072x3072 + 3072x12288), ggq_plan_launch): 6.04 GB written, 1.70 / 0.99 GB read;")
    print("# fill / copy / mix9:32 = no-arithmetic streams, 6 GiB written (copy: + 6 GiB read; mix: + 1.69 GiB read), same XCD run mapping")
    print(f"{'counter (mean per launch)':44s} " + " ".join(f"{c:>14s}" for c in cols))
    print(f"{'duration_us (all passes)':44s} " + " ".join(f"{(sum(durs[c]) / len(durs[c]) if durs[c] else float('nan')):14.1f}" for c in cols))
    for name, row in table.items():
        print(f"{name:44s} " + " ".join(f"{row.get(c, float('nan')):14.4g}" for c in cols))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc3")
