#!/usr/bin/env python3
"""Sweep the file -> HBM upload (ggq_gguf_upload) over reader-thread counts and chunk sizes on a synthetic
GGUF file of `--gb` gigabytes; prints GB/s per setting (best of 3 after one warm pass)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ggq_pkg import load_package  # noqa: E402
from gguf_writer import GGUFWriter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=4.0)
    args = ap.parse_args()
    pkg = load_package()
    w = GGUFWriter(arch="flux")
    per = 3072 * 12288 // 256 * 144
    n = max(1, int(args.gb * 1e9 / per))
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, per, dtype=np.uint8)
    for i in range(n):
        w.add_tensor(f"t{i}.weight", 12, (12288, 3072), blob)
    tmpdir = tempfile.mkdtemp(prefix="ggq_sweep_")
    path = w.write(os.path.join(tmpdir, "sweep.gguf"))
    size = os.path.getsize(path)
    out = {"file_GB": round(size / 1e9, 2)}
    try:
        with pkg.gguf_file.GGUFFile(path) as f:
            for threads in (2, 4, 8):
                for chunk_mib in (4, 16):
                    best = None
                    for rep in range(4):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        arena = f.upload("cuda:0", threads=threads, chunk_bytes=chunk_mib << 20)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t0
                        del arena
                        if rep and (best is None or dt < best):
                            best = dt
                    out[f"threads={threads},chunk={chunk_mib}MiB"] = round(size / best / 1e9, 1)
            # the reference's way: a read-only np.memmap view of the file (loader.py:104-106), then `.to(device)` (ops.py:209)
            import warnings
            best = None
            for rep in range(3):
                mm = np.memmap(path, dtype=np.uint8, mode="r")
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    view = torch.from_numpy(mm[f.data_offset:])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d = view.to("cuda:0")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                del d, view, mm
                best = dt if best is None else min(best, dt)
            out["torch .to(device) from a fresh np.memmap of the file (the reference's path)"] = round(size / best / 1e9, 1)
            # reference point: torch's own pageable -> device copy of the same bytes (what `.to(device)` does)
            data = torch.from_numpy(np.fromfile(path, dtype=np.uint8))
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d = data.to("cuda:0")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                del d
                best = dt if best is None else min(best, dt)
            out["torch .to(device) from pageable memory"] = round(size / best / 1e9, 1)
            pinned = data.pin_memory()
            best = None
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d = pinned.to("cuda:0", non_blocking=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                del d
                best = dt if best is None else min(best, dt)
            out["torch .to(device) from pinned memory (link ceiling)"] = round(size / best / 1e9, 1)
    finally:
        os.remove(path)
        os.rmdir(tmpdir)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
