#!/usr/bin/env python3
"""Build a second copy of libggq_hip.so with extra -D flags, for same-box A/B runs ("A/B knobs"):

    python tools/build_variant.py -DGGQ_SOLO_ONLY -o gpurun_tmp_libs/libggq_solo.so
    gpurun -- 'for i in 1 2; do python bench.py ...; GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_solo.so python bench.py ...; done'

Defines the sources understand (csrc/ggq_capi.hip): GGQ_SOLO_ONLY (one-wave teams everywhere), GGQ_COOP_ALL_MODES (workgroup
teams everywhere), GGQ_SOLO_CAST_OUT (one-wave teams whenever the output is not fp16), GGQ_CAST_SOLO_AT_LAYER_SIZE (... also for
single layers).  The variant goes through
the same FMA guard as the shipped build.  Keep variants out of comfyui-gguf_amd/_lib/ and out of git (*.so is ignored)."""
import argparse
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-o", "--out", required=True)
    args, defines = ap.parse_known_args()
    bad = [d for d in defines if not d.startswith("-D")]
    if bad:
        ap.error(f"only -D flags are passed on: {bad}")
    nat = importlib.import_module("comfyui-gguf_amd._native")
    out = os.path.abspath(args.out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="ggq_variant_") as tmp:
        lib = os.path.join(tmp, "lib.so")
        cmd = [nat.hipcc_path()] + nat.HIPCC_FLAGS + defines + ["-save-temps=obj", "-o", lib] + nat.SOURCES
        proc = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
        if proc.returncode:
            sys.exit(proc.stderr[-4000:])
        for f in os.listdir(tmp):
            if f.endswith(".s") and "amdgcn" in f:
                with open(os.path.join(tmp, f)) as fh:
                    nat.check_no_fma(fh.read())
        shutil.copyfile(lib, out)
    print(out)


if __name__ == "__main__":
    main()
