#!/usr/bin/env python3
"""Build a second copy of libggq_hip.so with extra -D flags, for same-box A/B runs ("A/B knobs"):

    python tools/build_variant.py -DGGQ_SOLO_ONLY -o gpurun_tmp_libs/libggq_solo.so
    gpurun -- 'for i in 1 2; do python bench.py ...; GGQ_HIP_LIB=$PWD/gpurun_tmp_libs/libggq_solo.so python bench.py ...; done'

Defines the sources understand (csrc/ggq_capi.hip): GGQ_SOLO_ONLY (one-wave teams everywhere), GGQ_COOP_ALL_MODES (workgroup
teams everywhere), GGQ_SOLO_CAST_OUT (one-wave teams whenever the output is not fp16), GGQ_CAST_SOLO_AT_LAYER_SIZE (... also for
single layers); csrc/ggq_device.hpp: GGQ_F32_PAIR_MODE=0|1|2 (fp32 output: decode every chunk twice / swap halves in workgroup teams only /
in every team); csrc/ggq_capi.hip: GGQ_F32_TEAMS_R4, GGQ_F32_HALF_GROUP, GGQ_Q2K_PLAIN_LOADS.  The variant goes through
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


def _objects_without_defines(nat, cache, skip=0):
    """The translation units the -D flags of an A/B build do not reach (all but nat.SOURCES[skip]), compiled once per
    source state into ``cache`` and shared by every variant."""
    import hashlib
    os.makedirs(cache, exist_ok=True)
    h = hashlib.sha256(" ".join(nat.HIPCC_FLAGS).encode())
    others = [p for i, p in enumerate(nat.SOURCES) if i != skip]
    for path in others + nat.HEADERS:
        with open(path, "rb") as f:
            h.update(f.read())
    tag = h.hexdigest()[:12]
    objs, jobs = [], []
    flags = [f for f in nat.HIPCC_FLAGS if f != "-shared"]
    for src in others:
        obj = os.path.join(cache, f"{os.path.splitext(os.path.basename(src))[0]}.{tag}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((obj, subprocess.Popen([nat.hipcc_path()] + flags + ["-c", "-o", obj + ".tmp", src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for obj, proc in jobs:
        _, err = proc.communicate()
        if proc.returncode:
            sys.exit(err[-4000:])
        os.replace(obj + ".tmp", obj)
    return objs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--all-sources", action="store_true", help="pass the -D flags to every translation unit (default: to csrc/ggq_capi.hip, which holds "
                                                               "every dequant kernel and its launch geometry; the other three are compiled once and shared)")
    ap.add_argument("--unit", default="ggq_capi.hip", help="the translation unit the -D flags go to (ggq_capi.hip: the dequant kernels; ggq_linear.hip: the fused linears)")
    args, defines = ap.parse_known_args()
    bad = [d for d in defines if not d.startswith("-D")]
    if bad:
        ap.error(f"only -D flags are passed on: {bad}")
    nat = importlib.import_module("comfyui-gguf_amd._native")
    out = os.path.abspath(args.out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="ggq_variant_") as tmp:
        lib = os.path.join(tmp, "lib.so")
        stamp = [f'-DGGQ_BUILD_ID="{nat.source_id()}+{"".join(defines)}"']
        if args.all_sources:
            cmd = [nat.hipcc_path()] + nat.HIPCC_FLAGS + defines + stamp + ["-save-temps=obj", "-o", lib] + nat.SOURCES
        else:
            unit = [i for i, p in enumerate(nat.SOURCES) if os.path.basename(p) == args.unit]
            if not unit:
                ap.error(f"--unit: one of {[os.path.basename(p) for p in nat.SOURCES]}")
            shared = _objects_without_defines(nat, os.path.join(os.path.dirname(out), ".obj"), unit[0])
            obj = os.path.join(tmp, "unit.o")
            compile_flags = [f for f in nat.HIPCC_FLAGS if f != "-shared"]
            proc = subprocess.run([nat.hipcc_path()] + compile_flags + defines + stamp + ["-save-temps=obj", "-c", "-o", obj, nat.SOURCES[unit[0]]], cwd=tmp, capture_output=True, text=True)
            if proc.returncode:
                sys.exit(proc.stderr[-4000:])
            cmd = [nat.hipcc_path()] + nat.HIPCC_FLAGS + ["-o", lib, obj] + shared
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
