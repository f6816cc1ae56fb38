"""Stage the reference's Python sources for the GPU box (TEST INFRASTRUCTURE; never used by the product).

/root/reference exists in the build container only.  The reference is Python, so there is nothing to
compile into ``oracle/_ref``; instead this recipe copies the four files the hot path and its parity
tests execute -- verbatim, byte for byte -- into the git-ignored ``oracle/_ref/`` (listed in .gitignore,
NOT in .gpurunignore), so that they travel with the gpurun snapshot exactly as a built ``.so`` does and
never enter the repository's history:

    dequant.py        the oracle of oracles: the torch block functions (timed by bench.py's cpu_baseline,
                      kind "reference"; executed on GPU tensors by tests/test_gpu_reference.py)
    ops.py            GGMLTensor / GGMLLayer / GGMLOps -- the boundary install() patches underneath
    loader.py         gguf_sd_loader & co. (tests/test_gguf.py runs it live against the loader mirror)
    tools/convert.py  detect_arch, which loader.py imports for files without general.architecture

``__graft_entry__.build()`` calls :func:`stage` whenever /root/reference is present.  ``MANIFEST.json``
records the sha256 of every staged file; :func:`verify` re-checks it on the GPU box before a test trusts
the copy.
"""
import hashlib
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE_DIR = "/root/reference"
STAGE_DIR = os.path.join(HERE, "_ref")
FILES = ("dequant.py", "ops.py", "loader.py", os.path.join("tools", "convert.py"))


def _sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def stage(source_dir=SOURCE_DIR, stage_dir=STAGE_DIR):
    """Copy FILES from the reference checkout into oracle/_ref/.  Returns the manifest, or None when the
    reference is not present (GPU box: the staged copy that travelled with the snapshot is used as is)."""
    if not os.path.isfile(os.path.join(source_dir, "dequant.py")):
        return None
    manifest = {}
    for rel in FILES:
        src, dst = os.path.join(source_dir, rel), os.path.join(stage_dir, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        manifest[rel] = _sha256(dst)
    with open(os.path.join(stage_dir, "MANIFEST.json"), "w") as f:
        json.dump({"source": source_dir, "sha256": manifest}, f, indent=1, sort_keys=True)
    with open(os.path.join(stage_dir, "README.txt"), "w") as f:
        f.write("VERBATIM copies of four files of city96/ComfyUI-GGUF, staged at build time by oracle/stage_reference.py so that the GPU box\n"
                "(which has no /root/reference) can run install() over the reference's real classes and time its own torch-CPU path.\n"
                "Test infrastructure only: git-ignored (never committed), never imported by the product, checked against MANIFEST.json before use.\n")
    return manifest


def verify(stage_dir=STAGE_DIR):
    """True when oracle/_ref holds every file of its manifest, unmodified."""
    try:
        with open(os.path.join(stage_dir, "MANIFEST.json")) as f:
            manifest = json.load(f)["sha256"]
    except (OSError, ValueError, KeyError):
        return False
    return set(manifest) == set(FILES) and all(
        os.path.isfile(os.path.join(stage_dir, rel)) and _sha256(os.path.join(stage_dir, rel)) == digest for rel, digest in manifest.items())


if __name__ == "__main__":
    m = stage()
    print("staged:" if m else "reference not present; staged copy valid:", m if m else verify())
