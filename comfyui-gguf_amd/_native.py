"""ctypes binding of libggq_hip.so (C ABI: include/ggq.h) and its in-tree build.

The shared library is built IN-TREE (``comfyui-gguf_amd/_lib/libggq_hip.so``) with hipcc for
gfx950 so that it travels with the repository snapshot to the GPU box.  There is no fallback of
any kind: if the library is missing or fails to load, :func:`lib` raises -- the HIP path is the
product, a silent detour through torch or the CPU would void every parity and performance claim.
"""
import ctypes
import os
import re
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
# GGQ_HIP_LIB: load another build of the library (A/B measurements); default = the in-tree build
LIB_PATH = os.environ.get("GGQ_HIP_LIB") or os.path.join(LIB_DIR, "libggq_hip.so")
SOURCES = [os.path.join(CSRC, "ggq_capi.hip"), os.path.join(CSRC, "ggq_gguf.hip"), os.path.join(CSRC, "ggq_linear.hip"), os.path.join(CSRC, "ggq_overlap.hip")]
HEADERS = [os.path.join(CSRC, "ggq_device.hpp"), os.path.join(CSRC, "ggq_linear.hpp"), os.path.join(CSRC, "ggq_mfma.hpp"), os.path.join(CSRC, "ggq_mfma16.hpp"), os.path.join(CSRC, "ggq_gemm.hpp"), os.path.join(CSRC, "ggq_host.hpp"), os.path.join(ROOT, "include", "ggq.h"), os.path.join(ROOT, "include", "ggq_gguf.h")]
ABI_VERSION = 11

# -ffp-contract=off is REQUIRED for parity: hipcc otherwise fuses the reference's separately
# rounded fp16 multiply and subtract into v_pk_fma_f16 (SURVEY.md section 0 finding 3).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

# fused multiply-add mnemonics that must not appear in the dequant kernels
_FMA_RE = re.compile(r"\b(v_(?:pk_)?(?:fma|fmac)_\w+|v_mad_(?:f16|f32|legacy_f\w+|mix\w*|mixlo\w*|mixhi\w*)\w*)")

GGQ_OK, GGQ_ERR_QTYPE, GGQ_ERR_ALIGN, GGQ_ERR_ARG, GGQ_ERR_HIP, GGQ_ERR_NOMEM, GGQ_ERR_IO, GGQ_ERR_FORMAT = range(8)
F16, BF16, F32 = 0, 1, 2          # ggq_dtype: compute and out dtypes
CAL_FILL, CAL_FILL_NT, CAL_COPY, CAL_COPY_NT, CAL_READ = range(5)      # ggq_cal_kind
OUT_F16, OUT_BF16, OUT_F32 = F16, BF16, F32

# every symbol include/ggq.h declares: name -> (restype, argtypes)
_u64, _u32, _int, _vp = ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p


class ggq_desc(ctypes.Structure):
    _fields_ = [("qtype", ctypes.c_int32), ("out_dtype", ctypes.c_int32), ("packed", _vp), ("out", _vp), ("n_blocks", _u64),
                ("compute_dtype", ctypes.c_int32), ("reserved", ctypes.c_int32)]


GGUF_MAX_DIMS = 8


class ggq_gguf_info(ctypes.Structure):
    _fields_ = [("version", _u32), ("alignment", _u32), ("n_tensors", _u64), ("n_kv", _u64), ("data_offset", _u64),
                ("data_bytes", _u64), ("file_bytes", _u64), ("base", _vp)]


class ggq_gguf_tensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("qtype", ctypes.c_int32), ("n_dims", _u32), ("dims", _u64 * GGUF_MAX_DIMS),
                ("offset", _u64), ("nbytes", _u64), ("n_elements", _u64)]


class ggq_gguf_kv(ctypes.Structure):
    _fields_ = [("key", ctypes.c_char_p), ("type", _u32), ("elem_type", _u32), ("count", _u64), ("data", _vp), ("nbytes", _u64)]


SYMBOLS = {
    "ggq_supported": (_int, [_int]),
    "ggq_block_size": (_int, [_int]),
    "ggq_type_size": (_int, [_int]),
    "ggq_strerror": (ctypes.c_char_p, [_int]),
    "ggq_last_hip_error": (_int, []),
    "ggq_abi_version": (_int, []),
    "ggq_build_id": (ctypes.c_char_p, []),
    "ggq_dequant": (_int, [_int, _vp, _u64, _vp, _int, _int, _vp]),
    "ggq_dequant_stream": (_int, [_int, _vp, _u64, _vp, _int, _int, _vp]),
    "ggq_dequant_f16": (_int, [_int, _vp, _u64, _vp, _vp]),
    "ggq_calibrate": (_int, [_int, _vp, _vp, _u64, _vp]),
    "ggq_plan_create": (_int, [ctypes.POINTER(ggq_desc), _u32, ctypes.POINTER(_vp)]),
    "ggq_plan_launch": (_int, [_vp, _vp]),
    "ggq_plan_bytes": (_u64, [_vp]),
    "ggq_plan_kernels": (_u32, [_vp]),
    "ggq_plan_destroy": (None, [_vp]),
    "ggq_linear_small": (_int, [_int, _vp, _u32, _u32, _vp, _u32, _vp, _vp, _int, _vp]),
    "ggq_overlap_create": (_int, [_int, ctypes.POINTER(_vp)]),
    "ggq_overlap_copy": (_int, [_vp, _int, _vp, _vp, _u64]),
    "ggq_overlap_prefetch": (_int, [_vp, _int, _int, _int, _vp, _u64, _vp, _int, _int, _vp]),
    "ggq_overlap_wait": (_int, [_vp, _int, _vp]),
    "ggq_overlap_destroy": (None, [_vp]),
    "ggq_linear_mfma": (_int, [_int, _vp, _u32, _u32, _vp, _u32, _vp, _vp, _int, _int, _vp]),
    "ggq_linear_mfma_ws": (_int, [_int, _vp, _u32, _u32, _vp, _u32, _vp, _vp, _int, _int, _vp, _u64, _vp]),
    "ggq_linear_mfma_workspace": (_u64, [_int, _u32, _u32, _u32, _int]),
    "ggq_dequant_rows": (_int, [_int, _vp, _u64, _u32, _vp, _u64, _vp, _int, _int, _vp]),
    # include/ggq_gguf.h
    "ggq_gguf_open": (_int, [ctypes.c_char_p, ctypes.POINTER(_vp)]),
    "ggq_gguf_close": (None, [_vp]),
    "ggq_gguf_get_info": (_int, [_vp, ctypes.POINTER(ggq_gguf_info)]),
    "ggq_gguf_get_tensor": (_int, [_vp, _u64, ctypes.POINTER(ggq_gguf_tensor)]),
    "ggq_gguf_find_kv": (ctypes.c_int64, [_vp, ctypes.c_char_p]),
    "ggq_gguf_get_kv": (_int, [_vp, _u64, ctypes.POINTER(ggq_gguf_kv)]),
    "ggq_gguf_kv_string": (_int, [_vp, _u64, _u64, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(_u64)]),
    "ggq_ggml_type_geometry": (_int, [_int, ctypes.POINTER(_u32), ctypes.POINTER(_u32)]),
    "ggq_gguf_upload": (_int, [_vp, _vp, _u64, _u64, _int, _u64, _vp]),
    "ggq_gguf_upload_release": (None, []),
}

_lib = None


class GGQNativeError(RuntimeError):
    """libggq_hip.so is missing / unloadable / returned a failing status."""


DEQUANT_KERNEL_FILES = ("ggq_device.hpp", "ggq_capi.hip", "ggq_host.hpp")     # what the dequant kernels and their launch geometry are made of


def source_id():
    """First 16 hex digits of the sha256 over the compiler flags and the sources of the DEQUANT kernels (device code + launch
    geometry; not the GGUF reader or the fused-linear kernels, which do not touch what the PMC figures measure): what
    ``ggq_build_id()`` of an in-tree build returns (stamped with -DGGQ_BUILD_ID at compile time)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for p in [os.path.join(CSRC, f) for f in DEQUANT_KERNEL_FILES]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def check_no_fma(asm_text):
    """Raise if a fused multiply-add made it into the device code (build-time parity guard)."""
    hits = sorted(set(m.group(1) for m in _FMA_RE.finditer(asm_text)))
    if hits:
        raise GGQNativeError(f"fused multiply-add instructions in the dequant kernels: {hits}; "
                             "the reference rounds after every op -- build with -ffp-contract=off")


_SPILL_RE = re.compile(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)")
MAX_SCRATCH_BYTES = 16


def check_no_spills(asm_text, limit=MAX_SCRATCH_BYTES):
    """Raise if a kernel keeps more than ``limit`` bytes per lane in scratch memory (build-time guard, round 6): a register spill is a vector-memory access in front of the
    in-order weight stream -- the 16-row fused kernel ran Q8_0 at half speed for most of round 6 with 72-108 bytes of it and nothing said so (EXPERIMENTS.md R6-9)."""
    bad = sorted((int(n), name) for name, n in _SPILL_RE.findall(asm_text) if int(n) > limit)
    if bad:
        raise GGQNativeError(f"register spills in {len(bad)} kernels (bytes of scratch per lane, mangled name): {bad[:6]}; lower the instantiation's occupancy request or "
                             "its widest workgroup (csrc/ggq_mfma16.hpp Mf16Occ, csrc/ggq_mfma.hpp mf_max_waves)")


def build(force=False, verbose=False):
    """Compile the HIP extension for gfx950 into comfyui-gguf_amd/_lib/ (cross-compiles without a GPU)."""
    global _lib
    build_fast(force)
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="ggq_build_") as tmp:
        out = os.path.join(tmp, "libggq_hip.so")
        # one hipcc -c per translation unit, all at once (the four take 20-40 s each), then one link
        compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]
        jobs = []
        for src in SOURCES:
            stem = os.path.splitext(os.path.basename(src))[0]
            d = os.path.join(tmp, stem)
            os.makedirs(d)
            cmd = [hipcc_path()] + compile_flags + [f'-DGGQ_BUILD_ID="{source_id()}"', "-save-temps=obj", "-c", "-o", os.path.join(d, stem + ".o"), src]
            jobs.append((cmd, d, subprocess.Popen(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        failed = None
        for cmd, d, proc in jobs:
            so, se = proc.communicate()
            if verbose or proc.returncode:
                print(" ".join(cmd))
                print(so + se)
            if proc.returncode and failed is None:
                failed = (proc.returncode, se)
        if failed:
            raise GGQNativeError(f"hipcc failed ({failed[0]}):\n{failed[1][-4000:]}")
        asm = [os.path.join(d, f) for _, d, _ in jobs for f in os.listdir(d) if f.endswith(".s") and "amdgcn" in f]
        if len(asm) < len(SOURCES):
            raise GGQNativeError("build produced no gfx950 assembly to check for FMA contraction")
        for f in asm:
            with open(f) as fh:
                text = fh.read()
            check_no_fma(text)
            check_no_spills(text)
        link = [hipcc_path()] + HIPCC_FLAGS + ["-o", out] + [os.path.join(d, os.path.basename(d) + ".o") for _, d, _ in jobs]
        proc = subprocess.run(link, cwd=tmp, capture_output=True, text=True)
        if verbose or proc.returncode:
            print(" ".join(link))
            print(proc.stdout + proc.stderr)
        if proc.returncode:
            raise GGQNativeError(f"hipcc link failed ({proc.returncode}):\n{proc.stderr[-4000:]}")
        shutil.copyfile(out, LIB_PATH + ".tmp")
        os.replace(LIB_PATH + ".tmp", LIB_PATH)
    _lib = None
    return LIB_PATH


FAST_SRC = os.path.join(CSRC, "ggq_pyfast.c")


def fast_path():
    import sysconfig
    return os.path.join(LIB_DIR, "_ggq_fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_fast(force=False):
    """gcc the optional CPython binding of ggq_dequant (csrc/ggq_pyfast.c).  Returns its path, or None when Python.h / gcc are
    not there -- dequant.py then keeps calling through ctypes."""
    import sysconfig
    out = fast_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(FAST_SRC):
        return out
    inc = sysconfig.get_paths().get("include")
    gcc = shutil.which("gcc")
    if not gcc or not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    os.makedirs(LIB_DIR, exist_ok=True)
    # a unique temporary name: several processes (one per GPU) may build at once; os.replace makes the last one win atomically
    fd, tmp = tempfile.mkstemp(prefix="_ggq_fast.", suffix=".tmp", dir=LIB_DIR)
    os.close(fd)
    try:
        proc = subprocess.run([gcc, "-O2", "-shared", "-fPIC", "-I" + inc, FAST_SRC, "-o", tmp], capture_output=True, text=True)
        if proc.returncode:
            # the binding is OPTIONAL (ctypes serves the same entry points): a compiler problem here must not fail the build of the product
            import warnings
            warnings.warn(f"comfyui-gguf_amd: gcc failed on ggq_pyfast.c, keeping the ctypes binding:\n{proc.stderr[-1000:]}")
            return None
        os.replace(tmp, out)
        tmp = None
    finally:
        if tmp is not None and os.path.exists(tmp):
            os.remove(tmp)
    return out


def fast():
    """The _ggq_fast module bound to the loaded library's ggq_dequant, or None if it has not been built."""
    path = fast_path()
    if not os.path.exists(path):
        return None
    import importlib.util
    if os.path.getmtime(path) < os.path.getmtime(FAST_SRC):
        return None                                            # stale against its source: ctypes until build() has refreshed it
    try:
        spec = importlib.util.spec_from_file_location("_ggq_fast", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    except (ImportError, OSError):
        return None
    L = lib()
    if getattr(mod, "ABI", None) != L.ggq_abi_version():
        return None                                            # built against another ggq.h: raw function pointers with other signatures
    mod.bind(ctypes.cast(L.ggq_dequant, ctypes.c_void_p).value)
    mod.bind_linear(ctypes.cast(L.ggq_linear_small, ctypes.c_void_p).value, ctypes.cast(L.ggq_linear_mfma, ctypes.c_void_p).value,
                    ctypes.cast(L.ggq_linear_mfma_ws, ctypes.c_void_p).value)
    return mod


def lib():
    """The loaded library with argtypes set.  Raises GGQNativeError if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GGQNativeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` at the repo root). "
                "There is no CPU or torch fallback for this path.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise GGQNativeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise GGQNativeError(f"{LIB_PATH} does not export {name} (stale build?)") from e
            fn.restype, fn.argtypes = res, args
        if L.ggq_abi_version() != ABI_VERSION:
            raise GGQNativeError(f"{LIB_PATH} has ABI {L.ggq_abi_version()}, expected {ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def check(status, what="ggq call"):
    if status != GGQ_OK:
        L = lib()
        msg = L.ggq_strerror(status).decode()
        if status == GGQ_ERR_HIP:
            msg += f" (hipError_t {L.ggq_last_hip_error()})"
        raise GGQNativeError(f"{what} failed: {msg}")
