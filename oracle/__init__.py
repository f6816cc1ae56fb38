"""CPU oracle for the GGUF dequant path -- TEST INFRASTRUCTURE, not product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker (or the CPU number reported beside the GPU one).  The
shipped path under ``comfyui-gguf_amd/`` never imports it.

``ggq_oracle.c`` is a plain-C restatement of the reference's block unpackers (each function cites
the reference dequant.py lines it follows); this module compiles it with gcc and binds it with
ctypes.  Parity is pinned: ``make_golden.py`` ran the reference's dequant.py verbatim to produce
``tests/golden/*.npz`` and ``tests/test_oracle.py`` holds the C code bit-exact to them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libggq_oracle.so")
_lib = None


def build(force=False):
    """Compile ggq_oracle.c with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("ggq_oracle.c", "ggq_oracle_simd.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL if not force else None)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u16p, f32p, u32p = (ctypes.POINTER(t) for t in (ctypes.c_uint8, ctypes.c_uint16, ctypes.c_float, ctypes.c_uint32))
        for name, outp in (("ggq_oracle_dequant_f16", u16p), ("ggq_oracle_dequant_f32", f32p), ("ggq_oracle_dequant_bf16", u16p),
                           ("ggq_oracle_simd_dequant_f16", u16p)):
            fn = getattr(L, name)
            fn.argtypes = [ctypes.c_int, u8p, ctypes.c_uint64, outp]
            fn.restype = ctypes.c_int
        L.ggq_oracle_bf16_to_f32.argtypes = [u8p, ctypes.c_uint64, u32p]
        L.ggq_oracle_cast_f16_to_bf16.argtypes = [u16p, ctypes.c_uint64, u16p]
        L.ggq_oracle_cast_f16_to_f32.argtypes = [u16p, ctypes.c_uint64, f32p]
        L.ggq_oracle_cast_f32_to_bf16.argtypes = [f32p, ctypes.c_uint64, u16p]
        L.ggq_oracle_cast_f32_to_f16.argtypes = [f32p, ctypes.c_uint64, u16p]
        L.ggq_oracle_block_size.argtypes = [ctypes.c_int]
        L.ggq_oracle_type_size.argtypes = [ctypes.c_int]
        L.ggq_oracle_d2h.argtypes = [ctypes.c_double]
        L.ggq_oracle_d2h.restype = ctypes.c_uint16
        L.ggq_oracle_h2d.argtypes = [ctypes.c_uint16]
        L.ggq_oracle_h2d.restype = ctypes.c_double
        for name in ("ggq_oracle_hmul", "ggq_oracle_hadd", "ggq_oracle_hsub"):
            fn = getattr(L, name)
            fn.argtypes = [ctypes.c_uint16, ctypes.c_uint16]
            fn.restype = ctypes.c_uint16
        L.ggq_oracle_set_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _ptr(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def geometry(qtype):
    L = lib()
    return L.ggq_oracle_block_size(int(qtype)), L.ggq_oracle_type_size(int(qtype))


def _prep(qtype, packed):
    packed = np.ascontiguousarray(np.asarray(packed).reshape(-1).view(np.uint8))
    bs, ts = geometry(qtype)
    if bs == 0:
        raise ValueError(f"oracle: unsupported qtype {int(qtype)}")
    n_blocks = packed.size // ts            # dequant.py:41
    return packed, n_blocks, bs


def simd_available():
    """True if the host can run the AVX2+F16C throughput leg (ggq_oracle_simd.c)."""
    return bool(lib().ggq_oracle_simd_available())


def dequant_f16(qtype, packed, threads=None, simd=False, out=None):
    """Default path (dequant_dtype=None): returns the fp16 result as a 1-D np.float16 array.
    simd=True runs the AVX2+F16C leg (same values; bench.py's cpu_baseline) instead of the soft-float checker;
    ``out`` (np.uint16, right size) avoids re-allocating the result when timing."""
    L = lib()
    if threads:
        L.ggq_oracle_set_threads(int(threads))
    packed, n_blocks, bs = _prep(qtype, packed)
    if out is None:
        out = np.empty(n_blocks * bs, dtype=np.uint16)
    fn = L.ggq_oracle_simd_dequant_f16 if simd else L.ggq_oracle_dequant_f16
    rc = fn(int(qtype), _ptr(packed, ctypes.c_uint8), n_blocks, _ptr(out, ctypes.c_uint16))
    if rc == -2:
        raise RuntimeError("oracle: this host has no AVX2+F16C")
    if rc:
        raise ValueError(f"oracle: unsupported qtype {int(qtype)}")
    return out.view(np.float16)


def dequant_f32(qtype, packed):
    """dequant_dtype=float32 mode: fp32 arithmetic, np.float32 result."""
    L = lib()
    packed, n_blocks, bs = _prep(qtype, packed)
    out = np.empty(n_blocks * bs, dtype=np.float32)
    rc = L.ggq_oracle_dequant_f32(int(qtype), _ptr(packed, ctypes.c_uint8), n_blocks, _ptr(out, ctypes.c_float))
    if rc:
        raise ValueError(f"oracle: unsupported qtype {int(qtype)}")
    return out


def dequant_bf16_bits(qtype, packed):
    """dequant_dtype=bfloat16 mode: bf16 arithmetic; returns the bf16 bit patterns (np.uint16)."""
    L = lib()
    packed, n_blocks, bs = _prep(qtype, packed)
    out = np.empty(n_blocks * bs, dtype=np.uint16)
    rc = L.ggq_oracle_dequant_bf16(int(qtype), _ptr(packed, ctypes.c_uint8), n_blocks, _ptr(out, ctypes.c_uint16))
    if rc:
        raise ValueError(f"oracle: unsupported qtype {int(qtype)}")
    return out


def bf16_to_f32(packed):
    """dequant.py:61-62: bf16 bits -> fp32."""
    L = lib()
    packed = np.ascontiguousarray(np.asarray(packed).reshape(-1).view(np.uint8))
    n = packed.size // 2
    out = np.empty(n, dtype=np.uint32)
    L.ggq_oracle_bf16_to_f32(_ptr(packed, ctypes.c_uint8), n, _ptr(out, ctypes.c_uint32))
    return out.view(np.float32)


def cast_f16_to_bf16_bits(h):
    L = lib()
    h = np.ascontiguousarray(h).view(np.uint16).reshape(-1)
    out = np.empty_like(h)
    L.ggq_oracle_cast_f16_to_bf16(_ptr(h, ctypes.c_uint16), h.size, _ptr(out, ctypes.c_uint16))
    return out


def cast_f32_to_bf16_bits(x):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty(x.size, dtype=np.uint16)
    L.ggq_oracle_cast_f32_to_bf16(_ptr(x, ctypes.c_float), x.size, _ptr(out, ctypes.c_uint16))
    return out


def cast_f32_to_f16_bits(x):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.empty(x.size, dtype=np.uint16)
    L.ggq_oracle_cast_f32_to_f16(_ptr(x, ctypes.c_float), x.size, _ptr(out, ctypes.c_uint16))
    return out


def dequant_tensor(qtype, packed, compute="f16", out="f16"):
    """dequantize_tensor(tensor, dtype=out, dequant_dtype=compute) (dequant.py:15-23):
    dequantize(..., dtype=compute).to(out).  compute / out in {"f16", "bf16", "f32"}.
    Returns np.float32 for out == "f32", else the 16-bit patterns (np.uint16)."""
    if compute == "f16":
        h = dequant_f16(qtype, packed)
        if out == "f16":
            return h.view(np.uint16)
        if out == "bf16":
            return cast_f16_to_bf16_bits(h)
        return h.astype(np.float32)                      # exact
    if compute == "f32":
        x = dequant_f32(qtype, packed)
    elif compute == "bf16":
        b = dequant_bf16_bits(qtype, packed)
        if out == "bf16":
            return b
        x = (b.astype(np.uint32) << 16).view(np.float32)  # exact
    else:
        raise ValueError(compute)
    if out == "f32":
        return x
    return cast_f32_to_bf16_bits(x) if out == "bf16" else cast_f32_to_f16_bits(x)


def canon_nan(bits_or_f32):
    """NaN payloads (and NaN signs) are not part of parity: map every NaN to one pattern.
    uint16 input is ambiguous between fp16 and bf16, so use canon_nan_f16 / canon_nan_bf16 for those."""
    a = np.ascontiguousarray(bits_or_f32)
    assert a.dtype == np.float32
    bits = a.view(np.uint32).copy()
    bits[(bits & 0x7FFFFFFF) > 0x7F800000] = 0x7FC00000
    return bits


def canon_nan_bf16(a):
    bits = np.ascontiguousarray(a).view(np.uint16).copy()
    bits[(bits & 0x7FFF) > 0x7F80] = 0x7FC0
    return bits


def canon_nan_f16(a):
    """fp16 bit patterns with every NaN mapped to one value (NaN payloads are not part of parity)."""
    bits = np.ascontiguousarray(a).view(np.uint16).copy()
    bits[(bits & 0x7FFF) > 0x7C00] = 0x7E00
    return bits
