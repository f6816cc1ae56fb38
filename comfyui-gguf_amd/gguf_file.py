"""GGUF container reader + file -> HBM streaming, over the native parser of include/ggq_gguf.h.

Stands in for the slice of the third-party ``gguf`` package the reference's loader touches
(``gguf.GGUFReader``: ``.tensors`` / ``.get_field``, loader.py:16-49,55-106) so that the dequant path
no longer needs ``gguf`` to be importable:

    f = GGUFFile(path)
    f.tensors[i].name / .tensor_type / .shape (ggml order) / .data (read-only mmap view, torch uint8)
    f.get_field(key) -> GGUFField(types, value) or None
    arena = f.upload("cuda:0")          # the whole tensor-data section in ONE device buffer, file layout kept
    f.device_bytes(arena, i)            # tensor i as a uint8 view into that arena (16-B aligned)

Parsing and the upload pipeline (pread into pinned buffers on several threads, overlapped async
H2D copies) are C++ behind the C ABI; this module is the ctypes binding plus the torch views.
"""
import ctypes
import os
import warnings
from collections import namedtuple

import numpy as np
import torch

from . import _native
from .qtypes import GGMLQuantizationType

# the container's metadata value types (== gguf.GGUFValueType)
UINT8, INT8, UINT16, INT16, UINT32, INT32, FLOAT32, BOOL, STRING, ARRAY, UINT64, INT64, FLOAT64 = range(13)
_NP = {UINT8: "<u1", INT8: "<i1", UINT16: "<u2", INT16: "<i2", UINT32: "<u4", INT32: "<i4", FLOAT32: "<f4",
       BOOL: "?", UINT64: "<u8", INT64: "<i8", FLOAT64: "<f8"}

GGUFField = namedtuple("GGUFField", "name types value")
GGUFField.__doc__ = """types: [type] or [ARRAY, elem_type] (as gguf.ReaderField.types); value: python scalar / str /
tuple (arrays) -- what the reference's get_field / get_list_field extract from .parts/.data (loader.py:28-49)."""


def _qtype(code):
    try:
        return GGMLQuantizationType(code)
    except ValueError:
        return int(code)            # a ggml type newer than the table: kept as its raw id


class GGUFTensorInfo:
    """One entry of ``GGUFFile.tensors`` -- the attributes loader.py reads off a gguf.ReaderTensor."""
    __slots__ = ("name", "tensor_type", "shape", "offset", "nbytes", "n_elements", "_file")

    def __init__(self, file, name, tensor_type, shape, offset, nbytes, n_elements):
        self._file, self.name, self.tensor_type, self.shape = file, name, tensor_type, shape
        self.offset, self.nbytes, self.n_elements = offset, nbytes, n_elements

    @property
    def data(self):
        """Read-only mmap view of the packed bytes as a torch uint8 tensor (loader.py:104-106)."""
        return self._file.cpu_bytes(self)

    def __repr__(self):
        t = getattr(self.tensor_type, "name", self.tensor_type)
        return f"GGUFTensorInfo({self.name!r}, {t}, ggml_shape={self.shape}, {self.nbytes} B @ {self.offset})"


class GGUFFile:
    def __init__(self, path):
        self.path = os.fspath(path)
        self._h = ctypes.c_void_p()
        self._mm = None
        L = _native.lib()
        rc = L.ggq_gguf_open(self.path.encode(), ctypes.byref(self._h))
        if rc != _native.GGQ_OK:
            self._h = None
            if rc == _native.GGQ_ERR_IO:
                raise OSError(ctypes.get_errno() or 5, f"cannot open/map GGUF file: {self.path}")
            raise ValueError(f"{self.path}: {L.ggq_strerror(rc).decode()}")
        info = _native.ggq_gguf_info()
        _native.check(L.ggq_gguf_get_info(self._h, ctypes.byref(info)), "ggq_gguf_get_info")
        self.version, self.alignment = int(info.version), int(info.alignment)
        self.data_offset, self.data_bytes, self.file_bytes = int(info.data_offset), int(info.data_bytes), int(info.file_bytes)
        self.n_kv = int(info.n_kv)
        self.tensors = []
        t = _native.ggq_gguf_tensor()
        for i in range(int(info.n_tensors)):
            _native.check(L.ggq_gguf_get_tensor(self._h, i, ctypes.byref(t)), "ggq_gguf_get_tensor")
            self.tensors.append(GGUFTensorInfo(self, t.name.decode("utf-8"), _qtype(t.qtype), tuple(int(d) for d in t.dims[:t.n_dims]),
                                               int(t.offset), int(t.nbytes), int(t.n_elements)))

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            _native.lib().ggq_gguf_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if not self._h:
            raise ValueError("GGUF file is closed")
        return self._h

    # ---- metadata
    def keys(self):
        L, kv = _native.lib(), _native.ggq_gguf_kv()
        out = []
        for i in range(self.n_kv):
            _native.check(L.ggq_gguf_get_kv(self._handle(), i, ctypes.byref(kv)), "ggq_gguf_get_kv")
            out.append(kv.key.decode("utf-8"))
        return out

    def _string(self, i, elem):
        p, n = ctypes.c_char_p(), ctypes.c_uint64()
        _native.check(_native.lib().ggq_gguf_kv_string(self._handle(), i, elem, ctypes.byref(p), ctypes.byref(n)), "ggq_gguf_kv_string")
        return ctypes.string_at(p, n.value) if n.value else b""

    def get_field(self, key):
        """``reader.get_field(key)``: None if absent, else GGUFField."""
        L = _native.lib()
        i = L.ggq_gguf_find_kv(self._handle(), key.encode("utf-8"))
        if i < 0:
            return None
        kv = _native.ggq_gguf_kv()
        _native.check(L.ggq_gguf_get_kv(self._handle(), i, ctypes.byref(kv)), "ggq_gguf_get_kv")
        if kv.type == STRING:
            return GGUFField(key, [STRING], self._string(i, 0).decode("utf-8"))
        if kv.type == ARRAY:
            if kv.elem_type == STRING:
                return GGUFField(key, [ARRAY, STRING], tuple(self._string(i, e).decode("utf-8") for e in range(kv.count)))
            arr = np.frombuffer(ctypes.string_at(kv.data, kv.nbytes), dtype=_NP[kv.elem_type]) if kv.count else np.empty(0, _NP[kv.elem_type])
            return GGUFField(key, [ARRAY, int(kv.elem_type)], tuple(v.item() for v in arr))
        val = np.frombuffer(ctypes.string_at(kv.data, kv.nbytes), dtype=_NP[kv.type])[0].item()
        return GGUFField(key, [int(kv.type)], val)

    # ---- tensor bytes
    def cpu_bytes(self, t):
        """uint8 view of tensor ``t``'s packed bytes in a read-only np.memmap of the file (the tensor
        keeps the mapping alive, as the reference's ``torch.from_numpy(tensor.data)`` does)."""
        if self._mm is None:
            self._mm = np.memmap(self.path, dtype=np.uint8, mode="r")
        a = self._mm[self.data_offset + t.offset: self.data_offset + t.offset + t.nbytes]
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
            return torch.from_numpy(a)

    def upload(self, device, threads=0, chunk_bytes=0):
        """Stream the whole tensor-data section into ONE device buffer (file layout preserved, so
        tensor ``t`` lives at ``arena[t.offset : t.offset + t.nbytes]``) and return it."""
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("upload() targets an AMD GPU (torch device type 'cuda')")
        arena = torch.empty(self.data_bytes, dtype=torch.uint8, device=device)
        if self.data_bytes:
            with torch.cuda.device(device):
                stream = torch.cuda.current_stream(device).cuda_stream
                rc = _native.lib().ggq_gguf_upload(self._handle(), arena.data_ptr(), 0, self.data_bytes, int(threads), int(chunk_bytes), stream)
            _native.check(rc, "ggq_gguf_upload")
        return arena

    def upload_tensors(self, device, tensors, threads=0, chunk_bytes=0):
        """Stream only ``tensors`` (a subset of ``self.tensors``, e.g. one rank's shard of the tensor list) into ONE
        device buffer: neighbours in the file are coalesced into runs, runs are packed back to back (each run start
        stays 256-byte aligned), and every run is one streamed upload.  Returns (arena, {tensor name: byte offset})."""
        device = torch.device(device)
        if device.type != "cuda":
            raise ValueError("upload_tensors() targets an AMD GPU (torch device type 'cuda')")
        order = sorted(tensors, key=lambda t: t.offset)
        runs, where, pos = [], {}, 0                      # runs: (file offset, nbytes, arena offset)
        for t in order:
            # adjacent up to the alignment padding -- and still 16-byte aligned on the device (what the HIP kernels ask of a tensor's first byte):
            # files whose general.alignment is 8 or 24 place tensors at 8 (mod 16), those start a run of their own
            if runs and t.offset <= runs[-1][0] + runs[-1][1] + self.alignment and (t.offset - runs[-1][0]) % 16 == 0:
                f0, n0, a0 = runs[-1]
                runs[-1] = (f0, max(n0, t.offset + t.nbytes - f0), a0)
            else:
                pos = (pos + 255) // 256 * 256
                runs.append((t.offset, t.nbytes, pos))
            f0, n0, a0 = runs[-1]
            where[t.name] = a0 + (t.offset - f0)
            pos = a0 + n0
        arena = torch.empty(pos, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            for f0, n0, a0 in runs:
                rc = _native.lib().ggq_gguf_upload(self._handle(), arena.data_ptr() + a0, f0, n0, int(threads), int(chunk_bytes), stream)
                _native.check(rc, "ggq_gguf_upload")
        return arena, where

    @staticmethod
    def device_bytes(arena, t):
        return arena[t.offset: t.offset + t.nbytes]
