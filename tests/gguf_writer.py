"""Minimal GGUF v3 writer -- TEST INFRASTRUCTURE (the product only reads GGUF).

Follows the public container layout [ggml docs/gguf.md] that include/ggq_gguf.h restates; written
independently of the native parser (pure struct.pack) so the two check each other.  The reference
writes its files with the third-party gguf.GGUFWriter (tools/convert.py:344-347), which is absent here.
"""
import struct

import numpy as np

UINT8, INT8, UINT16, INT16, UINT32, INT32, FLOAT32, BOOL, STRING, ARRAY, UINT64, INT64, FLOAT64 = range(13)
_FMT = {UINT8: "<B", INT8: "<b", UINT16: "<H", INT16: "<h", UINT32: "<I", INT32: "<i", FLOAT32: "<f", BOOL: "<?",
        UINT64: "<Q", INT64: "<q", FLOAT64: "<d"}


def _string(s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return struct.pack("<Q", len(b)) + b


def _value(vtype, value, elem_type=None):
    if vtype == STRING:
        return _string(value)
    if vtype == ARRAY:
        out = struct.pack("<IQ", elem_type, len(value))
        for v in value:
            out += _value(elem_type, v)
        return out
    return struct.pack(_FMT[vtype], value)


class GGUFWriter:
    def __init__(self, arch=None, alignment=32, version=3):
        self.kv, self.tensors, self.alignment, self.version = [], [], alignment, version
        if arch is not None:
            self.add("general.architecture", STRING, arch)
        if alignment != 32:
            self.add("general.alignment", UINT32, alignment)

    def add(self, key, vtype, value, elem_type=None):
        self.kv.append((key, vtype, value, elem_type))

    def add_tensor(self, name, qtype, ggml_dims, data):
        """ggml_dims: fastest-varying first (reversed torch shape); data: raw bytes of the packed tensor."""
        self.tensors.append((name, int(qtype), tuple(int(d) for d in ggml_dims), np.ascontiguousarray(data).view(np.uint8).reshape(-1)))

    def _head(self):
        a = self.alignment
        head = struct.pack("<IIQQ", 0x46554747, self.version, len(self.tensors), len(self.kv))
        for key, vtype, value, elem_type in self.kv:
            head += _string(key) + struct.pack("<I", vtype) + _value(vtype, value, elem_type)
        off = 0
        for name, qtype, dims, data in self.tensors:
            head += _string(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims)
            head += struct.pack("<IQ", qtype, off)
            off = (off + data.size + a - 1) // a * a
        return head + b"\0" * ((-len(head)) % a)

    def write(self, path):
        """Header, then every tensor padded to the alignment, streamed (multi-GB files are fine)."""
        a = self.alignment
        with open(path, "wb") as f:
            f.write(self._head())
            for _, _, _, data in self.tensors:
                f.write(memoryview(data))
                f.write(b"\0" * ((-data.size) % a))
        return path

    def tobytes(self):
        a = self.alignment
        return self._head() + b"".join(d.tobytes() + b"\0" * ((-d.size) % a) for _, _, _, d in self.tensors)
