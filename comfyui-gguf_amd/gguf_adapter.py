"""``gguf.GGUFReader`` as the reference's loader uses it (loader.py:51-141), served by the native parser of include/ggq_gguf.h.

``install(..., native_reader=True)`` (or ``GGQ_NATIVE_READER=1``; needs ``ref_loader``) rebinds the name ``gguf`` inside the reference's
``loader`` module to a proxy of the real ``gguf`` package whose ``GGUFReader`` is :class:`GGUFReaderAdapter`; every other attribute
(``GGMLQuantizationType``, ``GGUFValueType``, ``GGML_QUANT_SIZES`` ...) is the package's own.  ``gguf_sd_loader`` then runs verbatim over
it: same ownership and semantics as gguf-py -- CPU ``numpy`` views into a read-only memory map of the file, nothing of the model patcher
touched -- but the container is parsed once by the C++ reader (no per-field numpy objects: a FLUX file's 780 tensor infos + metadata in
~2 ms) and the package ``gguf`` itself is only needed for its enums.  What the loader touches, and nothing more:

    reader.tensors[i].name / .tensor_type (a member of gguf.GGMLQuantizationType) / .shape (ggml order) / .data (numpy view)
    reader.get_field(key) -> None or an object with .types (gguf.GGUFValueType members), .parts, .data  (loader.py:16-49)
"""
import numpy as np

from .gguf_file import STRING, GGUFFile


class AdapterField:
    """What loader.py reads off a ``gguf.ReaderField``: ``.types``, and the values through ``.parts[.data[i]]`` (loader.py:16-49)."""
    __slots__ = ("name", "types", "parts", "data")

    def __init__(self, value_type, field):
        self.name = field.name
        self.types = [value_type(t) for t in field.types]
        values = field.value if isinstance(field.value, tuple) else (field.value,)
        if field.types[-1] == STRING:
            self.parts = [np.frombuffer(v.encode("utf-8"), dtype=np.uint8) for v in values]
        else:
            self.parts = [np.array([v]) for v in values]
        self.data = list(range(len(self.parts)))


class AdapterTensor:
    """What loader.py reads off a ``gguf.ReaderTensor`` (loader.py:60-124).  ``data``: F32 / F16 as typed arrays (the loader ``view``s them to the
    logical shape), everything else as the packed bytes, (rows, bytes per row) like gguf-py's ``quant_shape_to_byte_shape``."""
    __slots__ = ("name", "tensor_type", "shape", "n_elements", "n_bytes", "data_offset", "_info", "_plain")

    def __init__(self, qtypes, info, data_offset):
        self._info = info
        self.name = info.name
        try:
            self.tensor_type = qtypes(int(info.tensor_type))
        except ValueError:
            self.tensor_type = int(info.tensor_type)
        self.shape = np.array(info.shape, dtype=np.uint64)
        self.n_elements, self.n_bytes, self.data_offset = info.n_elements, info.nbytes, data_offset + info.offset
        code = int(info.tensor_type)
        self._plain = {0: np.float32, 1: np.float16}.get(code)         # ggml F32 = 0, F16 = 1

    @property
    def data(self):
        raw = self._info.data.numpy()                                  # read-only view into the file's memory map
        if self._plain is not None:
            return raw.view(self._plain)
        # gguf-py's quant_shape_to_byte_shape: the torch-order dims with the last one in bytes -- (rows, bytes per row) for a 2-D weight
        lead = tuple(int(d) for d in reversed(self._info.shape[1:]))
        n = 1
        for d in lead:
            n *= d
        return raw.reshape(lead + (-1,)) if lead and n > 0 and raw.size % n == 0 else raw


class GGUFReaderAdapter:
    """``gguf.GGUFReader(path)`` for loader.py.  ``gguf_module``: the module whose enums the reference compares against."""

    gguf_module = None                                                 # set on the subclass make_reader() returns

    def __init__(self, path, mode="r"):
        g = self.gguf_module
        self._file = GGUFFile(str(path))
        self.alignment, self.data_offset = self._file.alignment, self._file.data_offset
        self.tensors = [AdapterTensor(g.GGMLQuantizationType, t, self._file.data_offset) for t in self._file.tensors]
        self._value_type = g.GGUFValueType

    def get_field(self, key):
        f = self._file.get_field(key)
        return None if f is None else AdapterField(self._value_type, f)

    def get_tensor(self, idx):
        return self.tensors[idx]


def make_reader(gguf_module):
    """The adapter class bound to ``gguf_module``'s enums."""
    return type("GGUFReader", (GGUFReaderAdapter,), {"gguf_module": gguf_module, "__doc__": GGUFReaderAdapter.__doc__})


class GGUFModuleProxy:
    """Stands in for the name ``gguf`` inside the reference's loader module: ``GGUFReader`` is the adapter, everything else the real package's."""

    def __init__(self, gguf_module):
        object.__setattr__(self, "_real", gguf_module)
        object.__setattr__(self, "GGUFReader", make_reader(gguf_module))

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_real"), name)
