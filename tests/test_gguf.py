"""The GGUF container reader (include/ggq_gguf.h, native) and the gguf_sd_loader mirror, on CPU.

Parity note: the reference reads these files through the third-party gguf.GGUFReader (loader.py:55;
gguf>=0.13.0, unpinned, absent from /root/reference and from this image), and ships no .gguf file and
no loader test.  The parser is therefore pinned to the PUBLIC container layout: (a) a byte-for-byte
hand-assembled file (below), (b) an independent pure-python writer (tests/gguf_writer.py).  The loader
mirror -- everything done WITH the parsed file -- is pinned to the reference's own loader.py, executed verbatim on
the same files behind a gguf-py-shaped adapter over our parser (last test; this container only), and
otherwise checked against the behaviour read off loader.py:51-141."""
import os
import struct

import numpy as np
import pytest
import torch

from gguf_writer import (ARRAY, BOOL, FLOAT32, FLOAT64, INT8, INT16, INT32, INT64, STRING, UINT8, UINT16, UINT32, UINT64,
                         GGUFWriter)


def _s(b):
    return struct.pack("<Q", len(b)) + b


# A complete GGUF v3 file assembled by hand from the layout in include/ggq_gguf.h:
# 2 KV pairs, 2 tensors (F32 [3] and Q8_0 [32 x 2]), alignment 32.
_HAND_HEAD = (
    b"GGUF" + struct.pack("<I", 3) + struct.pack("<Q", 2) + struct.pack("<Q", 2)
    + _s(b"general.architecture") + struct.pack("<I", 8) + _s(b"flux")
    + _s(b"answer") + struct.pack("<I", 4) + struct.pack("<I", 42)
    + _s(b"bias") + struct.pack("<I", 1) + struct.pack("<Q", 3) + struct.pack("<I", 0) + struct.pack("<Q", 0)
    + _s(b"w.weight") + struct.pack("<I", 2) + struct.pack("<Q", 32) + struct.pack("<Q", 2) + struct.pack("<I", 8) + struct.pack("<Q", 32)
)
_HAND_PAD = (-len(_HAND_HEAD)) % 32
_HAND_BIAS = struct.pack("<3f", 1.0, -2.5, 3.25)
_HAND_Q8 = bytes(range(68))
HAND_FILE = _HAND_HEAD + b"\0" * _HAND_PAD + _HAND_BIAS + b"\0" * 20 + _HAND_Q8


@pytest.fixture()
def gf(pkg):
    return pkg.gguf_file


def _write(tmp_path, data, name="t.gguf"):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


def test_hand_assembled_file(pkg, gf, tmp_path):
    Q = pkg.qtypes.Q
    with gf.GGUFFile(_write(tmp_path, HAND_FILE)) as f:
        assert (f.version, f.alignment, f.n_kv, len(f.tensors)) == (3, 32, 2, 2)
        assert f.data_offset == len(_HAND_HEAD) + _HAND_PAD and f.data_bytes == 32 + 68
        assert f.keys() == ["general.architecture", "answer"]
        assert f.get_field("general.architecture") == gf.GGUFField("general.architecture", [gf.STRING], "flux")
        assert f.get_field("answer").value == 42 and f.get_field("answer").types == [gf.UINT32]
        assert f.get_field("nope") is None
        b, w = f.tensors
        assert (b.name, b.tensor_type, b.shape, b.offset, b.nbytes, b.n_elements) == ("bias", Q.F32, (3,), 0, 12, 3)
        assert (w.name, w.tensor_type, w.shape, w.offset, w.nbytes, w.n_elements) == ("w.weight", Q.Q8_0, (32, 2), 32, 68, 64)
        assert b.data.dtype == torch.uint8 and b.data.view(torch.float32).tolist() == [1.0, -2.5, 3.25]
        assert bytes(w.data.numpy()) == _HAND_Q8
    with pytest.raises(ValueError, match="closed"):
        f.get_field("answer")
    assert bytes(w.data.numpy()) == _HAND_Q8          # CPU views outlive the handle (own mapping)


def test_every_value_type_round_trips(pkg, gf, tmp_path):
    w = GGUFWriter(arch="sd3")
    scalars = [("u8", UINT8, 200), ("i8", INT8, -7), ("u16", UINT16, 65535), ("i16", INT16, -3), ("u32", UINT32, 4000000000),
               ("i32", INT32, -5), ("f32", FLOAT32, 1.5), ("b", BOOL, True), ("u64", UINT64, 2**63 + 5), ("i64", INT64, -2**62),
               ("f64", FLOAT64, 1e-300), ("s", STRING, "héllo ✓"), ("empty", STRING, "")]
    for k, t, v in scalars:
        w.add(k, t, v)
    w.add("arr.i32", ARRAY, [3072, 64, -1], INT32)
    w.add("arr.f32", ARRAY, [0.5, -0.25], FLOAT32)
    w.add("arr.str", ARRAY, ["<pad>", "", "▁the", "x" * 300], STRING)
    w.add("arr.empty", ARRAY, [], UINT8)
    w.add("arr.bool", ARRAY, [True, False, True], BOOL)
    with gf.GGUFFile(w.write(str(tmp_path / "kv.gguf"))) as f:
        assert f.n_kv == len(scalars) + 6 and len(f.tensors) == 0 and f.data_bytes == 0
        for k, t, v in scalars:
            fld = f.get_field(k)
            assert fld.types == [t] and fld.value == v and type(fld.value) is type(v), k
        assert f.get_field("arr.i32") == gf.GGUFField("arr.i32", [ARRAY, INT32], (3072, 64, -1))
        assert f.get_field("arr.f32").value == (0.5, -0.25)
        assert f.get_field("arr.str").value == ("<pad>", "", "▁the", "x" * 300) and f.get_field("arr.str").types == [ARRAY, STRING]
        assert f.get_field("arr.empty").value == () and f.get_field("arr.bool").value == (True, False, True)
        # the reference's accessors (loader.py:26-49)
        ld = pkg.loader
        assert ld.get_field(f, "general.architecture", str) == "sd3" and ld.get_field(f, "i32", int) == -5
        assert ld.get_field(f, "f32", float) == 1.5 and ld.get_field(f, "b", bool) is True and ld.get_field(f, "zz", int) is None
        with pytest.raises(TypeError, match="expected string"):
            ld.get_field(f, "u8", str)
        with pytest.raises(TypeError, match="Unknown field type"):
            ld.get_field(f, "u8", bytes)
        assert ld.get_list_field(f, "arr.str", str)[2] == "▁the" and ld.get_list_field(f, "arr.f32", float) == (0.5, -0.25)
        assert ld.get_list_field(f, "arr.i32", int) == (3072, 64, -1) and ld.get_list_field(f, "zz", int) is None


@pytest.mark.parametrize("alignment,version", [(32, 3), (64, 3), (256, 2), (8, 3)])
def test_tensor_table_offsets_and_views(pkg, gf, tmp_path, alignment, version):
    Q, synth = pkg.qtypes.Q, pkg.synth
    w = GGUFWriter(arch="flux", alignment=alignment, version=version)
    spec = [("a.weight", Q.Q4_K, (4, 512)), ("b.weight", Q.Q8_0, (3, 96)), ("c.bias", Q.F32, (7,)), ("d.weight", Q.Q6_K, (1, 256)),
            ("e.scale", Q.F16, (5,)), ("f.weight", Q.IQ4_XS, (2, 2, 256)), ("g.w", Q.BF16, (3, 4))]
    raw = {}
    for i, (name, q, shape) in enumerate(spec):
        if q in (Q.F32, Q.F16, Q.BF16):
            n = int(np.prod(shape)) * (4 if q == Q.F32 else 2)
            data = np.random.default_rng(i).integers(0, 256, n, dtype=np.uint8)
        else:
            data = synth.make_tensor_bytes(q, shape, seed=i)
        raw[name] = data
        w.add_tensor(name, q, tuple(reversed(shape)), data)
    path = w.write(str(tmp_path / "t.gguf"))
    with gf.GGUFFile(path) as f:
        assert f.version == version and f.alignment == alignment and f.data_offset % alignment == 0
        assert [t.name for t in f.tensors] == [s[0] for s in spec]
        whole = np.fromfile(path, dtype=np.uint8)
        for t, (name, q, shape) in zip(f.tensors, spec):
            assert t.tensor_type == q and t.shape == tuple(reversed(shape)) and t.offset % alignment == 0
            assert t.nbytes == raw[name].size and t.n_elements == int(np.prod(shape))
            assert np.array_equal(t.data.numpy(), raw[name])
            assert np.array_equal(whole[f.data_offset + t.offset: f.data_offset + t.offset + t.nbytes], raw[name])
        assert f.file_bytes == whole.size == f.data_offset + f.data_bytes


def test_type_geometry_table(pkg):
    import ctypes
    L, qt = pkg._native.lib(), pkg.qtypes
    bs, ts = ctypes.c_uint32(), ctypes.c_uint32()
    for q, geo in qt.GGML_QUANT_SIZES.items():
        assert L.ggq_ggml_type_geometry(int(q), ctypes.byref(bs), ctypes.byref(ts)) == 0 and (bs.value, ts.value) == geo, q
    assert L.ggq_ggml_type_geometry(4, ctypes.byref(bs), ctypes.byref(ts)) != 0 and (bs.value, ts.value) == (0, 0)   # removed Q4_2


def test_corrupt_and_truncated_files_are_rejected_not_crashed_on(pkg, gf, tmp_path):
    good = HAND_FILE
    with pytest.raises(OSError):
        gf.GGUFFile(str(tmp_path / "missing.gguf"))
    cases = {
        "magic": b"GGML" + good[4:],
        "version1": good[:4] + struct.pack("<I", 1) + good[8:],
        "bigendian": good[:4] + struct.pack(">I", 3) + good[8:],
        "huge_kv_count": good[:16] + struct.pack("<Q", 2**60) + good[24:],
        "huge_tensor_count": good[:8] + struct.pack("<Q", 2**61) + good[16:],
        "short": good[:20],
        "empty": b"",
    }
    for name, data in cases.items():
        with pytest.raises(ValueError, match="GGUF"):
            gf.GGUFFile(_write(tmp_path, data, f"{name}.gguf"))
    # every truncation point: either a clean parse error or (past the last byte any table needs) success
    for cut in range(0, len(good)):
        p = _write(tmp_path, good[:cut], "cut.gguf")
        try:
            gf.GGUFFile(p).close()
            ok = True
        except ValueError:
            ok = False
        assert ok == (cut >= len(good)), cut       # the last tensor's bytes end the file: any cut loses data
    # single-field corruptions of the tensor table
    head = bytearray(good)
    q8 = good.index(b"w.weight") + 8
    bad_dims = bytes(head[:q8]) + struct.pack("<I", 9) + bytes(head[q8 + 4:])
    with pytest.raises(ValueError):
        gf.GGUFFile(_write(tmp_path, bad_dims, "dims.gguf"))
    off_pos = q8 + 4 + 16 + 4
    assert struct.unpack_from("<Q", good, off_pos)[0] == 32
    for bad_off in (16, 64, 2**40):                                       # misaligned / past the end
        data = good[:off_pos] + struct.pack("<Q", bad_off) + good[off_pos + 8:]
        with pytest.raises(ValueError):
            gf.GGUFFile(_write(tmp_path, data, "off.gguf"))
    bad_row = good[:q8 + 4] + struct.pack("<Q", 33) + good[q8 + 12:]      # Q8_0 row of 33 elements: not whole blocks
    with pytest.raises(ValueError):
        gf.GGUFFile(_write(tmp_path, bad_row, "row.gguf"))
    # random byte flips in the header must never crash the process
    rng = np.random.default_rng(0)
    for _ in range(300):
        b = bytearray(good)
        for pos in rng.integers(0, len(_HAND_HEAD), size=3):
            b[pos] = rng.integers(0, 256)
        try:
            gf.GGUFFile(_write(tmp_path, bytes(b), "fuzz.gguf")).close()
        except (ValueError, UnicodeDecodeError):
            pass


def _model_file(pkg, tmp_path, arch="flux", prefix="model.diffusion_model.", extra_kv=()):
    Q, synth = pkg.qtypes.Q, pkg.synth
    w = GGUFWriter(arch=arch)
    for kv in extra_kv:
        w.add(*kv)
    spec = [(prefix + "double_blocks.0.img_attn.qkv.weight", Q.Q5_K, (24, 256)), (prefix + "double_blocks.0.img_attn.proj.weight", Q.Q4_K, (8, 512)),
            (prefix + "img_in.bias", Q.F32, (16,)), (prefix + "img_in.weight", Q.F16, (16, 4)), (prefix + "norm.scale", Q.BF16, (6,)),
            ("other.tensor", Q.Q8_0, (2, 32)), (prefix + "x.conv.weight", Q.Q8_0, (2, 3, 1, 32))]
    packed = {}
    for i, (name, q, shape) in enumerate(spec):
        n = int(np.prod(shape))
        if q == Q.F32:
            data = np.arange(n, dtype=np.float32) * 0.5
        elif q == Q.F16:
            data = (np.arange(n, dtype=np.float32) - 3).astype(np.float16)
        elif q == Q.BF16:
            data = (np.array([1.0, -2.0, 0.5, 3.0, 100.0, -0.125], dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        else:
            data = synth.make_tensor_bytes(q, shape, seed=40 + i, mode="signed")
        packed[name] = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        w.add_tensor(name, q, tuple(reversed(shape)), data)
    return w.write(str(tmp_path / f"{arch}.gguf")), spec, packed


def test_gguf_sd_loader_mirrors_the_reference(pkg, tmp_path):
    import oracle
    Q, ld = pkg.qtypes.Q, pkg.loader
    pre = "model.diffusion_model."
    path, spec, packed = _model_file(pkg, tmp_path, extra_kv=[(f"comfy.gguf.orig_shape.{pre}x.conv.weight", ARRAY, [2, 3, 32], INT32)])
    sd, arch = ld.gguf_sd_loader(path, return_arch=True)
    assert arch == "flux"
    # prefix present -> only prefixed tensors, prefix stripped (loader.py:57-71)
    assert set(sd) == {n[len(pre):] for n, _, _ in spec if n.startswith(pre)} and "other.tensor" not in sd
    qkv = sd["double_blocks.0.img_attn.qkv.weight"]
    assert isinstance(qkv, pkg.ops.GGMLTensor) and qkv.tensor_type == Q.Q5_K and qkv.shape == torch.Size((24, 256))
    assert qkv.dtype == torch.uint8 and not qkv.is_cuda and np.array_equal(qkv.numpy(), packed[pre + "double_blocks.0.img_attn.qkv.weight"])
    assert getattr(qkv, "is_largest_weight", False) and not getattr(sd["double_blocks.0.img_attn.proj.weight"], "is_largest_weight", False)
    # F32 / F16 are viewed to their dtype and logical shape (loader.py:119-120)
    assert sd["img_in.bias"].dtype == torch.float32 and torch.equal(torch.Tensor(sd["img_in.bias"]), torch.arange(16) * 0.5)
    assert sd["img_in.weight"].dtype == torch.float16 and tuple(sd["img_in.weight"].size()) == (16, 4)
    # 1-D BF16 is dequantized to fp32 at load (loader.py:123-125)
    assert sd["norm.scale"].dtype == torch.float32 and sd["norm.scale"].tolist() == [1.0, -2.0, 0.5, 3.0, 100.0, -0.125]
    # comfy.gguf.orig_shape.* overrides the reversed ggml dims (loader.py:16-24,108-110)
    assert sd["x.conv.weight"].shape == torch.Size((2, 3, 32)) and sd["x.conv.weight"].tensor_type == Q.Q8_0
    # the packed views feed the oracle to the same values the file was built from
    want = oracle.dequant_f16(Q.Q4_K, packed[pre + "double_blocks.0.img_attn.proj.weight"])
    assert np.array_equal(oracle.dequant_f16(Q.Q4_K, sd["double_blocks.0.img_attn.proj.weight"].numpy()), want)
    # no prefix in the file -> every tensor, names untouched; handle_prefix=None likewise
    path2, spec2, _ = _model_file(pkg, tmp_path, arch="sd3", prefix="")
    assert set(ld.gguf_sd_loader(path2)) == {n for n, _, _ in spec2}
    assert set(ld.gguf_sd_loader(path, handle_prefix=None)) == {n for n, _, _ in spec}


def test_gguf_sd_loader_architecture_checks(pkg, tmp_path):
    ld = pkg.loader
    path_t5, _, _ = _model_file(pkg, tmp_path, arch="t5", prefix="")
    with pytest.raises(ValueError, match="Unexpected architecture type in GGUF file: 't5'"):
        ld.gguf_sd_loader(path_t5)
    assert ld.gguf_sd_loader(path_t5, is_text_model=True, return_arch=True)[1] == "t5"
    path_flux, _, _ = _model_file(pkg, tmp_path, arch="flux")
    with pytest.raises(ValueError, match="Unexpected text model architecture"):
        ld.gguf_sd_loader(path_flux, is_text_model=True)
    path_vis, _, _ = _model_file(pkg, tmp_path, arch="clip", prefix="", extra_kv=[("general.type", STRING, "mmproj")])
    assert ld.gguf_sd_loader(path_vis, is_text_model=True, return_arch=True)[1] == "clip"
    # no architecture key: sd.cpp compatibility mode needs the (control-plane) detector
    w = GGUFWriter(arch=None)
    w.add_tensor("a.weight", pkg.qtypes.Q.F32, (4,), np.zeros(4, np.float32))
    p = w.write(str(tmp_path / "noarch.gguf"))
    with pytest.raises(ValueError, match="incompatible with llama.cpp"):
        ld.gguf_sd_loader(p, is_text_model=True)
    with pytest.raises(ValueError, match="not currently supported"):
        ld.gguf_sd_loader(p)

    class _Arch:
        arch = "sdxl"
    sd, arch = ld.gguf_sd_loader(p, return_arch=True, detect_arch=lambda keys: _Arch())
    assert arch == "sdxl" and set(sd) == {"a.weight"}
    w = GGUFWriter(arch="flux")
    w.add("comfy.gguf.orig_shape.a.weight", ARRAY, [2, 2], INT64)
    w.add_tensor("a.weight", pkg.qtypes.Q.F32, (4,), np.zeros(4, np.float32))
    with pytest.raises(TypeError, match="Bad original shape metadata"):
        ld.gguf_sd_loader(w.write(str(tmp_path / "badshape.gguf")))


def test_upload_argument_checks_without_a_gpu(pkg, gf, tmp_path):
    nat = pkg._native
    with gf.GGUFFile(_write(tmp_path, HAND_FILE)) as f:
        L = nat.lib()
        assert L.ggq_gguf_upload(f._h, None, 0, 0, 0, 0, None) == nat.GGQ_OK                 # nothing to copy
        assert L.ggq_gguf_upload(f._h, None, 0, 10, 0, 0, None) == nat.GGQ_ERR_ARG           # NULL destination
        assert L.ggq_gguf_upload(f._h, 4096, 90, 20, 0, 0, None) == nat.GGQ_ERR_ARG          # range past the data section
        assert L.ggq_gguf_upload(None, 4096, 0, 10, 0, 0, None) == nat.GGQ_ERR_ARG
        with pytest.raises(ValueError, match="AMD GPU"):
            f.upload("cpu")
    assert b"GGUF" in nat.lib().ggq_strerror(nat.GGQ_ERR_FORMAT)


# ---------------------------------------------------------------- the reference's loader.py, run verbatim

def _reference_loader(pkg, monkeypatch):
    """loader.py of /root/reference executed from its own source, with the two things it needs from outside stubbed: `comfy`
    (test_host._fake_comfy) and `gguf` -- the enum / sizes stub of oracle/reference.py plus, as its GGUFReader, the PRODUCT's adapter
    (comfyui-gguf_amd/gguf_adapter.py: our parser's view of the file in gguf-py's attribute layout, what install(native_reader=True) puts under the
    reference's loader).  So the container parsing is not what this pins; everything the loader does with it is."""
    import enum
    import importlib.util
    import sys
    import types
    from oracle import reference
    from test_host import _fake_comfy
    reference.ensure_gguf()
    stub = sys.modules["gguf"]
    Q = pkg.qtypes.Q
    vt = enum.IntEnum("GGUFValueType", dict(UINT8=0, INT8=1, UINT16=2, INT16=3, UINT32=4, INT32=5, FLOAT32=6, BOOL=7, STRING=8, ARRAY=9,
                                            UINT64=10, INT64=11, FLOAT64=12))

    monkeypatch.setattr(stub, "GGUFValueType", vt, raising=False)
    monkeypatch.setattr(stub, "GGUFReader", pkg.gguf_adapter.make_reader(stub), raising=False)
    for k, v in _fake_comfy().items():
        monkeypatch.setitem(sys.modules, k, v)
    root = types.ModuleType("refldr")
    root.__path__ = [reference.REFERENCE_DIR]
    monkeypatch.setitem(sys.modules, "refldr", root)
    tools = types.ModuleType("refldr.tools")
    tools.__path__ = [os.path.join(reference.REFERENCE_DIR, "tools")]
    monkeypatch.setitem(sys.modules, "refldr.tools", tools)
    mods = {}
    for name in ("dequant", "ops", "loader"):
        spec = importlib.util.spec_from_file_location(f"refldr.{name}", os.path.join(reference.REFERENCE_DIR, f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        monkeypatch.setitem(sys.modules, f"refldr.{name}", m)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["loader"]


def _outcome(fn, *a, **k):
    try:
        return fn(*a, **k), None
    except Exception as e:                                     # noqa: BLE001 -- the comparison is about which exception
        return None, e


def _same_state_dict(ours, theirs):
    assert list(ours) == list(theirs)                          # same keys, same order
    for key in ours:
        a, b = ours[key], theirs[key]
        ta, tb = getattr(a, "tensor_type", None), getattr(b, "tensor_type", None)        # None: a plain (dequantized) tensor
        assert (None if ta is None else int(ta)) == (None if tb is None else int(tb)), key
        assert tuple(a.shape) == tuple(b.shape) and a.dtype == b.dtype and a.numel() == b.numel(), key
        # (the packed bytes of a quantized tensor: gguf-py -- and the adapter -- hand them out as (rows, bytes per row), this package's loader flat)
        assert torch.equal(a.as_subclass(torch.Tensor).reshape(-1), b.as_subclass(torch.Tensor).reshape(-1)), key
        assert bool(getattr(a, "is_largest_weight", False)) == bool(getattr(b, "is_largest_weight", False)), key


LOADER_CALLS = [dict(), dict(return_arch=True), dict(handle_prefix=None), dict(handle_prefix="other."), dict(is_text_model=True, return_arch=True),
                dict(handle_prefix="", return_arch=True)]


def _loader_case_files(pkg, tmp_path):
    """The files the loader is compared on (deterministic: same bytes wherever they are built)."""
    Q = pkg.qtypes.Q
    pre = "model.diffusion_model."
    files = {
        "flux": _model_file(pkg, tmp_path, extra_kv=[(f"comfy.gguf.orig_shape.{pre}x.conv.weight", ARRAY, [2, 3, 32], INT32),
                                                     ("tokens", ARRAY, ["a", "bc", ""], STRING), ("merges", ARRAY, [3, 1, 2], INT32),
                                                     ("pi", FLOAT32, 3.25), ("flag", BOOL, True), ("n", UINT64, 2 ** 40)])[0],
        "sd3": _model_file(pkg, tmp_path, arch="sd3", prefix="")[0],
        "t5": _model_file(pkg, tmp_path, arch="t5", prefix="")[0],
        "mmproj": _model_file(pkg, tmp_path, arch="clip", prefix="", extra_kv=[("general.type", STRING, "mmproj")])[0],
        "unknown": _model_file(pkg, tmp_path, arch="nonsense")[0],
    }
    w = GGUFWriter(arch=None)
    w.add_tensor("a.weight", Q.F32, (4,), np.zeros(4, np.float32))
    files["noarch"] = w.write(str(tmp_path / "noarch_live.gguf"))
    w = GGUFWriter(arch="flux")
    w.add("comfy.gguf.orig_shape.a.weight", ARRAY, [2, 2], INT64)
    w.add_tensor("a.weight", Q.F32, (4,), np.zeros(4, np.float32))
    files["badshape"] = w.write(str(tmp_path / "badshape_live.gguf"))
    w = GGUFWriter(arch="flux")
    w.add("general.architecture2", INT32, 7)
    w.add_tensor("a.weight", Q.F32, (4,), np.zeros(4, np.float32))
    files["plain"] = w.write(str(tmp_path / "plain_live.gguf"))

    # files without architecture metadata (stable-diffusion.cpp exports, "pig"): the compatibility branch, which asks the reference's
    # key-based detector (tools/convert.py detect_arch, control plane -- handed to our loader as a callable)
    for label, arch, names in (("compat_flux", None, [(pre + "double_blocks.0.img_attn.proj.weight", Q.Q4_K, (8, 512))]),
                               ("compat_pig", "pig", [("double_blocks.0.img_attn.proj.weight", Q.Q4_K, (8, 512))]),
                               ("compat_sdxl", None, [("label_emb.0.0.weight", Q.F32, (4,)), ("input_blocks.4.1.proj_in.weight", Q.F16, (8, 4, 1, 1)),
                                                      ("input_blocks.4.1.proj_out.weight", Q.F16, (8, 4, 1, 1)), ("mid.conv.weight", Q.F16, (2, 2, 1, 1))])):
        w = GGUFWriter(arch=arch)
        for i, (name, q, shape) in enumerate(names):
            n = int(np.prod(shape))
            data = np.arange(n, dtype=np.float32) if q == Q.F32 else np.arange(n, dtype=np.float16) if q == Q.F16 else pkg.synth.make_tensor_bytes(q, shape, seed=90 + i)
            w.add_tensor(name, q, tuple(reversed(shape)), data)
        files[label] = w.write(str(tmp_path / f"{label}.gguf"))
    return files


@pytest.mark.filterwarnings("ignore::DeprecationWarning")          # the reference's int(np.array([x])) under numpy 2
@pytest.mark.skipif(not os.path.isfile("/root/reference/loader.py"), reason="/root/reference not present (GPU box)")
def test_loader_equals_the_reference_loader_run_live(pkg, tmp_path, monkeypatch):
    """gguf_sd_loader / get_field / get_list_field / get_orig_shape against the reference's own functions executed verbatim on
    the same files: same state dicts (keys and their order, types, logical shapes, dtypes, bytes, largest-weight mark),
    same architecture, same exception types and messages."""
    ours = pkg.loader
    ref = _reference_loader(pkg, monkeypatch)
    pre = "model.diffusion_model."
    files = _loader_case_files(pkg, tmp_path)
    import importlib
    detector = importlib.import_module("refldr.tools.convert").detect_arch

    calls = LOADER_CALLS
    compared = 0
    for label, path in files.items():
        for kw in calls:
            (got, e1), (want, e2) = _outcome(ours.gguf_sd_loader, path, detect_arch=detector, **kw), _outcome(ref.gguf_sd_loader, path, **kw)
            assert (e1 is None) == (e2 is None), (label, kw, e1, e2)
            if e2 is not None:
                assert type(e1) is type(e2), (label, kw, e1, e2)
                assert str(e1).split(", got ")[0] == str(e2).split(", got ")[0], (label, kw)     # enum reprs differ after "got"
                compared += 1
                continue
            if kw.get("return_arch"):
                assert got[1] == want[1], (label, kw)
                got, want = got[0], want[0]
            _same_state_dict(got, want)
            compared += 1
    assert compared == len(files) * len(calls)
    # the metadata accessors on their own
    a, b = pkg.gguf_file.GGUFFile(files["flux"]), sys_modules_reader(files["flux"])
    for key, ftype in [("general.architecture", str), ("pi", float), ("flag", bool), ("n", int), ("merges", int), ("missing", int)]:
        assert _outcome(ours.get_field, a, key, ftype)[0] == _outcome(ref.get_field, b, key, ftype)[0], key
    for key, ftype in [("tokens", str), ("merges", int), ("merges", float), ("missing", str)]:
        assert ours.get_list_field(a, key, ftype) == ref.get_list_field(b, key, ftype), key
    for key, ftype in [("pi", str), ("general.architecture", dict)]:
        (_, e1), (_, e2) = _outcome(ours.get_field, a, key, ftype), _outcome(ref.get_field, b, key, ftype)
        assert type(e1) is type(e2) is TypeError and str(e1).split(", got ")[0] == str(e2).split(", got ")[0]
    assert ours.get_orig_shape(a, pre + "x.conv.weight") == ref.get_orig_shape(b, pre + "x.conv.weight") == torch.Size((2, 3, 32))
    assert ours.get_orig_shape(a, "nope") is ref.get_orig_shape(b, "nope") is None
    a.close()


@pytest.mark.filterwarnings("ignore::DeprecationWarning")
@pytest.mark.skipif(not os.path.isfile("/root/reference/loader.py"), reason="/root/reference not present (GPU box)")
def test_install_native_reader_puts_the_adapter_under_the_reference_loader(pkg, tmp_path, monkeypatch):
    """install(native_reader=True) (Next #6): the reference's verbatim gguf_sd_loader, with nothing but the name `gguf` in its module rebound,
    reads the mixed fixture files through the native parser and returns what this package's own loader returns -- keys, order, types, logical shapes,
    dtypes, bytes, largest-weight mark, architecture -- and uninstall() puts the real module back.  The real `gguf` here is a stub whose GGUFReader
    RAISES: the state dicts can only have come through the adapter."""
    import sys
    ref = _reference_loader(pkg, monkeypatch)
    stub = sys.modules["gguf"]

    def no_reader(path):
        raise AssertionError("gguf.GGUFReader was called: the native reader is not in place")
    monkeypatch.setattr(stub, "GGUFReader", no_reader, raising=False)
    rd, ro = sys.modules["refldr.dequant"], sys.modules["refldr.ops"]
    files = _loader_case_files(pkg, tmp_path)
    with pytest.raises(ValueError):
        pkg.install.install(rd, ro, native_reader=True)                     # needs ref_loader
    assert rd.dequantize_tensor.__module__ == "refldr.dequant"              # ... and left nothing patched behind
    pkg.install.install(rd, ro, ref, native_reader=True, exact=True)
    try:
        assert type(ref.gguf).__name__ == "GGUFModuleProxy" and ref.gguf.GGMLQuantizationType is stub.GGMLQuantizationType
        for label in ("flux", "sd3", "t5", "plain"):
            for kw in (dict(return_arch=True), dict(handle_prefix=None), dict(is_text_model=True, return_arch=True)):
                (got, e1), (want, e2) = _outcome(ref.gguf_sd_loader, files[label], **kw), _outcome(pkg.loader.gguf_sd_loader, files[label], **kw)
                assert (e1 is None) == (e2 is None) and type(e1) is type(e2), (label, kw, e1, e2)
                if e1 is None:
                    if kw.get("return_arch"):
                        assert got[1] == want[1]
                        got, want = got[0], want[0]
                    _same_state_dict(want, got)
        assert "native_reader=True" in pkg.install.describe(rd)
    finally:
        pkg.install.uninstall(rd)
    assert ref.gguf is stub
    with pytest.raises(AssertionError):
        ref.gguf_sd_loader(files["flux"])


def sys_modules_reader(path):
    import sys
    return sys.modules["gguf"].GGUFReader(path)


# ---------------------------------------------------------------- random containers (property test)

def test_random_containers_round_trip(pkg, gf, tmp_path):
    """Random metadata (every scalar type, strings with arbitrary unicode, arrays) and random tensor tables (types, ranks,
    alignments, versions) written by the independent writer come back from the native parser value for value and byte for byte."""
    hyp = pytest.importorskip("hypothesis")
    st = pytest.importorskip("hypothesis.strategies")
    Q = pkg.qtypes.Q
    ints = {UINT8: (0, 2 ** 8 - 1), INT8: (-2 ** 7, 2 ** 7 - 1), UINT16: (0, 2 ** 16 - 1), INT16: (-2 ** 15, 2 ** 15 - 1), UINT32: (0, 2 ** 32 - 1),
            INT32: (-2 ** 31, 2 ** 31 - 1), UINT64: (0, 2 ** 64 - 1), INT64: (-2 ** 63, 2 ** 63 - 1)}
    text = st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=40)

    def scalar(t):
        if t in ints:
            return st.integers(*ints[t])
        if t == FLOAT32:
            return st.floats(width=32, allow_nan=False)
        if t == FLOAT64:
            return st.floats(allow_nan=False)
        return st.booleans() if t == BOOL else text

    scalar_types = list(ints) + [FLOAT32, FLOAT64, BOOL, STRING]
    kv = st.one_of([st.tuples(st.just(t), scalar(t), st.none()) for t in scalar_types]
                   + [st.tuples(st.just(ARRAY), st.lists(scalar(t), max_size=6), st.just(t)) for t in scalar_types])
    qtypes = [Q.F32, Q.F16, Q.BF16, Q.Q4_0, Q.Q8_0, Q.Q4_K, Q.Q6_K, Q.IQ4_XS, Q.Q2_K]
    tensor = st.tuples(st.sampled_from(qtypes), st.lists(st.integers(1, 3), min_size=0, max_size=3), st.integers(1, 3))

    @hyp.settings(max_examples=60, deadline=None, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(kvs=st.lists(kv, max_size=8), tensors=st.lists(tensor, max_size=5), alignment=st.sampled_from([8, 16, 32, 64, 4096]),
               version=st.sampled_from([2, 3]), seed=st.integers(0, 2 ** 16))
    def check(kvs, tensors, alignment, version, seed):
        w = GGUFWriter(arch="flux", alignment=alignment, version=version)
        for i, (t, v, et) in enumerate(kvs):
            w.add(f"k{i}.é", t, v, et)
        rng, raw = np.random.default_rng(seed), []
        for i, (q, outer, row_blocks) in enumerate(tensors):
            bs, ts = pkg.qtypes.GGML_QUANT_SIZES[q]
            dims = (row_blocks * bs,) + tuple(outer)                      # ggml order: the row (whole blocks) first
            data = rng.integers(0, 256, int(np.prod(dims)) // bs * ts, dtype=np.uint8)
            raw.append((f"t{i}.weight", q, dims, data))
            w.add_tensor(f"t{i}.weight", q, dims, data)
        path = w.write(str(tmp_path / "prop.gguf"))
        with gf.GGUFFile(path) as f:
            assert f.version == version and f.alignment == alignment and len(f.tensors) == len(raw)
            for i, (t, v, et) in enumerate(kvs):
                fld = f.get_field(f"k{i}.é")
                assert fld.types == ([t] if t != ARRAY else [ARRAY, et])
                got, want = (fld.value, tuple(v)) if t == ARRAY else ((fld.value,), (v,))
                et = t if t != ARRAY else et
                if et == FLOAT32:
                    want = tuple(float(np.float32(x)) for x in want)
                assert got == want and all(type(a) is type(b) for a, b in zip(got, want)), (t, et)
            for info, (name, q, dims, data) in zip(f.tensors, raw):
                assert (info.name, info.tensor_type, info.shape, info.nbytes) == (name, q, dims, data.size)
                assert info.offset % alignment == 0 and np.array_equal(info.data.numpy(), data)

    check()


# ---------------------------------------------------------------- the same comparison against COMMITTED reference outcomes

def _stand_in_detector(keys):
    """Architecture of a file without metadata, for the two shapes the case files use (the reference's own detector,
    tools/convert.py, produced the committed expectations; it does not travel to the GPU box)."""
    class _A:
        arch = "flux" if "double_blocks.0.img_attn.proj.weight" in keys else "sdxl" if "label_emb.0.0.weight" in keys else None
    if _A.arch is None:
        raise AssertionError("Unknown model architecture!")
    return _A()


def _loader_outcome(fn, path, tmp_path, **kw):
    """JSON-able summary of one gguf_sd_loader call: the state dict entry by entry (bytes by sha256), the architecture, or the error."""
    import hashlib
    try:
        res = fn(path, **kw)
    except Exception as e:                                     # noqa: BLE001
        return {"error": type(e).__name__, "message": str(e).split(", got ")[0].replace(str(tmp_path), "<dir>")}
    sd, arch = res if kw.get("return_arch") else (res, None)
    entries = []
    for key, t in sd.items():
        tt = getattr(t, "tensor_type", None)
        raw = t.as_subclass(torch.Tensor).contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()
        entries.append([key, None if tt is None else int(tt), [int(d) for d in t.shape], str(t.dtype), [int(d) for d in t.size()],
                        hashlib.sha256(raw).hexdigest()[:16], bool(getattr(t, "is_largest_weight", False))])
    return {"arch": arch, "entries": entries}


def write_loader_golden(pkg, out_path, tmp_path):
    """tests/golden/loader_cases.json: the REFERENCE loader's outcome (loader.py executed verbatim, _reference_loader) for every
    case file x call.  Run by tests/make_loader_golden.py in the build container."""
    import json
    from _pytest.monkeypatch import MonkeyPatch
    with MonkeyPatch.context() as mp:
        ref = _reference_loader(pkg, mp)
        files = _loader_case_files(pkg, tmp_path)
        cases = {label: [_loader_outcome(ref.gguf_sd_loader, path, tmp_path, **kw) for kw in LOADER_CALLS] for label, path in files.items()}
    with open(out_path, "w") as f:
        json.dump({"calls": [{k: v for k, v in kw.items()} for kw in LOADER_CALLS], "cases": cases}, f, indent=1, sort_keys=True)
    return cases


def test_loader_matches_the_committed_reference_outcomes(pkg, tmp_path, golden_dir):
    """Everywhere (GPU box included): the case files are rebuilt byte for byte and gguf_sd_loader must give what the reference's
    loader gave for them when tests/golden/loader_cases.json was generated."""
    import json
    with open(os.path.join(golden_dir, "loader_cases.json")) as f:
        golden = json.load(f)
    assert golden["calls"] == [dict(kw) for kw in LOADER_CALLS]
    files = _loader_case_files(pkg, tmp_path)
    assert sorted(files) == sorted(golden["cases"])
    for label, path in files.items():
        for kw, want in zip(LOADER_CALLS, golden["cases"][label]):
            got = _loader_outcome(pkg.loader.gguf_sd_loader, path, tmp_path, detect_arch=_stand_in_detector, **kw)
            assert got == want, (label, kw)
