"""Helpers that drive the REFERENCE's own classes (ops.py / dequant.py executed verbatim, oracle/reference.py) with
``install()`` applied or not.  Shared by tests/test_gpu_reference.py (device = the MI355X: the wiring the north star's
"Unet Loader (GGUF) works unchanged" rests on) and by a CPU dry run in tests/test_host.py (device = cpu: install() falls
through to the reference, so the comparisons are trivially true -- the dry run only proves the harness itself is sound
before GPU minutes are spent on it)."""
import numpy as np
import torch

import oracle

KIND = {torch.float16: "f16", torch.bfloat16: "bf16", torch.float32: "f32"}
DEQUANT_DTYPES = (None, "target", torch.float32, torch.bfloat16)
DTYPES = (torch.float16, torch.bfloat16, torch.float32)


def bits(t):
    """Bit patterns with every NaN canonicalised (payloads are not part of the contract)."""
    t = t.detach()
    if type(t) is not torch.Tensor:
        t = t.as_subclass(torch.Tensor)
    t = t.cpu().contiguous()
    view = {2: torch.int16, 4: torch.int32}[t.element_size()]
    b = t.view(view).clone()
    b[torch.isnan(t)] = 0x7FFF if t.element_size() == 2 else 0x7FFFFFFF
    return b


def same_bits(a, b):
    return a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape) and bool(torch.equal(bits(a), bits(b)))


def oracle_tensor(q, packed, dtype, dequant_dtype, shape):
    """What dequantize_tensor(tensor, dtype, dequant_dtype) must return, from the C oracle."""
    compute = dtype if dequant_dtype == "target" else dequant_dtype
    ck = "f16" if compute is None else KIND[compute]
    out = oracle.dequant_tensor(q, packed, ck, KIND[dtype])
    if KIND[dtype] == "f32":
        t = torch.from_numpy(np.ascontiguousarray(out).view(np.float32).copy())
    else:
        t = torch.from_numpy(np.ascontiguousarray(out).view(np.int16).copy()).view(dtype)
    return t.reshape(shape)


def ggml(ro, array, qtype, shape, device, patches=None, rows=None):
    """A reference GGMLTensor the way the loader makes one (loader.py:104-124: torch.from_numpy of the file's bytes, CPU),
    then moved with the reference's own ``GGMLTensor.to`` (ops.py:57-62) -- the only way a GGMLTensor reaches the GPU."""
    data = torch.from_numpy(np.ascontiguousarray(array).copy())
    if rows:
        data = data.reshape(rows, -1)                       # gguf-py hands quantized tensors out as (rows, bytes per row)
    t = ro.GGMLTensor(data, tensor_type=qtype, tensor_shape=torch.Size(shape), patches=list(patches or []))
    return t.to(device) if torch.device(device).type != "cpu" else t


def param(t):
    return torch.nn.Parameter(t, requires_grad=False)


def make_linear(ro, pkg, q, out_features, in_features, device, seed=0, bias=True, mode="signed", patches=None, dequant_dtype=None):
    Q = pkg.qtypes.Q
    packed = pkg.synth.make_tensor_bytes(q, (out_features, in_features), seed=seed, mode=mode)
    lin = ro.GGMLOps.Linear(in_features, out_features)
    lin.weight = param(ggml(ro, packed, q, (out_features, in_features), device, patches=patches, rows=out_features))
    if bias:
        g = torch.Generator().manual_seed(seed + 1)
        b = torch.randn(out_features, generator=g, dtype=torch.float32)
        lin.bias = param(ggml(ro, b.numpy(), Q.F32, (out_features,), device))
    lin.dequant_dtype = dequant_dtype
    return lin, packed


def make_embedding(ro, pkg, q, n_rows, cols, device, seed=0, dequant_dtype=None):
    packed = pkg.synth.make_tensor_bytes(q, (n_rows, cols), seed=seed, mode="signed")
    emb = ro.GGMLOps.Embedding(n_rows, cols, device="meta")
    emb.weight = param(ggml(ro, packed, q, (n_rows, cols), device, rows=n_rows))
    emb.dequant_dtype = dequant_dtype
    return emb, packed


def make_conv2d(ro, pkg, q, out_c, in_c, kh, kw, device, seed=0):
    Q = pkg.qtypes.Q
    packed = pkg.synth.make_tensor_bytes(q, (out_c, in_c * kh * kw), seed=seed, mode="signed")
    conv = ro.GGMLOps.Conv2d(in_c, out_c, (kh, kw), padding=1, device="meta")
    conv.weight = param(ggml(ro, packed, q, (out_c, in_c, kh, kw), device, rows=out_c))
    g = torch.Generator().manual_seed(seed + 1)
    conv.bias = param(ggml(ro, torch.randn(out_c, generator=g).numpy(), Q.F32, (out_c,), device))
    return conv, packed


def lora_patch(shape, seed, strength=0.5):
    """tensor.patches as ComfyUI's model patcher leaves it: [(patch list, key)]; here one "diff" patch (oracle/fake_comfy.py)."""
    g = torch.Generator().manual_seed(seed)
    return [([(strength, torch.randn(shape, generator=g, dtype=torch.float32) * 0.01)], "diffusion_model.key")]


class Installed:
    """``with Installed(pkg, mods, **options):`` -- install() over the reference modules for the duration of the block.  ``exact=True`` unless the
    test says ``fast`` / ``exact`` itself: these tests compare with the reference's own output BIT FOR BIT, which is what ``exact`` promises (the
    round-5 default additionally fuses small linears: tolerance parity, tested where it is asked for by name)."""

    def __init__(self, pkg, mods, **options):
        if "fast" not in options:
            options.setdefault("exact", True)
        self.pkg, self.mods, self.options = pkg, mods, options

    def __enter__(self):
        self.pkg.install.install(self.mods["dequant"], self.mods["ops"], **self.options)
        return self

    def __exit__(self, *exc):
        self.pkg.install.uninstall(self.mods["dequant"])
        return False

    @property
    def cache(self):
        return self.pkg.install.dense_cache(self.mods["dequant"])


class LaunchCounter:
    """Counts calls of the C entry point ggq_dequant made by the host mirror (dequant._ggq_dequant: the one binding every
    dequantize_tensor / dequantize / block-function call goes through): a GPU test that compares "installed" with "reference" must
    also prove the installed side really ran the HIP path."""

    def __init__(self, pkg, monkeypatch):
        self.n = 0
        real = pkg.dequant._ggq_dequant or pkg.dequant._bind()

        def counted(*a):
            self.n += 1
            return real(*a)

        monkeypatch.setattr(pkg.dequant, "_ggq_dequant", counted)
