"""Import the reference's Python sources VERBATIM (TEST INFRASTRUCTURE).

Where they come from, in this order: ``$GGQ_REFERENCE_DIR``; ``/root/reference`` (the build container);
``oracle/_ref`` -- the git-ignored copy ``oracle/stage_reference.py`` makes at build time so that the files
travel to the GPU box with the snapshot (``/root/reference`` does not exist there).

``gguf`` is not installed here and cannot be (no network).  dequant.py uses three symbols of it:
``GGMLQuantizationType`` (dequant.py:7,288-300), ``GGML_QUANT_SIZES`` (dequant.py:34) and
``quants.dequantize`` (numpy fallback, dequant.py:27, unreachable for the formats in scope).  A stub
module providing the first two is injected as ``sys.modules['gguf']`` and the reference file is then
executed unmodified.  Callers must check ``available()``.
"""
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGED_DIR = os.path.join(_HERE, "_ref")


def _resolve():
    env = os.environ.get("GGQ_REFERENCE_DIR")
    if env:
        return env
    for d in ("/root/reference", STAGED_DIR):
        if os.path.isfile(os.path.join(d, "dequant.py")):
            return d
    return "/root/reference"


REFERENCE_DIR = _resolve()
_cached = None


def available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "dequant.py"))


def source():
    """'live' (/root/reference or $GGQ_REFERENCE_DIR) or 'staged' (oracle/_ref, verified against its manifest)."""
    if os.path.abspath(REFERENCE_DIR) != os.path.abspath(STAGED_DIR):
        return "live"
    from . import stage_reference
    return "staged" if stage_reference.verify(STAGED_DIR) else "staged-UNVERIFIED"


def _gguf_stub():
    # the package's own restatement of ggml's public enum / block sizes
    from ._pkg import load_package
    qt = load_package().qtypes
    mod = types.ModuleType("gguf")
    mod.GGMLQuantizationType = qt.GGMLQuantizationType
    mod.GGML_QUANT_SIZES = dict(qt.GGML_QUANT_SIZES)
    quants = types.ModuleType("gguf.quants")

    def _no_numpy_fallback(*a, **k):
        raise RuntimeError("gguf stub: gguf.quants.dequantize is third-party code and is not available")

    quants.dequantize = _no_numpy_fallback
    mod.quants = quants
    mod.__ggq_stub__ = True
    return mod


def ensure_gguf():
    """Make ``import gguf`` work: the real package if installed, else the stub."""
    try:
        import gguf  # noqa: F401
    except ImportError:
        stub = _gguf_stub()
        sys.modules["gguf"] = stub
        sys.modules["gguf.quants"] = stub.quants
    return sys.modules["gguf"]


def load_reference_dequant():
    """The reference ``dequant`` module, executed from its own source file, unmodified."""
    global _cached
    if _cached is None:
        if not available():
            raise FileNotFoundError(f"{REFERENCE_DIR}/dequant.py not present (neither /root/reference nor the staged oracle/_ref)")
        ensure_gguf()
        spec = importlib.util.spec_from_file_location("ggq_reference_dequant", os.path.join(REFERENCE_DIR, "dequant.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cached = mod
    return _cached


def load_reference_package(name="ggq_refpkg", modules=("dequant", "ops"), setitem=None):
    """Execute the reference's ``dequant.py`` + ``ops.py`` as the package ``name`` (``ops.py`` does
    ``from .dequant import ...``) over the fake ``comfy`` of oracle/fake_comfy.py.  ``setitem(mapping, key, value)``
    registers the modules in ``sys.modules`` (pass ``monkeypatch.setitem`` so a test cleans up after itself; default:
    plain assignment).  Returns {"dequant": module, "ops": module, ...}."""
    from . import fake_comfy
    if not available():
        raise FileNotFoundError(f"{REFERENCE_DIR}: reference sources not present")
    if setitem is None:
        def setitem(mapping, key, value):
            mapping[key] = value
    ensure_gguf()
    for k, v in fake_comfy.build().items():
        setitem(sys.modules, k, v)
    root = types.ModuleType(name)
    root.__path__ = [REFERENCE_DIR]
    setitem(sys.modules, name, root)
    mods = {}
    for m in modules:
        spec = importlib.util.spec_from_file_location(f"{name}.{m}", os.path.join(REFERENCE_DIR, f"{m}.py"))
        mod = importlib.util.module_from_spec(spec)
        setitem(sys.modules, f"{name}.{m}", mod)
        spec.loader.exec_module(mod)
        mods[m] = mod
    return mods
