"""The literal drop-in (Next #6): this package's directory inside ``ComfyUI/custom_nodes/`` next to ComfyUI-GGUF.
A fake ``custom_nodes`` tree is imported the way ComfyUI's ``load_custom_node`` imports it (by path, module name = the path, the
directory order deciding who comes first) -- in BOTH orders -- and afterwards the reference's modules must be running on install()'s
wrappers.  The reference's dequant.py / ops.py are the real files (copied into the temporary tree at test time), over the fake
``comfy`` of oracle/fake_comfy.py.  CPU only: the wrappers hand CPU tensors to the reference's own functions."""
import importlib.util
import logging
import os
import shutil
import sys
import types

import pytest
import torch

from oracle import fake_comfy, reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not reference.available(), reason="reference sources not present")


def _comfy_load_custom_node(module_path):
    """ComfyUI nodes.py load_custom_node, reduced to what matters here: import by path, require NODE_CLASS_MAPPINGS."""
    name = module_path.replace(".", "_x_")
    spec = importlib.util.spec_from_file_location(name, os.path.join(module_path, "__init__.py"))
    module = importlib.util.module_from_spec(spec)
    sys.modules[name] = module
    spec.loader.exec_module(module)
    assert hasattr(module, "NODE_CLASS_MAPPINGS"), f"ComfyUI would skip {module_path}: no NODE_CLASS_MAPPINGS"
    return module


@pytest.fixture
def custom_nodes(tmp_path, monkeypatch):
    reference.ensure_gguf()
    for k, v in fake_comfy.build().items():
        monkeypatch.setitem(sys.modules, k, v)
    monkeypatch.setitem(sys.modules, "folder_paths", types.ModuleType("folder_paths"))      # what ComfyUI has imported before any custom node
    monkeypatch.delenv("GGQ_AUTO_INSTALL", raising=False)
    root = tmp_path / "ComfyUI" / "custom_nodes"
    ref = root / "ComfyUI-GGUF"
    ref.mkdir(parents=True)
    for f in ("dequant.py", "ops.py"):
        shutil.copy(os.path.join(reference.REFERENCE_DIR, f), ref / f)
    # the reference's __init__.py imports its nodes (which need all of ComfyUI); what reaches this path is `from .ops import GGMLOps` (nodes.py:15)
    (ref / "__init__.py").write_text("from .ops import GGMLOps\nNODE_CLASS_MAPPINGS = {'UnetLoaderGGUF': object}\n")
    os.symlink(os.path.join(ROOT, "comfyui-gguf_amd"), root / "comfyui-gguf_amd")
    before, meta = set(sys.modules), list(sys.meta_path)
    yield str(ref), str(root / "comfyui-gguf_amd")
    sys.meta_path[:] = meta
    for k in set(sys.modules) - before:
        del sys.modules[k]


def _assert_installed(ref_mod, amd_mod):
    rd, ro = sys.modules[ref_mod.__name__ + ".dequant"], sys.modules[ref_mod.__name__ + ".ops"]
    assert amd_mod.NODE_CLASS_MAPPINGS == {} and amd_mod.autoinstall._state["installed"] == ref_mod.__name__
    assert hasattr(rd.dequantize_tensor, "__wrapped__") and hasattr(rd.dequantize, "__wrapped__")
    assert ro.dequantize_tensor is rd.dequantize_tensor                       # the name ops.py bound at import time (ops.py:9)
    assert not any(type(f).__name__ == "_AfterOpsImport" for f in sys.meta_path)   # the hook took itself out
    # CPU-resident weights still run the reference's own code through the wrappers
    Q = amd_mod.qtypes.Q
    blocks = amd_mod.synth.make_blocks(Q.Q4_K, 8 * 2, seed=3)
    w = ro.GGMLTensor(torch.from_numpy(blocks.reshape(-1).copy()), tensor_type=Q.Q4_K, tensor_shape=torch.Size((8, 512)))
    assert torch.equal(rd.dequantize_tensor(w, torch.float32), rd.dequantize_tensor.__wrapped__(w, torch.float32))
    amd_mod.install.uninstall(rd)
    assert not hasattr(rd.dequantize_tensor, "__wrapped__")


def test_reference_first_then_this_package(custom_nodes, caplog):
    ref_dir, amd_dir = custom_nodes
    with caplog.at_level(logging.INFO, logger="comfyui-gguf_amd"):
        ref_mod = _comfy_load_custom_node(ref_dir)                # "ComfyUI-GGUF" sorts before "comfyui-gguf_amd": ComfyUI's usual order
        amd_mod = _comfy_load_custom_node(amd_dir)
    _assert_installed(ref_mod, amd_mod)
    assert sum("HIP dequant path installed over" in r.getMessage() for r in caplog.records) == 1      # one line, saying what was patched
    # the default since round 5: the no-VRAM opt-ins are ON (install.DEFAULT_FAST), and the one log line says so and names the way back
    assert "dequantize_tensor" in caplog.text and "fused_small_m=True" in caplog.text and "fused_mfma=256" in caplog.text and "GGQ_EXACT=1" in caplog.text


def test_this_package_first_then_reference(custom_nodes):
    ref_dir, amd_dir = custom_nodes
    amd_mod = _comfy_load_custom_node(amd_dir)
    assert amd_mod.autoinstall._state == {"armed": True, "installed": None, "gave_up": False}
    assert any(type(f).__name__ == "_AfterOpsImport" for f in sys.meta_path)
    import json                                                   # unrelated imports pass through the armed hook untouched
    assert json.loads("1") == 1
    ref_mod = _comfy_load_custom_node(ref_dir)
    _assert_installed(ref_mod, amd_mod)


def test_not_armed_outside_comfyui(custom_nodes, monkeypatch):
    """Imported by anything else (tests, bench.py, a script), the package patches nothing and installs no import hook."""
    ref_dir, amd_dir = custom_nodes
    monkeypatch.delitem(sys.modules, "folder_paths")
    amd_mod = _comfy_load_custom_node(amd_dir)
    ref_mod = _comfy_load_custom_node(ref_dir)
    assert "autoinstall" not in amd_mod.__dict__ and not any(type(f).__name__ == "_AfterOpsImport" for f in sys.meta_path)
    assert not hasattr(sys.modules[ref_mod.__name__ + ".dequant"].dequantize_tensor, "__wrapped__")


def test_ggq_exact_restores_the_bit_exact_default(custom_nodes, monkeypatch, caplog):
    """GGQ_EXACT=1: nothing above dequantize / dequantize_tensor is patched -- every linear runs unpack + F.linear, as in rounds 1-4."""
    ref_dir, amd_dir = custom_nodes
    monkeypatch.setenv("GGQ_EXACT", "1")
    ref_mod = _comfy_load_custom_node(ref_dir)
    with caplog.at_level(logging.INFO, logger="comfyui-gguf_amd"):
        amd_mod = _comfy_load_custom_node(amd_dir)
    ro, rd = sys.modules[ref_mod.__name__ + ".ops"], sys.modules[ref_mod.__name__ + ".dequant"]
    assert hasattr(rd.dequantize_tensor, "__wrapped__")
    assert not hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__")
    assert not hasattr(ro.GGMLOps.Embedding.forward_ggml_cast_weights, "__wrapped__")
    assert not hasattr(ro.GGMLLayer.ggml_save_to_state_dict, "__wrapped__")
    assert "exact=True" in caplog.text and "fused_small_m" not in caplog.text
    amd_mod.install.uninstall(rd)


def test_unrelated_ops_modules_pass_through_the_armed_hook(custom_nodes, tmp_path, monkeypatch):
    """``torchvision.ops``, ``comfy.ops`` ... -- any ``*.ops`` whose package has no ``dequant`` sibling -- must not be re-resolved
    through the other finders nor get its loader touched while the hook waits for ComfyUI-GGUF."""
    ref_dir, amd_dir = custom_nodes
    amd_mod = _comfy_load_custom_node(amd_dir)
    other = tmp_path / "otherpkg"
    other.mkdir()
    (other / "__init__.py").write_text("")
    (other / "ops.py").write_text("GGMLOps = GGMLTensor = GGMLLayer = object\n")        # even with the right names: no dequant.py next to it
    monkeypatch.syspath_prepend(str(tmp_path))
    import otherpkg.ops
    assert type(otherpkg.ops.__spec__.loader).__name__ != "_HookedLoader"
    assert amd_mod.autoinstall._state["installed"] is None and any(type(f).__name__ == "_AfterOpsImport" for f in sys.meta_path)
    ref_mod = _comfy_load_custom_node(ref_dir)                     # the real one still gets picked up afterwards
    _assert_installed(ref_mod, amd_mod)


def test_a_failing_install_is_logged_once_and_the_hook_gives_up(custom_nodes, monkeypatch, caplog):
    """install() raising must not leave the finder at sys.meta_path[0] retrying (and logging) on every later ``*.ops`` import."""
    ref_dir, amd_dir = custom_nodes
    amd_mod = _comfy_load_custom_node(amd_dir)
    calls = []

    def boom(*a, **k):
        calls.append(1)
        raise RuntimeError("no such device")
    monkeypatch.setattr(amd_mod.install, "install", boom)
    with caplog.at_level(logging.ERROR, logger="comfyui-gguf_amd"):
        ref_mod = _comfy_load_custom_node(ref_dir)
    rd = sys.modules[ref_mod.__name__ + ".dequant"]
    assert calls == [1] and amd_mod.autoinstall._state["gave_up"] and amd_mod.autoinstall._state["installed"] is None
    assert not hasattr(rd.dequantize_tensor, "__wrapped__")                                  # the reference keeps its own torch path
    assert not any(type(f).__name__ == "_AfterOpsImport" for f in sys.meta_path)             # gone, not waiting for the next *.ops
    assert sum("could not install over" in r.getMessage() for r in caplog.records) == 1
    amd_mod.autoinstall._install_over(sys.modules[ref_mod.__name__ + ".ops"])               # a later attempt is a no-op: no second traceback
    assert calls == [1]
    amd_mod.autoinstall._state.update(armed=False, installed=None, gave_up=False)


def test_ggq_fast_switch_turns_on_the_no_vram_opt_ins(custom_nodes, monkeypatch, caplog):
    """GGQ_FAST=1: fused_small_m + fused_mfma + gather_embedding in one switch; an option's own variable still wins."""
    ref_dir, amd_dir = custom_nodes
    monkeypatch.setenv("GGQ_FAST", "1")
    monkeypatch.setenv("GGQ_GATHER_EMBEDDING", "0")
    ref_mod = _comfy_load_custom_node(ref_dir)
    with caplog.at_level(logging.INFO, logger="comfyui-gguf_amd"):
        amd_mod = _comfy_load_custom_node(amd_dir)
    ro, rd = sys.modules[ref_mod.__name__ + ".ops"], sys.modules[ref_mod.__name__ + ".dequant"]
    assert hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__")
    assert not hasattr(ro.GGMLOps.Embedding.forward_ggml_cast_weights, "__wrapped__")             # GGQ_GATHER_EMBEDDING=0 wins
    assert "fused_small_m=True" in caplog.text and "fused_mfma=256" in caplog.text and "gather_embedding" not in caplog.text
    amd_mod.install.uninstall(rd)
    assert not hasattr(ro.GGMLOps.Linear.forward_ggml_cast_weights, "__wrapped__")


@pytest.mark.parametrize("with_loader", [True, False], ids=["loader.py present", "checkout without loader.py"])
def test_ggq_native_reader_under_the_drop_in(custom_nodes, monkeypatch, caplog, with_loader):
    """GGQ_NATIVE_READER=1 (round 6): the hook fires right after <package>.ops has executed -- before nodes.py reaches `from .loader import ...` -- so it imports
    the loader module itself and rebinds the name `gguf` inside it to the proxy whose GGUFReader is the native adapter; a checkout without loader.py keeps working
    (one warning, everything else installed)."""
    ref_dir, amd_dir = custom_nodes
    if with_loader:
        shutil.copy(os.path.join(reference.REFERENCE_DIR, "loader.py"), os.path.join(ref_dir, "loader.py"))
    monkeypatch.setenv("GGQ_NATIVE_READER", "1")
    amd_mod = _comfy_load_custom_node(amd_dir)                               # this package first: the hook is armed, the reference arrives later
    with caplog.at_level(logging.INFO, logger="comfyui-gguf_amd"):
        ref_mod = _comfy_load_custom_node(ref_dir)
    rd = sys.modules[ref_mod.__name__ + ".dequant"]
    assert amd_mod.autoinstall._state["installed"] == ref_mod.__name__ and hasattr(rd.dequantize_tensor, "__wrapped__")
    if with_loader:
        ldr = sys.modules[ref_mod.__name__ + ".loader"]
        assert type(ldr.gguf).__name__ == "GGUFModuleProxy" and issubclass(ldr.gguf.GGUFReader, amd_mod.gguf_adapter.GGUFReaderAdapter)
        assert ldr.gguf.GGMLQuantizationType is sys.modules["gguf"].GGMLQuantizationType and "native_reader=True" in caplog.text
        amd_mod.install.uninstall(rd)
        assert ldr.gguf is sys.modules["gguf"]
    else:
        assert "GGUFReader stays" in caplog.text and "native_reader=True" not in amd_mod.install.describe(rd)
        amd_mod.install.uninstall(rd)
