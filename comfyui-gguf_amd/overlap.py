"""Opt-in: unpack layer i+1 while layer i computes -- and, in low-VRAM mode, move its packed bytes host->device meanwhile.

The reference's ``GGMLLayer.cast_bias_weight`` (ops.py:194-211) does, per layer and per forward, on ONE stream:

    weight = s.weight.to(device)          # low-VRAM mode: a PCIe copy of the packed bytes (ops.py:209)
    weight = dequantize_tensor(weight)    # HBM-bound unpack (ops.py:177)
    F.linear(input, weight, bias)         # MFMA-bound GEMM (ops.py:244)

Three different resources, used strictly one after the other.  ``LayerPrefetcher`` learns the order in which layers are called
(module A is followed by module B -- it repeats every denoising step) and, when A is called, enqueues B's copy + unpack on a
side streams (include/ggq.h ``ggq_overlap_*``): the unpack into one of two persistent dense scratch buffers, event-ordered after
everything the main stream holds (so the buffer's previous consumer is done); in low-VRAM mode the host->device copies on a stream of
their own, TWO layers ahead into three staging buffers, gated only by the unpack that last read the staging buffer -- the PCIe link
never waits for the GEMMs.  When B is called its dense weight is already there (or on its way: the
main stream waits on the event, never the host).  The values are the very same kernels' output: results stay bit-identical.

Measured (tools/flux_forward_emulation.py, FLUX.1-dev, 4608 tokens, bf16; profiles/r02_flux_forward_emulation_overlap.json):
low-VRAM mode 205 -> 166 ms per step; with the packed weights already in HBM the same mechanism LOSES (74.8 -> 77.1 ms), so by
default (``overlap=True``) only CPU-resident weights are prefetched; ``overlap="all"`` prefetches resident ones too.

What it changes for the caller, and why it is opt-in (``install(..., overlap=True)`` / ``GGQ_OVERLAP=1``):
  * the dense weight handed to ``F.linear`` is a view into a scratch buffer that is rewritten two layers later.  The reference's
    layers consume the weight immediately (ops.py:242-271), so nothing notices -- but code that stashes the result of
    ``cast_bias_weight`` would;
  * low-VRAM mode keeps a PINNED host copy of the packed weight of every LIVE layer module it has seen (the async copy needs page-locked
    memory): host RAM of the size of the packed models currently loaded (6.8 GB for FLUX.1-dev Q4_K_M).  The copies are tied to the
    modules by weak references: a model that is unloaded gives its pinned memory back (``stats()["pinned_host_bytes"]``);
  * 2 x the largest dense weight of scratch per device (2 x 132 MB for FLUX.1-dev in bf16), plus 3 x the largest packed weight of
    staging in low-VRAM mode (3 x 37 MB).
LoRA-patched weights (patched in place, ops.py:183-190), non-quantized weights, dtypes the kernels do not emit, tracing under
torch.compile and stream capture all take the reference's path untouched; a mispredicted order only costs the prefetch.
"""
import ctypes
import sys
import threading
import weakref

import torch

from . import _native
from . import dequant as _dq

N_SLOTS = 2               # dense scratch slots: the weight being consumed and the one being unpacked
N_STAGING = 3             # packed staging slots (low-VRAM mode): being unpacked, copied for the next layer, copied for the one after


class _Slot:
    __slots__ = ("dense", "owner")

    def __init__(self):
        self.dense = None         # uint8 scratch on the device, grown to the largest dense weight seen
        self.owner = None         # weak reference to the module whose prefetched weight currently lives here


class _ModuleState:
    """Everything the prefetcher remembers about ONE layer module.  Lives in a WeakKeyDictionary keyed by the module itself, so it
    dies with the module (a model that is unloaded takes its pinned host copies, its learnt successor and its pending prefetch
    with it) and a recycled ``id()`` can never inherit it.  The pinned buffer is not freed on the spot: a raw-stream copy that
    torch's host allocator knows nothing about may still be reading it, so it is handed to the prefetcher's ``_retired`` list,
    which is emptied behind a device synchronisation."""
    __slots__ = ("next", "last", "pinned", "pending", "_retired", "__weakref__")

    def __init__(self, retired):
        self.next = None          # weak reference to the module that was called after this one last time
        self.last = None          # (dtype, device index) of its last call
        self.pinned = None        # (weak reference to the CPU weight object, its version, pinned uint8 copy)
        self.pending = None       # (weight ref, version, dtype, compute dtype, device index, slot index, dense view)
        self._retired = retired

    def retire_pinned(self):
        if self.pinned is not None:
            self._retired.append(self.pinned[2])
            self.pinned = None

    def __del__(self):
        try:
            self.retire_pinned()
        except Exception:
            pass


class _Staging:
    __slots__ = ("packed", "owner")

    def __init__(self):
        self.packed = None        # uint8 staging for packed bytes copied from the host
        self.owner = None         # (weak reference to the CPU weight, its version) whose packed bytes were last copied here


class _Device:
    def __init__(self, index):
        self.index = index
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(index):
            _native.check(_native.lib().ggq_overlap_create(max(N_SLOTS, N_STAGING), ctypes.byref(self.handle)), "ggq_overlap_create")
        self.slots = [_Slot() for _ in range(N_SLOTS)]
        self.staging = [_Staging() for _ in range(N_STAGING)]
        self.turn = 0
        self.staging_turn = 0

    def close(self):
        if self.handle:
            _native.lib().ggq_overlap_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class LayerPrefetcher:
    """``weight = prefetcher.weight_for(module, dtype, device)`` returns what ``module.get_weight(module.weight.to(device), dtype)``
    returns (same values), from the side stream's buffer when the call was predicted, else computed on the spot; either way it
    then schedules the predicted NEXT module's weight.  One instance serves one thread of forward calls (ComfyUI runs a model on
    one thread); calls from other threads take the reference's path."""

    def __init__(self, resident=False):
        # resident=False (default): only CPU-resident packed weights (low-VRAM mode) are prefetched.  For weights that already
        # live in HBM the side stream LOSES on MI355X (emulated FLUX.1-dev step 74.8 -> 77.1 ms at 4608 tokens, 21.4 -> 23.3 ms at
        # 512, profiles/r02_flux_forward_emulation_overlap.json): the unpack kernels are 6-18 us, a cross-queue event dependency
        # costs about as much twice per layer, and the unpack's streaming traffic disturbs the GEMM's cache residency.
        self.resident = bool(resident)
        self._devices = {}
        self._state = weakref.WeakKeyDictionary()     # module -> _ModuleState (weak keys: nothing here keeps a module, or what it pinned, alive)
        self._retired = []         # pinned host buffers whose owner died or was replaced; freed behind a device sync (_drain_retired)
        self._prev = None          # weakref to the module of the previous call
        self._owner_thread = None
        self.hits = self.misses = self.mispredicted = self.bypassed = 0

    # ---- what qualifies
    def eligible(self, module, dtype, device):
        w = getattr(module, "weight", None)
        if w is None or getattr(w, "patches", None) or dtype not in _dq._OUT_CODE:
            return False
        if not self.resident and w.device.type != "cpu":
            return False
        qtype = getattr(w, "tensor_type", None)
        if qtype not in _dq._HIP_TABLE and _dq._qtype_key(qtype) not in _dq._HIP_TABLE:
            return False
        if device is None or torch.device(device).type != "cuda":
            return False
        tid = threading.get_ident()
        if self._owner_thread is None:
            self._owner_thread = tid
        if tid != self._owner_thread or _dq._is_compiling():
            return False
        if torch.cuda.is_current_stream_capturing():
            # under HIP-graph capture the side streams would be pulled into the capture without ever joining it back, and the
            # main stream would wait on events recorded outside of it: the capture takes the reference's single-stream path
            return False
        index = torch.device(device).index
        index = torch.cuda.current_device() if index is None else index
        return bool(_dq._DEVICE_OK.get(index) or _dq._device_served(index))

    def break_chain(self):
        self._prev = None
        self.bypassed += 1

    # ---- internals
    def _st(self, module):
        st = self._state.get(module)
        if st is None:
            st = self._state[module] = _ModuleState(self._retired)
        return st

    def _drain_retired(self):
        """Free the pinned buffers of dead / replaced weights -- after everything in flight on the devices (the raw copy streams
        included) is done with them.  Runs once per model unload, not per call."""
        if self._retired:
            for index in list(self._devices):
                torch.cuda.synchronize(index)
            del self._retired[:]

    def _dev(self, index):
        d = self._devices.get(index)
        if d is None:
            d = self._devices[index] = _Device(index)
        return d

    @staticmethod
    def _compute_dtype(module, dtype):
        dd = getattr(module, "dequant_dtype", None)
        return dtype if dd == "target" else dd

    def _host_bytes(self, module, w):
        """Pinned copy of a CPU-resident packed weight (made once per weight object and version)."""
        st = self._st(module)
        ent = st.pinned
        if ent is not None and ent[0]() is w and ent[1] == w._version:
            return ent[2]
        st.retire_pinned()                                    # another weight object / version: the old copy may still be in flight
        with _dq._NoTorchFunction():
            flat = _dq._as_bytes(w, align=False)
            pinned = torch.empty(flat.numel(), dtype=torch.uint8, pin_memory=True)
            pinned.copy_(flat)
        st.pinned = (weakref.ref(w), w._version, pinned)
        return pinned

    def _stage_copy(self, module, index):
        """Low-VRAM mode: enqueue the host -> device copy of module's packed weight on the COPY stream (not ordered against the main
        stream: it runs as far ahead as there are staging slots).  Returns the staging slot index, or None."""
        w = getattr(module, "weight", None)
        if w is None or getattr(w, "patches", None):
            return None
        with _dq._NoTorchFunction():
            if w.is_cuda:
                return None
            host = self._host_bytes(module, w)
        dev = self._dev(index)
        for si, st in enumerate(dev.staging):
            if st.owner is not None and st.owner[0]() is w and st.owner[1] == w._version:
                return si                                     # already on its way (scheduled two layers ahead); identity, not id(): ids are recycled
        si = dev.staging_turn
        st = dev.staging[si]
        nbytes = host.numel()
        if st.packed is None or st.packed.numel() < nbytes:
            torch.cuda.synchronize(index)                     # growing a buffer frees the old one: nothing may still be using it
            with torch.cuda.device(index):
                st.packed = torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{index}")
        with torch.cuda.device(index):
            rc = _native.lib().ggq_overlap_copy(dev.handle, si, host.data_ptr(), st.packed.data_ptr(), nbytes)
        _native.check(rc, "ggq_overlap_copy")
        st.owner = (weakref.ref(w), w._version)
        dev.staging_turn = (si + 1) % N_STAGING
        return si

    def _schedule(self, module, dtype, index, main_stream, avoid_slot=None):
        """Enqueue module's unpack (after its copy, in low-VRAM mode) on the unpack stream of device `index`.  Returns False if it
        cannot be prefetched.  ``avoid_slot``: the dense slot whose weight is being handed out by THIS call -- its consumer is not
        enqueued yet, so the event recorded now would not cover it (in the learnt order the rotation never picks it; after a reordering
        it can)."""
        w = module.weight
        qtype = getattr(w, "tensor_type", None)
        ent = _dq._HIP_TABLE.get(qtype) or _dq._HIP_TABLE.get(_dq._qtype_key(qtype))
        if ent is None or getattr(w, "patches", None):
            return False
        compute = self._compute_dtype(module, dtype)
        try:
            compute_code = _dq._check_compute(compute)
        except _dq.GGQUnsupported:
            return False
        qid, block_size, type_size = ent
        dev = self._dev(index)
        slot_index = dev.turn
        if slot_index == avoid_slot:
            slot_index = (slot_index + 1) % N_SLOTS
        slot = dev.slots[slot_index]
        with _dq._NoTorchFunction():
            on_host = not w.is_cuda
            if on_host:
                staging_index = self._stage_copy(module, index)
                if staging_index is None:
                    return False
                data = dev.staging[staging_index].packed
                nbytes = self._st(module).pinned[2].numel()
            else:
                staging_index = -1
                if w.device.index != index:
                    return False
                data = w
                if data.dtype is not torch.uint8 or not data.is_contiguous() or data.data_ptr() & 15:
                    return False                          # views / misaligned starts: the ordinary path copies them first
                nbytes = data.numel()
            n_blocks = nbytes // type_size
            shape = tuple(getattr(w, "tensor_shape", ()))
            n = n_blocks * block_size
            numel = 1
            for s in shape:
                numel *= int(s)
            if numel != n or n == 0:
                return False
            dense_bytes = n * (4 if dtype is torch.float32 else 2)
            if slot.dense is None or slot.dense.numel() < dense_bytes:
                torch.cuda.synchronize(index)             # growing a buffer frees the old one: nothing on any stream may still be using it
                with torch.cuda.device(index):
                    slot.dense = torch.empty(dense_bytes, dtype=torch.uint8, device=f"cuda:{index}")
            dense = slot.dense[:dense_bytes].view(dtype).view(shape)
            prev_owner = slot.owner() if slot.owner is not None else None
            if prev_owner is not None and prev_owner is not module:
                pst = self._state.get(prev_owner)
                if pst is not None and pst.pending is not None and pst.pending[5] == slot_index:
                    pst.pending = None                    # whatever lived in this slot is about to be overwritten
            with torch.cuda.device(index):
                rc = _native.lib().ggq_overlap_prefetch(dev.handle, slot_index, staging_index, qid, data.data_ptr(), n_blocks, dense.data_ptr(),
                                                        compute_code, _dq._OUT_CODE[dtype], main_stream)
        _native.check(rc, "ggq_overlap_prefetch")
        slot.owner = weakref.ref(module)
        dev.turn = (slot_index + 1) % N_SLOTS
        self._st(module).pending = (weakref.ref(w), w._version, dtype, compute, index, slot_index, dense)
        return True

    # ---- the call
    def weight_for(self, module, dtype, device, compute_now):
        """``compute_now()`` produces the weight the reference's way (used on a miss).  Call only when ``eligible()``."""
        index = torch.device(device).index
        index = _dq._cur_device() if index is None else index
        main_stream = _dq._raw_stream(index)
        if self._retired:
            self._drain_retired()
        st = self._st(module)
        w = module.weight
        pend, st.pending = st.pending, None
        dense, used_slot = None, None
        if pend is not None:
            wref, version, p_dtype, p_compute, p_index, slot_index, p_dense = pend
            dev = self._devices[p_index]
            if (wref() is w and version == w._version and p_dtype is dtype and p_index == index and p_compute == self._compute_dtype(module, dtype)):
                _native.check(_native.lib().ggq_overlap_wait(dev.handle, slot_index, main_stream), "ggq_overlap_wait")
                dense, used_slot = p_dense, slot_index
                self.hits += 1
            else:
                self.mispredicted += 1
            owner = dev.slots[slot_index].owner
            if owner is not None and owner() is module:
                dev.slots[slot_index].owner = None
        if dense is None:
            dense = compute_now()
            self.misses += 1
        # learn the order, then look one layer ahead
        prev = self._prev() if self._prev is not None else None
        if prev is not None:
            self._st(prev).next = weakref.ref(module)
        self._prev = weakref.ref(module)
        st.last = (dtype, index)
        nxt = st.next() if st.next is not None else None
        if nxt is not None:
            nst = self._state.get(nxt)
            if nst is not None:
                if nst.pending is None and nst.last is not None and nst.last[1] == index:
                    self._schedule(nxt, nst.last[0], index, main_stream, avoid_slot=used_slot)
                # low-VRAM mode: the layer after the next one starts its PCIe copy now, so the link never idles behind the unpack
                nxt2 = nst.next() if nst.next is not None else None
                if nxt2 is not None and nxt2 is not module:
                    n2st = self._state.get(nxt2)
                    if n2st is not None and n2st.last is not None:
                        self._stage_copy(nxt2, index)
        return dense

    def stats(self):
        return {"hits": self.hits, "misses": self.misses, "mispredicted": self.mispredicted, "bypassed": self.bypassed,
                "modules_tracked": len(self._state),
                "pinned_host_bytes": sum(st.pinned[2].numel() for st in list(self._state.values()) if st.pinned is not None),
                "retired_host_bytes": sum(t.numel() for t in self._retired),
                "scratch_bytes": self.scratch_bytes()}

    def scratch_bytes(self, index=None):
        """Device memory this prefetcher holds outside of what the reference's VRAM estimate knows about (dense scratch slots +
        packed staging slots), in bytes; ``index``: one device, default all.  INTEGRATION.md section 5 shows where a maintainer adds
        it to the ``temp.weight`` reservation of ``ggml_save_to_state_dict`` (reference ops.py:153-158)."""
        devs = self._devices.values() if index is None else [d for i, d in self._devices.items() if i == index]
        return sum(sum(s.dense.numel() for s in d.slots if s.dense is not None) + sum(s.packed.numel() for s in d.staging if s.packed is not None)
                   for d in devs)

    def close(self):
        for index in list(self._devices):
            torch.cuda.synchronize(index)
        for d in self._devices.values():
            d.close()
        self._devices.clear()
        for st in list(self._state.values()):
            st.pinned = None                                   # everything was synchronised above: nothing reads the pinned copies any more
        self._state = weakref.WeakKeyDictionary()
        del self._retired[:]
        self._prev = None


def attach(layer_cls, prefetcher=None, resident=False):
    """Wrap ``layer_cls.cast_bias_weight`` (the reference's ``GGMLLayer``, ops.py:194-211, or this package's stand-in).
    Returns ((owner, name, original) for uninstall, the prefetcher)."""
    pf = prefetcher or LayerPrefetcher(resident=resident)
    original = layer_cls.cast_bias_weight
    ops_module = sys.modules.get(layer_cls.__module__)
    comfy = getattr(ops_module, "comfy", None)            # the reference's `import comfy.ops` / `comfy.model_management`

    def cast_bias_weight(s, input=None, dtype=None, device=None, bias_dtype=None):
        if input is not None:                              # ops.py:195-201, verbatim semantics
            if dtype is None:
                dtype = getattr(input, "dtype", torch.float32)
            if bias_dtype is None:
                bias_dtype = dtype
            if device is None:
                device = input.device
        if not pf.eligible(s, dtype, device):
            pf.break_chain()
            return original(s, input, dtype, device, bias_dtype)
        bias = None
        if comfy is not None:
            non_blocking = comfy.model_management.device_supports_non_blocking(device)
            if s.bias is not None:
                bias = s.get_weight(s.bias.to(device), dtype)
                bias = comfy.ops.cast_to(bias, bias_dtype, device, non_blocking=non_blocking, copy=False)
        elif s.bias is not None:
            bias = s.get_weight(s.bias.to(device), dtype).to(device=device, dtype=bias_dtype)
        weight = pf.weight_for(s, dtype, device, lambda: s.get_weight(s.weight.to(device), dtype).to(device=device, dtype=dtype))
        return weight, bias

    cast_bias_weight.__wrapped__ = original
    layer_cls.cast_bias_weight = cast_bias_weight
    return (layer_cls, "cast_bias_weight", original), pf
