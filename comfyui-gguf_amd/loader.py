"""The input side of the dequant path: GGUF file -> state dict of ``GGMLTensor`` (reference loader.py:16-141),
without the third-party ``gguf`` package, plus what the reference cannot do: stream the packed weights
straight into HBM in one pass (``device=...``) so the dequant kernels find them resident.

Mirrored surface (same names, argument meaning, results and errors):
    get_orig_shape(reader, tensor_name)                     loader.py:16-24
    get_field(reader, field_name, field_type)               loader.py:26-37
    get_list_field(reader, field_name, field_type)          loader.py:39-49
    gguf_sd_loader(path, handle_prefix, return_arch, is_text_model)     loader.py:51-141
``reader`` is a :class:`gguf_file.GGUFFile` instead of a ``gguf.GGUFReader``.

Not mirrored (the reference's control plane, SURVEY.md section 2 -- left to the reference's own code,
which keeps working on top of this module's state dicts): key remapping for text encoders, tokenizer
reconstruction, mmproj discovery (loader.py:143-406), and architecture *detection* for files that carry
no ``general.architecture`` (tools/convert.py ``detect_arch``) -- pass it in as ``detect_arch=`` if needed.
"""
import logging

import torch

from .dequant import dequantize_tensor, is_quantized
from .gguf_file import ARRAY, INT32, STRING, GGUFFile
from .ops import GGMLTensor
from .qtypes import GGMLQuantizationType as Q

IMG_ARCH_LIST = {"flux", "sd1", "sdxl", "sd3", "aura", "hidream", "cosmos", "ltxv", "hyvid", "wan", "lumina2", "qwen_image"}
TXT_ARCH_LIST = {"t5", "t5encoder", "llama", "qwen2vl", "qwen3", "qwen3vl"}
VIS_TYPE_LIST = {"clip-vision", "mmproj"}


def get_orig_shape(reader, tensor_name):
    field_key = f"comfy.gguf.orig_shape.{tensor_name}"
    field = reader.get_field(field_key)
    if field is None:
        return None
    if field.types != [ARRAY, INT32]:
        raise TypeError(f"Bad original shape metadata for {field_key}: Expected ARRAY of INT32, got {field.types}")
    return torch.Size(tuple(int(v) for v in field.value))


def get_field(reader, field_name, field_type):
    field = reader.get_field(field_name)
    if field is None:
        return None
    elif field_type == str:
        if field.types != [STRING]:
            raise TypeError(f"Bad type for GGUF {field_name} key: expected string, got {field.types!r}")
        return field.value
    elif field_type in [int, float, bool]:
        value = field.value
        return field_type(value[-1] if isinstance(value, tuple) else value)
    else:
        raise TypeError(f"Unknown field type {field_type}")


def get_list_field(reader, field_name, field_type):
    field = reader.get_field(field_name)
    if field is None:
        return None
    value = field.value if isinstance(field.value, tuple) else (field.value,)
    if field_type == str:
        return tuple(str(v) for v in value)
    elif field_type in [int, float, bool]:
        return tuple(field_type(v) for v in value)
    else:
        raise TypeError(f"Unknown field type {field_type}")


def gguf_sd_loader(path, handle_prefix="model.diffusion_model.", return_arch=False, is_text_model=False, *,
                   device=None, detect_arch=None, upload_threads=0):
    """Read state dict as ``GGMLTensor``s (loader.py:51-141).

    ``device=None`` (the reference's behaviour): every tensor is a read-only mmap view on the CPU.
    ``device="cuda:N"``: the file's tensor-data section is streamed into ONE HBM buffer and every
    tensor is a view into it -- packed weights resident, 16-byte aligned, ready for the HIP kernels
    (``GGUFFile.upload``; the arena is kept alive by the views).
    """
    reader = GGUFFile(path)
    try:
        # filter and strip prefix
        has_prefix = False
        if handle_prefix is not None:
            prefix_len = len(handle_prefix)
            has_prefix = any(t.name.startswith(handle_prefix) for t in reader.tensors)

        tensors = []
        for tensor in reader.tensors:
            sd_key = tensor_name = tensor.name
            if has_prefix:
                if not tensor_name.startswith(handle_prefix):
                    continue
                sd_key = tensor_name[prefix_len:]
            tensors.append((sd_key, tensor))

        # detect and verify architecture
        compat = None
        arch_str = get_field(reader, "general.architecture", str)
        type_str = get_field(reader, "general.type", str)
        if arch_str in [None, "pig", "cow"]:
            if is_text_model:
                raise ValueError(f"This gguf file is incompatible with llama.cpp!\nConsider using safetensors or a compatible gguf file\n({path})")
            compat = "sd.cpp" if arch_str is None else arch_str
            try:
                if detect_arch is None:
                    raise NotImplementedError("no architecture metadata and no detect_arch callable supplied")
                arch_str = detect_arch(set(val[0] for val in tensors)).arch
            except Exception as e:
                raise ValueError(f"This model is not currently supported - ({e})")
        elif arch_str not in TXT_ARCH_LIST and is_text_model:
            if type_str not in VIS_TYPE_LIST:
                raise ValueError(f"Unexpected text model architecture type in GGUF file: {arch_str!r}")
        elif arch_str not in IMG_ARCH_LIST and not is_text_model:
            raise ValueError(f"Unexpected architecture type in GGUF file: {arch_str!r}")

        if compat:
            logging.warning(f"Warning: This gguf model file is loaded in compatibility mode '{compat}' [arch:{arch_str}]")

        arena = reader.upload(device, threads=upload_threads) if device is not None else None

        # main loading loop
        state_dict = {}
        qtype_dict = {}
        for sd_key, tensor in tensors:
            tensor_name = tensor.name
            torch_tensor = reader.device_bytes(arena, tensor) if arena is not None else tensor.data

            shape = get_orig_shape(reader, tensor_name)
            if shape is None:
                shape = torch.Size(tuple(int(v) for v in reversed(tensor.shape)))
                # Workaround for stable-diffusion.cpp SDXL detection.
                if compat == "sd.cpp" and arch_str == "sdxl":
                    if any([tensor_name.endswith(x) for x in (".proj_in.weight", ".proj_out.weight")]):
                        while len(shape) > 2 and shape[-1] == 1:
                            shape = shape[:-1]

            # add to state dict
            if tensor.tensor_type in {Q.F32, Q.F16}:
                torch_tensor = torch_tensor.view(torch.float32 if tensor.tensor_type == Q.F32 else torch.float16).view(*shape)
            state_dict[sd_key] = GGMLTensor(torch_tensor, tensor_type=tensor.tensor_type, tensor_shape=shape)

            # 1D tensors shouldn't be quantized, this is a fix for BF16
            if len(shape) <= 1 and tensor.tensor_type == Q.BF16:
                state_dict[sd_key] = dequantize_tensor(state_dict[sd_key], dtype=torch.float32)

            # keep track of loaded tensor types
            tensor_type_str = getattr(tensor.tensor_type, "name", repr(tensor.tensor_type))
            qtype_dict[tensor_type_str] = qtype_dict.get(tensor_type_str, 0) + 1

        # print loaded tensor type counts
        logging.info("gguf qtypes: " + ", ".join(f"{k} ({v})" for k, v in qtype_dict.items()))

        # mark largest tensor for vram estimation
        qsd = {k: v for k, v in state_dict.items() if is_quantized(v)}
        if len(qsd) > 0:
            max_key = max(qsd.keys(), key=lambda k: qsd[k].numel())
            state_dict[max_key].is_largest_weight = True
    finally:
        reader.close()

    if return_arch:
        return (state_dict, arch_str)
    return state_dict


def state_dict_plan(state_dict, dtype=torch.float16, dequant_dtype=None):
    """One ``DequantPlan`` over every GPU-resident quantized tensor of a state dict that has a HIP
    unpacker: the whole weight set dequantized by one launch per (format, mode).  Returns
    (plan, keys) with ``plan.outputs[i]`` the dense tensor of ``keys[i]``."""
    from .dequant import hip_supported
    from .grouped import DequantPlan
    keys = [k for k, v in state_dict.items() if is_quantized(v) and hip_supported(v.tensor_type) and v.is_cuda]
    if not keys:
        raise ValueError("no GPU-resident quantized tensors with a HIP unpacker in this state dict")
    items = [(state_dict[k].as_subclass(torch.Tensor), state_dict[k].tensor_type, tuple(state_dict[k].tensor_shape)) for k in keys]
    return DequantPlan(items, out_dtype=dtype, dequant_dtype=dequant_dtype), keys
