"""Synthetic weight manifests for the BASELINE.json configurations: lists of
(name, qtype, logical_shape).  No checkpoint exists in the build environment, so shapes come
from the public architectures [background, SURVEY.md section 8d] and the per-tensor quant types
from the rules the reference's quantizer patch applies (tools/lcpp.patch): only 2-D tensors are
quantized (lcpp.patch:426-429), K-quants need cols % 256 == 0 (lcpp.patch:227-253), some
tensors are bumped one level in the *_M mixes (lcpp.patch:179-192), in/out projection and
embedding layers of the image models stay unquantized (lcpp.patch:329-337, 350-360) and are
therefore absent here (they never reach the dequant kernels).
"""
from .qtypes import GGMLQuantizationType as Q

FLUX_B = (3072, 3072)       # BASELINE shape B
FLUX_C = (3072, 12288)      # BASELINE shape C


def flux_linear_pool(qtype, pairs=64):
    """configs[1]/[2]: ``pairs`` x (3072x3072 + 3072x12288) FLUX.1-dev-shaped linears of one format.
    64 pairs = 3.0 G elements: even the smallest format's PACKED bytes (Q2_K, 0.99 GB) are ~4x the
    256 MiB Infinity Cache, so neither the inputs nor the outputs of one step survive to the next."""
    out = []
    for i in range(pairs):
        out.append((f"pool.{i}.proj", Q(int(qtype)), FLUX_B))
        out.append((f"pool.{i}.mlp", Q(int(qtype)), FLUX_C))
    return out


def flux_dev(mix="Q4_K_M"):
    """configs[3]: the quantized 2-D weights of FLUX.1-dev (19 double + 38 single blocks, hidden 3072)."""
    base, bump = {"Q4_K_M": (Q.Q4_K, Q.Q5_K), "Q5_K_M": (Q.Q5_K, Q.Q6_K), "Q4_0": (Q.Q4_0, Q.Q4_0),
                  "Q8_0": (Q.Q8_0, Q.Q8_0), "Q6_K": (Q.Q6_K, Q.Q6_K)}[mix]
    m = []
    for i in range(19):
        for s in ("img", "txt"):
            p = f"double_blocks.{i}.{s}"
            m += [(f"{p}_attn.qkv.weight", bump, (9216, 3072)), (f"{p}_attn.proj.weight", base, (3072, 3072)),
                  (f"{p}_mlp.0.weight", base, (12288, 3072)), (f"{p}_mlp.2.weight", base, (3072, 12288)),
                  (f"{p}_mod.lin.weight", base, (18432, 3072))]
    for i in range(38):
        p = f"single_blocks.{i}"
        m += [(f"{p}.linear1.weight", base, (21504, 3072)), (f"{p}.linear2.weight", base, (3072, 15360)),
              (f"{p}.modulation.lin.weight", base, (9216, 3072))]
    return m


def sd35_large(mix="Q4_K_M"):
    """SD3.5-large MMDiT: 38 joint blocks, hidden 2432 -- not a multiple of 256, so K-quant rows fall
    back to a legacy format for cols=2432 (lcpp.patch:227-253 falls back when cols % 256 != 0)."""
    base = {"Q4_K_M": Q.Q4_K, "Q4_0": Q.Q4_0, "Q8_0": Q.Q8_0}[mix]
    h, m = 2432, []

    def qt(cols):
        if base in (Q.Q4_K,) and cols % 256:
            return Q.Q5_0          # the fallback llama.cpp picks for Q4_K when the row length does not fit
        return base
    for i in range(38):
        for s in ("x_block", "context_block"):
            p = f"joint_blocks.{i}.{s}"
            m += [(f"{p}.attn.qkv.weight", qt(h), (3 * h, h)), (f"{p}.attn.proj.weight", qt(h), (h, h)),
                  (f"{p}.mlp.fc1.weight", qt(h), (4 * h, h)), (f"{p}.mlp.fc2.weight", qt(4 * h), (h, 4 * h)),
                  (f"{p}.adaLN_modulation.1.weight", qt(h), (6 * h, h))]
    return m


def t5_xxl_encoder(mix="Q4_K_M"):
    """T5-v1.1-xxl encoder: 24 layers, d_model 4096, d_ff 10240, 64 heads x 64; token embedding 32128 x 4096."""
    base, emb = {"Q4_K_M": (Q.Q4_K, Q.Q6_K), "Q8_0": (Q.Q8_0, Q.Q8_0), "Q4_0": (Q.Q4_0, Q.Q4_0)}[mix]
    m = [("token_embd.weight", emb, (32128, 4096))]
    for i in range(24):
        p = f"enc.blk.{i}"
        m += [(f"{p}.attn_q.weight", base, (4096, 4096)), (f"{p}.attn_k.weight", base, (4096, 4096)),
              (f"{p}.attn_v.weight", base, (4096, 4096)), (f"{p}.attn_o.weight", base, (4096, 4096)),
              (f"{p}.ffn_gate.weight", base, (10240, 4096)), (f"{p}.ffn_up.weight", base, (10240, 4096)),
              (f"{p}.ffn_down.weight", base, (4096, 10240))]
    return m


def sd35_t5(mix="Q4_K_M"):
    """configs[4]: SD3.5-large + T5-xxl weight tensors (sharded across GPUs by sharding.partition)."""
    return sd35_large(mix) + t5_xxl_encoder(mix)
