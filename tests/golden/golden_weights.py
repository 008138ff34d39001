"""Deterministic synthetic Llama weights for the golden fixtures and the tests.

The fixtures store only (config, seed, inputs, outputs); the weights are
regenerated from the seed with numpy's PCG64 stream (stable across numpy
versions), in the fixed order below.  Names follow the HF Llama checkpoint
layout (`model.layers.{i}.self_attn.q_proj.weight` ...), which is what
`LlamaBiDense.load` / `LlamaBiSparse.load` read.
"""
import numpy as np


def param_shapes(cfg):
    H = cfg["hidden_size"]
    I = cfg["intermediate_size"]
    V = cfg["vocab_size"]
    nh = cfg["num_attention_heads"]
    nkv = cfg.get("num_key_value_heads") or nh
    hd = cfg.get("head_dim") or H // nh
    shapes = [("model.embed_tokens.weight", (V, H))]
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        shapes += [
            (p + "self_attn.q_proj.weight", (nh * hd, H)),
            (p + "self_attn.k_proj.weight", (nkv * hd, H)),
            (p + "self_attn.v_proj.weight", (nkv * hd, H)),
            (p + "self_attn.o_proj.weight", (H, nh * hd)),
            (p + "mlp.gate_proj.weight", (I, H)),
            (p + "mlp.up_proj.weight", (I, H)),
            (p + "mlp.down_proj.weight", (H, I)),
            (p + "input_layernorm.weight", (H,)),
            (p + "post_attention_layernorm.weight", (H,)),
        ]
    shapes.append(("model.norm.weight", (H,)))
    if not cfg.get("tie_word_embeddings", False):
        shapes.append(("lm_head.weight", (V, H)))
    return shapes


def make_weights(cfg, seed, embed_std=1.0):
    """name -> float32 array.  Linear: N(0, 1/fan_in); embed: N(0, embed_std); norms: U(0.5, 1.5)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes(cfg):
        if len(shape) == 1:
            w = 0.5 + rng.random(shape, dtype=np.float32)
        elif "embed_tokens" in name:
            w = (rng.standard_normal(shape, dtype=np.float32) * embed_std).astype(np.float32)
        else:
            w = (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(shape[1])).astype(np.float32)
        out[name] = w
    return out
