"""ORACLE (test infrastructure, not product code): numpy restatement of the
reference's encode path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Parity status: PINNED against tests/golden/enc_*.npz, which were produced by
the reference's own LlamaBiDense / LlamaBiSparse heads
(/root/reference/scaling_retriever/modeling/llm_encoder.py:186-196, 424-443)
around a stock HF Llama driven with the reference's bidirectional key-padding
mask (/root/reference/scaling_retriever/modeling/bidirectional_llama.py:138-161).
LoRA merge (peft 0.14.0, not installed, not under /root/reference) is
"parity unpinned": restated from peft's published merge formula and
self-checked (merged forward == unmerged forward with the low-rank branch).

The transformer arithmetic itself lives in the third-party dependency
transformers==4.43.1 (requirements.txt:106), absent from /root/reference; it is
restated here from its published algorithm:
  RMSNorm      x * rsqrt(mean(x^2) + eps) * w                (fp32)
  RoPE         half-split rotate_half, inv_freq = theta^(-2i/d), optional
               "llama3" frequency scaling
  attention    softmax(q k^T / sqrt(d) + mask) v, GQA by repeating kv heads,
               mask = 0 / finfo(fp32).min on padded KEY columns, NOT causal
  MLP          down(silu(gate(x)) * up(x))
"""
import math

import numpy as np

F32_MIN = np.finfo(np.float32).min


def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (numpy emulation)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    y = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isnan(x), x, y)


def rope_inv_freq(cfg):
    """HF ROPE_INIT_FUNCTIONS['default'|'llama3'] (transformers [3P])."""
    nh = cfg["num_attention_heads"]
    hd = cfg.get("head_dim") or cfg["hidden_size"] // nh
    base = float(cfg.get("rope_theta", 10000.0))
    inv = 1.0 / (base ** (np.arange(0, hd, 2, dtype=np.float64) / hd))
    rs = cfg.get("rope_scaling") or cfg.get("rope_parameters")
    rtype = None
    if rs:
        rtype = rs.get("rope_type", rs.get("type"))
    if rs and rtype == "llama3":
        factor = rs["factor"]
        lo, hi = rs["low_freq_factor"], rs["high_freq_factor"]
        old = rs["original_max_position_embeddings"]
        low_wl, high_wl = old / lo, old / hi
        wl = 2 * math.pi / inv
        inv_l = np.where(wl > low_wl, inv / factor, inv)
        smooth = (old / wl - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = (~(wl < high_wl)) & (~(wl > low_wl))
        inv = np.where(medium, smoothed, inv_l)
    elif rs and rtype not in (None, "default"):
        raise ValueError(f"unsupported rope scaling {rtype}")
    return inv.astype(np.float32)


def _rmsnorm(x, w, eps):
    x = x.astype(np.float32)
    var = np.mean(x * x, axis=-1, keepdims=True, dtype=np.float32)
    return (x * (1.0 / np.sqrt(var + np.float32(eps)))).astype(np.float32) * w


def _rotate_half(x):
    d = x.shape[-1] // 2
    return np.concatenate([-x[..., d:], x[..., :d]], axis=-1)


def _silu(x):
    return x / (1.0 + np.exp(-x))


class Hooks:
    """Rounding hooks used to emulate the reference's bf16 autocast GEMMs."""

    def __init__(self, bf16=False):
        self.bf16 = bf16

    def lin(self, x, w):  # y = x @ w.T ; nn.Linear without bias
        if self.bf16:
            return bf16_round(bf16_round(x) @ bf16_round(w).T)
        return x @ w.T

    def act(self, x):
        return bf16_round(x) if self.bf16 else x


def forward_hidden(weights, cfg, input_ids, attention_mask, hooks=None, final_norm=True, tap=None):
    """LlamaModel.forward with the reference's bidirectional mask.

    tap: optional dict whose keys are layer counts d; tap[d] receives a copy of the residual stream after d layers
    (before the final norm) - the hidden state of the d-layer model built from the first d layers of `weights`
    (tests/test_encoder_depth_gpu.py measures the error per depth from ONE oracle pass).

    input_ids, attention_mask: [B, L] ints.  Returns last_hidden_state [B, L, H]
    float32 for ALL positions (pad query rows included, as in the reference:
    only padded KEYS are masked, bidirectional_llama.py:150-161).
    position_ids = arange(L) for every row (left pads consume positions).
    """
    hooks = hooks or Hooks()
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask)
    B, L = ids.shape
    H = cfg["hidden_size"]
    nh = cfg["num_attention_heads"]
    nkv = cfg.get("num_key_value_heads") or nh
    hd = cfg.get("head_dim") or H // nh
    eps = cfg.get("rms_norm_eps", 1e-6)
    g = nh // nkv

    inv = rope_inv_freq(cfg)
    pos = np.arange(L, dtype=np.float32)
    freqs = pos[:, None] * inv[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    cos, sin = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)  # [L, hd]

    add_mask = np.where(mask[:, None, None, :] == 0, F32_MIN, np.float32(0)).astype(np.float32)  # [B,1,1,L]

    x = weights["model.embed_tokens.weight"][ids].astype(np.float32)  # [B,L,H]
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        h = _rmsnorm(x, weights[p + "input_layernorm.weight"], eps)
        q = hooks.lin(h, weights[p + "self_attn.q_proj.weight"]).reshape(B, L, nh, hd).transpose(0, 2, 1, 3)
        k = hooks.lin(h, weights[p + "self_attn.k_proj.weight"]).reshape(B, L, nkv, hd).transpose(0, 2, 1, 3)
        v = hooks.lin(h, weights[p + "self_attn.v_proj.weight"]).reshape(B, L, nkv, hd).transpose(0, 2, 1, 3)
        q = hooks.act(q * cos[None, None] + _rotate_half(q) * sin[None, None])
        k = hooks.act(k * cos[None, None] + _rotate_half(k) * sin[None, None])
        k = np.repeat(k, g, axis=1)
        v = np.repeat(v, g, axis=1)
        s = (q @ k.transpose(0, 1, 3, 2)) * np.float32(1.0 / math.sqrt(hd)) + add_mask
        s = s - s.max(axis=-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(axis=-1, keepdims=True)
        o = hooks.act((hooks.act(pr) @ v)).transpose(0, 2, 1, 3).reshape(B, L, nh * hd)
        x = x + hooks.lin(o, weights[p + "self_attn.o_proj.weight"])
        h = _rmsnorm(x, weights[p + "post_attention_layernorm.weight"], eps)
        gate = hooks.lin(h, weights[p + "mlp.gate_proj.weight"])
        up = hooks.lin(h, weights[p + "mlp.up_proj.weight"])
        a = hooks.act(hooks.act(_silu(gate)) * up)
        x = x + hooks.lin(a, weights[p + "mlp.down_proj.weight"])
        if tap is not None and (i + 1) in tap:
            tap[i + 1] = x.astype(np.float32).copy()
    if final_norm:
        x = _rmsnorm(x, weights["model.norm.weight"], eps)
    return x.astype(np.float32)


def dense_encode(weights, cfg, input_ids, attention_mask, hooks=None):
    """DecoderOnlyBiDense.encode (llm_encoder.py:424-443).

    Per-token L2 normalise (F.normalize: x / max(||x||, 1e-12)), then the mean of
    the LAST `length` positions, length = attention_mask.sum(-1) - the literal
    `seq_reps[i, -length:, :]` slice (so with right padding it averages pad
    positions; reproduced, not fixed).  length == 0 gives `[-0:]` = all rows.
    """
    return dense_pool(forward_hidden(weights, cfg, input_ids, attention_mask, hooks), attention_mask)


def final_norm(weights, cfg, x):
    """model.norm over a residual stream taken with forward_hidden(..., tap=...)."""
    return _rmsnorm(x, weights["model.norm.weight"], cfg.get("rms_norm_eps", 1e-6)).astype(np.float32)


def dense_pool(hs, attention_mask):
    """The pooling half of DecoderOnlyBiDense.encode (llm_encoder.py:430-443) over last_hidden_state [B, L, H]."""
    nrm = np.sqrt((hs * hs).sum(-1, keepdims=True, dtype=np.float32))
    hs = hs / np.maximum(nrm, np.float32(1e-12))
    lens = np.asarray(attention_mask).sum(-1)
    L = hs.shape[1]
    out = np.stack([hs[i, (L - n) if n > 0 else 0:, :].mean(axis=0, dtype=np.float32)
                    for i, n in enumerate(lens)], axis=0)
    return out.astype(np.float32)


def sparse_encode(weights, cfg, input_ids, attention_mask, hooks=None):
    """DecoderOnlyBiSparse.encode (llm_encoder.py:186-196).

    logits = lm_head(final_norm(h)); logits *= H**-0.25;
    reps = log(relu(max_L(logits + (1 - mask) * -1e6)) + 1).
    """
    hooks = hooks or Hooks()
    return sparse_pool(weights, cfg, forward_hidden(weights, cfg, input_ids, attention_mask, hooks), attention_mask, hooks)


def sparse_pool(weights, cfg, hs, attention_mask, hooks=None):
    """The head half of DecoderOnlyBiSparse.encode (llm_encoder.py:187-196) over last_hidden_state [B, L, H]."""
    hooks = hooks or Hooks()
    w = weights.get("lm_head.weight")
    if w is None:
        w = weights["model.embed_tokens.weight"]
    logits = hooks.lin(hs, w).astype(np.float32)
    logits = logits * np.float32(cfg["hidden_size"] ** -0.25)
    m = np.asarray(attention_mask)[:, :, None].astype(np.float32)
    logits = logits + (1 - m) * np.float32(-1e6)
    mx = logits.max(axis=1)
    return np.log(np.maximum(mx, 0).astype(np.float32) + np.float32(1)).astype(np.float32)


def lora_merge(W, A, B, lora_alpha, r):
    """peft merge_and_unload [3P peft==0.14.0, parity unpinned]: W + (alpha/r) * B @ A.

    W: [out, in], A (lora_A.weight): [r, in], B (lora_B.weight): [out, r].
    """
    return (W.astype(np.float32) + np.float32(lora_alpha / r) * (B.astype(np.float32) @ A.astype(np.float32))).astype(np.float32)
