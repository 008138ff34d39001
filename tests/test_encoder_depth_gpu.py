"""GPU parity of the encoder at FULL DEPTH: Lion-1B dimensions with all 16 layers and Lion-8B dimensions with
all 32, both precision regimes, both heads, against oracle/llama_bi.py.  Every other encoder test stops at 2-3 layers
(tests/golden/make_golden.py, test_encoder_scale_gpu.py, test_encoder_8b_width_gpu.py); bench.py runs 16 and 32.

What is compared (the reference's heads, /root/reference/scaling_retriever/modeling/llm_encoder.py:186-196 sparse, :424-443 dense,
over LlamaBiModel, modeling/bidirectional_llama.py:67-188):
  fp32 regime  (dense queries, eval_dense.py:94-106)   relative L2 <= 2e-5 vs the oracle's fp32 pass, at every depth;
  bf16 regime  (documents, indexer.py:46-52)           relative L2 <= 3 x the deviation of the oracle's OWN bf16-autocast
               emulation (Hooks(bf16=True): GEMM inputs and outputs rounded to bf16) from its fp32 pass at the same depth.
The error per depth (2 / 8 / 16 / 32 layers: the d-layer model is the first d layers of the same weights + the final norm)
comes from ONE oracle pass per regime (forward_hidden's `tap`), is printed, and written to gpurun_out/encoder_depth_errors.json
for DESIGN.md.

Weights are drawn on the GPU (8B dims = 8 G parameters; numpy would need minutes) with the distributions of
tests/golden/golden_weights.py and handed to the oracle layer by layer (`_HostView`), so the host never holds a second copy.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import llama_bi as LB

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-5
BF16_FACTOR = 3.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG_1B = {"hidden_size": 2048, "intermediate_size": 8192, "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 64,
          "num_hidden_layers": 16, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
          "rope_scaling": {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                           "original_max_position_embeddings": 8192},
          "tie_word_embeddings": True, "max_position_embeddings": 131072}
CFG_8B = {"hidden_size": 4096, "intermediate_size": 14336, "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 128,
          "num_hidden_layers": 32, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
          "tie_word_embeddings": False, "max_position_embeddings": 8192}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _draw_weights(cfg, seed, dev):
    """name -> fp32 cuda tensor; Linear N(0, 1 / fan_in), embedding N(0, 0.05), norm weights U(0.5, 1.5)."""
    from golden_weights import param_shapes
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shape in param_shapes(cfg):
        if len(shape) == 1:
            out[name] = 0.5 + torch.rand(shape, device=dev, generator=g)
        elif "embed_tokens" in name:
            out[name] = torch.randn(shape, device=dev, generator=g) * 0.05
        else:
            out[name] = torch.randn(shape, device=dev, generator=g) / shape[1] ** 0.5
    return out


class _HostView:
    """What the oracle indexes as `weights`: tensors come to the host when asked for (one layer's worth alive at a time).
    rounded=True hands out bf16-rounded matrices (what autocast feeds nn.Linear) - the embedding LOOKUP stays fp32, a tied
    lm_head is the rounded embedding."""

    def __init__(self, dev_weights, rounded, tied):
        self.w, self.rounded, self.tied = dev_weights, rounded, tied

    def _fetch(self, name, as_linear):
        t = self.w[name]
        if self.rounded and as_linear and t.dim() == 2:
            t = t.bfloat16().float()
        return t.cpu().numpy()

    def __getitem__(self, name):
        return self._fetch(name, as_linear=(name != "model.embed_tokens.weight"))

    def get(self, name, default=None):
        if name == "lm_head.weight":
            if self.tied:
                return self._fetch("model.embed_tokens.weight", as_linear=True)
            return self._fetch(name, as_linear=True) if name in self.w else default
        return self[name] if name in self.w else default


class _PreRounded(LB.Hooks):
    """Hooks(bf16=True) for weights that arrive already rounded (rounding 8 G parameters in numpy takes minutes)."""

    def __init__(self):
        super().__init__(bf16=True)

    def lin(self, x, w):
        return LB.bf16_round(LB.bf16_round(x) @ w.T)


def _batch(cfg, n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=n)
    lens[0] = hi
    S = int(lens.max())
    ids = rng.integers(3, cfg["vocab_size"], size=(n, S)).astype(np.int64)
    mask = np.zeros((n, S), dtype=np.int64)
    for i, l in enumerate(lens):
        mask[i, S - l:] = 1          # left padding: the reference's eval setting (eval_dense.py:185,206)
    ids[mask == 0] = 0
    return ids, mask


def _oracle_heads(dev_w, cfg, ids, mask, depths, rounded):
    """{depth: (dense [B, H], sparse [B, V])} from one oracle pass."""
    view = _HostView(dev_w, rounded, cfg["tie_word_embeddings"])
    hooks = _PreRounded() if rounded else LB.Hooks()
    tap = {d: None for d in depths}
    LB.forward_hidden(view, cfg, ids, mask, hooks, final_norm=False, tap=tap)
    out = {}
    for d in depths:
        hs = LB.final_norm(view, cfg, tap[d])
        out[d] = (LB.dense_pool(hs, mask), LB.sparse_pool(view, cfg, hs, mask, hooks))
        tap[d] = None
    return out


def _hip_heads(dev_w, cfg, depth, ids, mask, budget):
    """{regime: (dense, sparse)} of the `depth`-layer model through sr_encode_both (bit-identical to the single-head encoders,
    tests/test_hybrid_gpu.py)."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiHybrid
    c = dict(cfg, num_hidden_layers=depth)
    keep = {k: v for k, v in dev_w.items() if ".layers." not in k or int(k.split(".layers.")[1].split(".")[0]) < depth}
    model = LlamaBiHybrid.from_weights(c, keep, max_batch_tokens=budget, max_batch_seqs=64).to("cuda").eval()
    t_ids, t_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    out = {}
    with torch.inference_mode():
        sp, de = model.encode(input_ids=t_ids, attention_mask=t_mask)                       # no autocast: fp32 regime
        out["fp32"] = (de.cpu().numpy(), sp.cpu().numpy())
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sp, de = model.encode(input_ids=t_ids, attention_mask=t_mask)
            out["bf16"] = (de.cpu().numpy(), sp.cpu().numpy())
    del model
    torch.cuda.empty_cache()
    return out


def _run(tag, cfg, depths, n_rows, lo, hi, seed):
    dev = torch.device("cuda", 0)
    dev_w = _draw_weights(cfg, seed, dev)
    ids, mask = _batch(cfg, n_rows, lo, hi, seed + 1)
    ref32 = _oracle_heads(dev_w, cfg, ids, mask, depths, rounded=False)
    ref16 = _oracle_heads(dev_w, cfg, ids, mask, depths, rounded=True)
    table, failures = [], []
    for d in depths:
        hip = _hip_heads(dev_w, cfg, d, ids, mask, budget=max(4096, n_rows * hi))
        row = {"model": tag, "layers": d}
        for h, head in enumerate(("dense", "sparse")):
            own = _rel(ref16[d][h], ref32[d][h])                     # the oracle's bf16-autocast emulation vs its fp32 pass
            e32 = _rel(hip["fp32"][h], ref32[d][h])
            e16 = _rel(hip["bf16"][h], ref32[d][h])
            row[head] = {"fp32_regime": e32, "bf16_regime": e16, "oracle_bf16_emulation": own}
            if not e32 <= FP32_TOL:
                failures.append(f"{tag} {d} layers {head}: fp32 regime {e32:.2e} > {FP32_TOL:.0e}")
            if not e16 <= BF16_FACTOR * own:
                failures.append(f"{tag} {d} layers {head}: bf16 regime {e16:.2e} > {BF16_FACTOR} x {own:.2e}")
        # the sparse head's support: an entry active on one side only must be a small one
        sp_h, sp_r = hip["fp32"][1], ref32[d][1]
        flips = (sp_h > 0) != (sp_r > 0)
        if flips.any() and np.maximum(sp_h, sp_r)[flips].max() > 1e-3:
            failures.append(f"{tag} {d} layers: sparse support differs at an entry of {np.maximum(sp_h, sp_r)[flips].max():.2e}")
        table.append(row)
        print(f"{tag} depth {d:2d}: dense fp32 {row['dense']['fp32_regime']:.2e} bf16 {row['dense']['bf16_regime']:.2e} "
              f"(oracle's own bf16 {row['dense']['oracle_bf16_emulation']:.2e}) | sparse fp32 {row['sparse']['fp32_regime']:.2e} "
              f"bf16 {row['sparse']['bf16_regime']:.2e} (own {row['sparse']['oracle_bf16_emulation']:.2e})", flush=True)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "encoder_depth_errors.json")
        old = json.load(open(path)) if os.path.exists(path) else []
        json.dump([r for r in old if r["model"] != tag] + table, open(path, "w"), indent=1)
    except OSError:
        pass
    assert not failures, "\n".join(failures)


def test_full_depth_1b_16_layers_both_regimes_both_heads():
    _run("1B", CFG_1B, depths=(2, 8, 16), n_rows=16, lo=8, hi=128, seed=2024)


def test_full_depth_8b_32_layers_both_regimes_both_heads():
    free, _ = torch.cuda.mem_get_info()
    if free < 200 << 30:
        pytest.skip("needs 200 GB of free HBM (32 GB of fp32 draws + the 32-layer model in both regimes)")
    _run("8B", CFG_8B, depths=(2, 8, 16, 32), n_rows=8, lo=6, hi=64, seed=4048)
