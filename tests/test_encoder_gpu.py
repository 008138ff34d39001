"""GPU parity: HIP encoder (through the C ABI) vs the oracle / golden vectors.

Tolerances.  The reference runs documents under torch.autocast(bf16)
(/root/reference/scaling_retriever/indexer.py:46-52); its own bf16-vs-fp32 deviation on
the golden cases is 0.4-0.6 % relative L2 (tests/golden/*.npz, *_bf16autocast).  The HIP
path (bf16 GEMM inputs, fp32 accumulate / residual / norms / softmax) must stay within
REL_TOL of the fp32 golden, i.e. in the same band as the reference's own mixed precision.
Floating-point kernels (GEMM, attention) are additionally checked against a plain PyTorch
fp32 reference of the same op on the same bf16-rounded inputs.
"""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB

pytestmark = pytest.mark.gpu

REL_TOL = 1.5e-2


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _lib():
    from scaling_retriever_amd import _lib as L
    return L, L.load()


# --------------------------------------------------------------------------- GEMM
def _gemm(A, W, epi, seq_of=None, C=None, n_seq=0):
    L, lib = _lib()
    M, K = A.shape
    N = W.shape[0]
    if epi == 0:
        C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    elif epi == 2:
        C = torch.empty((M, N // 2), dtype=torch.bfloat16, device="cuda")
    elif epi == 3:
        C = torch.zeros((n_seq, N), dtype=torch.float32, device="cuda")
    elif epi == 4:
        C = torch.empty((M, N), dtype=torch.float32, device="cuda")
    L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, epi, C.data_ptr(),
                             seq_of.data_ptr() if seq_of is not None else None, L.stream_ptr()), "sr_gemm_bf16")
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 384, 128), (77, 320, 256), (1000, 3072, 2048), (1, 128, 64),
                                   (513, 2048, 8192)])
def test_gemm_store_f32_and_bf16(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    ref = A.float() @ W.float().T
    out32 = _gemm(A, W, 4)
    torch.testing.assert_close(out32, ref, rtol=1e-4, atol=1e-4)
    out16 = _gemm(A, W, 0)
    torch.testing.assert_close(out16.float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    assert (out16.float() - ref).abs().max() <= 0.02 * ref.abs().max()


def test_gemm_asymmetric_identity_catches_transposes():
    K = 128
    A = torch.eye(K, device="cuda").bfloat16()                        # A = I
    W = (torch.arange(256 * K, device="cuda").reshape(256, K) % 251).float().bfloat16()  # asymmetric, exact in bf16
    out = _gemm(A, W, 4)
    assert torch.equal(out, W.float().T.contiguous())


def test_gemm_residual_epilogue():
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 300, 256, 512
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    X = torch.randn((M, N), device="cuda", generator=g)
    ref = X + A.float() @ W.float().T
    _gemm(A, W, 1, C=X)
    torch.testing.assert_close(X, ref, rtol=1e-4, atol=1e-4)


def test_gemm_swiglu_epilogue():
    g = torch.Generator(device="cuda").manual_seed(2)
    M, I, K = 260, 384, 256
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    Wg = (torch.randn((I, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    Wu = (torch.randn((I, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    # interleave gate/up in 16-row blocks, as sr_model_set_weight lays them out
    Wgu = torch.stack([Wg.reshape(I // 16, 16, K), Wu.reshape(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
    out = _gemm(A, Wgu, 2)
    gate, up = A.float() @ Wg.float().T, A.float() @ Wu.float().T
    ref = torch.nn.functional.silu(gate) * up
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("tile", ["", "128", "256", "split:256"])
@pytest.mark.parametrize("lens", [[5, 130, 1, 64, 63, 200, 17], [3] * 90 + [1, 2, 250], [700]])
def test_gemm_segmented_max_epilogue(tile, lens, monkeypatch):
    """Per-sequence max of the logits straight from the accumulators (DPP row reduction + integer atomicMax): sequences
    shorter than, equal to and longer than a wave's 64-row slab, many sequences per slab, masked rows, every tiling."""
    monkeypatch.setenv("SR_GEMM_TILE", tile)
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = sum(lens), 320, 128
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    seq = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens)).int()
    seq[7] = -2                                  # a masked token inside a sequence: skipped
    seq[M - 1] = -2
    out = _gemm(A, W, 3, seq_of=seq.cuda(), n_seq=len(lens))
    logits = A.float() @ W.float().T
    ref = torch.zeros((len(lens), N), device="cuda")
    for s in range(len(lens)):
        rows = (seq == s).nonzero()[:, 0].cuda()
        ref[s] = logits[rows].max(dim=0).values.clamp_min(0)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------- attention
def _rope_tables(hd, max_pos, theta=500000.0):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv.float()[None, :]
    return torch.cos(ang).cuda().contiguous(), torch.sin(ang).cuda().contiguous()


def _attn_reference(qkv, lens, pos, key_valid, cos, sin, nh, nkv, hd):
    T = qkv.shape[0]
    q = qkv[:, :nh * hd].float().reshape(T, nh, hd)
    k = qkv[:, nh * hd:(nh + nkv) * hd].float().reshape(T, nkv, hd)
    v = qkv[:, (nh + nkv) * hd:].float().reshape(T, nkv, hd)
    c = torch.cat([cos[pos.long()], cos[pos.long()]], -1)[:, None, :]
    s = torch.cat([sin[pos.long()], sin[pos.long()]], -1)[:, None, :]
    rot = lambda x: torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    q = (q * c + rot(q) * s).bfloat16().float()
    k = (k * c + rot(k) * s).bfloat16().float()
    out = torch.zeros((T, nh, hd), device="cuda")
    t0 = 0
    for n in lens:
        sl = slice(t0, t0 + n)
        kk = k[sl].repeat_interleave(nh // nkv, dim=1)
        vv = v[sl].repeat_interleave(nh // nkv, dim=1)
        sc = torch.einsum("qhd,khd->hqk", q[sl], kk) / hd ** 0.5
        sc = sc.masked_fill(~key_valid[sl].bool()[None, None, :], float("-inf"))
        out[sl] = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), vv)
        t0 += n
    return out.reshape(T, nh * hd)


@pytest.mark.parametrize("nh,nkv,hd,lens", [
    (4, 1, 64, [75, 1, 32, 33, 192]),
    (2, 2, 64, [5, 64]),
    (32, 8, 64, [80, 17]),
    (2, 1, 128, [70, 3, 129]),
    (4, 1, 64, [300, 260]),           # more than one 256-key chunk
    (8, 1, 64, [100]),                # 8 q heads per kv head: several item rounds
])
def test_attention_matches_torch_fp32(nh, nkv, hd, lens):
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(sum(lens) + nh)
    T = sum(lens)
    qkv = torch.randn((T, (nh + 2 * nkv) * hd), device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    pos = torch.cat([torch.arange(n) + 3 for n in lens]).int().cuda()
    key_valid = torch.ones(T, dtype=torch.uint8, device="cuda")
    key_valid[1] = 0 if lens[0] > 2 else 1              # one masked key inside the first sequence
    cos, sin = _rope_tables(hd, 1024)
    out = torch.empty((T, nh * hd), dtype=torch.bfloat16, device="cuda")
    L.check(lib.sr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), pos.data_ptr(), key_valid.data_ptr(),
                                    cos.data_ptr(), sin.data_ptr(), len(lens), nh, nkv, hd, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = _attn_reference(qkv, lens, pos, key_valid, cos, sin, nh, nkv, hd)
    assert rel(out.float().cpu(), ref.cpu()) < 1e-2
    torch.testing.assert_close(out.float(), ref, rtol=3e-2, atol=3e-2)


# ---------------------------------------------------------------------- whole model
ENC = ["enc_tiny_a", "enc_hd64", "enc_hd128", "enc_toy_q", "enc_toy_d"]


def _load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["config_json"]))
    return z, cfg, make_weights(cfg, int(z["weight_seed"]))


@pytest.mark.parametrize("name", ENC)
@pytest.mark.parametrize("side", ["left", "right"])
def test_dense_encode_matches_reference_golden(golden_dir, name, side):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    z, cfg, w = _load_case(golden_dir, name)
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda").eval()
    ids = torch.from_numpy(z[f"{side}:input_ids"]).cuda()
    mask = torch.from_numpy(z[f"{side}:attention_mask"]).cuda()
    out = model.doc_encode(input_ids=ids, attention_mask=mask)
    assert out.dtype == torch.float32 and out.is_cuda and tuple(out.shape) == z[f"{side}:dense"].shape
    out = out.cpu().numpy()
    assert rel(out, z[f"{side}:dense"]) < REL_TOL, rel(out, z[f"{side}:dense"])
    # no worse than ~3x the reference's own bf16-autocast deviation from fp32
    assert rel(out, z[f"{side}:dense"]) < 3 * rel(z[f"{side}:dense_bf16autocast"], z[f"{side}:dense"]) + 2e-3
    # query_encode is the same function (llm_encoder.py:66-70)
    out2 = model.query_encode(input_ids=ids, attention_mask=mask).cpu().numpy()
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("name", ENC)
@pytest.mark.parametrize("side", ["left", "right"])
def test_sparse_encode_matches_reference_golden(golden_dir, name, side):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    z, cfg, w = _load_case(golden_dir, name)
    model = LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda").eval()
    ids = torch.from_numpy(z[f"{side}:input_ids"]).cuda()
    mask = torch.from_numpy(z[f"{side}:attention_mask"]).cuda()
    out = model.encode(input_ids=ids, attention_mask=mask)
    assert out.dtype == torch.float32 and tuple(out.shape) == z[f"{side}:sparse"].shape
    assert model.vocab_size == cfg["vocab_size"]
    out = out.cpu().numpy()
    assert (out >= 0).all()
    ref = z[f"{side}:sparse"]
    assert rel(out, ref) < REL_TOL, rel(out, ref)
    # zero pattern: only entries that are tiny in the reference may flip
    flips = (out > 0) != (ref > 0)
    assert np.all(np.maximum(out, ref)[flips] < 0.05)


def test_hidden_states_match_oracle(golden_dir):
    """last_hidden_state of the packed real tokens vs the fp32 oracle (per-token check)."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    z, cfg, w = _load_case(golden_dir, "enc_hd64")
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda")
    ids, mask = z["left:input_ids"], z["left:attention_mask"]
    model.encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())
    hs = model.base_model.last_hidden_state_packed().cpu().numpy()
    ref = z["left:last_hidden_state"][mask.astype(bool)]          # left padded: packed = the real tokens, row-major
    assert hs.shape == ref.shape
    per_tok = np.linalg.norm(hs - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert per_tok.max() < 3e-2, per_tok.max()


def test_toy_config_scores(golden_dir):
    """BASELINE config 1: 2 queries x 2 passages, scores = q @ d.T (examples/quick_start.py:27-30)."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    zq, cfg, w = _load_case(golden_dir, "enc_toy_q")
    zd = np.load(os.path.join(golden_dir, "enc_toy_d.npz"))
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda")
    q = model.query_encode(input_ids=torch.from_numpy(zq["left:input_ids"]).cuda(),
                           attention_mask=torch.from_numpy(zq["left:attention_mask"]).cuda())
    d = model.doc_encode(input_ids=torch.from_numpy(zd["left:input_ids"]).cuda(),
                         attention_mask=torch.from_numpy(zd["left:attention_mask"]).cuda())
    scores = torch.matmul(q, d.T).cpu().numpy()
    ref = zq["left:dense"] @ zd["left:dense"].T
    np.testing.assert_allclose(scores, ref, atol=5e-3)


def test_batch_composition_invariance(golden_dir):
    """Encoding a row alone or inside a padded batch gives the same vector (packing drops the pads)."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    z, cfg, w = _load_case(golden_dir, "enc_hd64")
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda")
    ids, mask = torch.from_numpy(z["left:input_ids"]).cuda(), torch.from_numpy(z["left:attention_mask"]).cuda()
    full = model.encode(input_ids=ids, attention_mask=mask)
    n = int(mask[3].sum())
    alone = model.encode(input_ids=ids[3:4, -n:].contiguous(), attention_mask=mask[3:4, -n:].contiguous())
    assert rel(alone.cpu().numpy()[0], full.cpu().numpy()[3]) < 5e-3


def test_encode_argument_errors(golden_dir):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    z, cfg, w = _load_case(golden_dir, "enc_tiny_a")
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda")
    ids = torch.from_numpy(z["left:input_ids"]).cuda()
    mask = torch.from_numpy(z["left:attention_mask"]).cuda().clone()
    mask[2] = 0
    # A row with an all-zero attention_mask does not abort the batch: it gets the zero vector (the reference's sparse
    # head returns exactly that, log(1 + relu(max(logits - 1e6))) = 0, llm_encoder.py:189-193) and the other rows are
    # what they are without it.
    full = model.encode(input_ids=ids, attention_mask=torch.from_numpy(z["left:attention_mask"]).cuda())
    holed = model.encode(input_ids=ids, attention_mask=mask)
    assert torch.equal(holed[2], torch.zeros_like(holed[2]))
    keep = [0, 1, 3, 4]
    assert rel(holed[keep].cpu().numpy(), full[keep].cpu().numpy()) < 1e-6
    sp = LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda")
    sp_holed = sp.encode(input_ids=ids, attention_mask=mask)
    assert torch.equal(sp_holed[2], torch.zeros_like(sp_holed[2])) and sp_holed[keep].abs().sum() > 0
    assert torch.equal(model.encode(input_ids=ids, attention_mask=torch.zeros_like(mask)), torch.zeros_like(full))
    with pytest.raises(ValueError):
        model.encode(input_ids=ids, attention_mask=mask[:, :3])
    bad = dict(cfg, num_attention_heads=8)                      # head_dim 16: unsupported
    with pytest.raises(ValueError):
        LlamaBiSparse.from_weights(bad, w, precision="bf16").to("cuda")


# ------------------------------------------------------------------ loaders / LoRA
def _write_checkpoint(tmp, cfg, w, bare):
    from safetensors.numpy import save_file
    os.makedirs(tmp, exist_ok=True)
    sd = {}
    for k, v in w.items():
        if bare:
            if k.startswith("lm_head"):
                continue
            sd[k[len("model."):]] = v
        else:
            sd[k] = v
    save_file(sd, os.path.join(tmp, "model.safetensors"))
    json.dump(cfg, open(os.path.join(tmp, "config.json"), "w"))


def _write_adapter(tmp, base_dir, cfg, prefix, rng, r=4, alpha=8, base_cls="LlamaBiModel"):
    from safetensors.numpy import save_file
    os.makedirs(tmp, exist_ok=True)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    hd = H // nh
    shapes = {"self_attn.q_proj": (nh * hd, H), "self_attn.k_proj": (nkv * hd, H), "self_attn.v_proj": (nkv * hd, H),
              "self_attn.o_proj": (H, nh * hd), "mlp.gate_proj": (I, H), "mlp.up_proj": (I, H), "mlp.down_proj": (H, I)}
    sd, AB = {}, {}
    for i in range(cfg["num_hidden_layers"]):
        for mod, (o, inn) in shapes.items():
            A = (rng.standard_normal((r, inn)) / inn ** 0.5).astype(np.float32)
            B = (rng.standard_normal((o, r)) * 0.3).astype(np.float32)
            sd[f"{prefix}layers.{i}.{mod}.lora_A.weight"] = A
            sd[f"{prefix}layers.{i}.{mod}.lora_B.weight"] = B
            AB[f"model.layers.{i}.{mod}.weight"] = (A, B)
    save_file(sd, os.path.join(tmp, "adapter_model.safetensors"))
    json.dump({"base_model_name_or_path": base_dir, "r": r, "lora_alpha": alpha,
               "target_modules": ["q_proj", "v_proj", "o_proj", "k_proj", "down_proj", "up_proj", "gate_proj"],
               "auto_mapping": {"base_model_class": base_cls, "parent_library": "x"}, "peft_type": "LORA"},
              open(os.path.join(tmp, "adapter_config.json"), "w"))
    return AB


def test_load_from_lora_dense_and_sparse(golden_dir, tmp_path):
    """load_from_lora (llm_encoder.py:131-150) + merge W + (alpha/r) B A, both adapter key layouts
    (llm_encoder.py:494-495; preprocess/lora_rewrite_from_mntp_to_bimodel.py:19-24)."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    z, cfg, w = _load_case(golden_dir, "enc_hd64")
    rng = np.random.default_rng(0)
    ids, mask = z["left:input_ids"], z["left:attention_mask"]
    t_ids, t_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    # dense: bare LlamaBiModel checkpoint, keys base_model.model.layers...
    base = str(tmp_path / "base_dense")
    _write_checkpoint(base, cfg, w, bare=True)
    AB = _write_adapter(str(tmp_path / "lora_dense"), base, cfg, "base_model.model.", rng)
    merged = dict(w)
    for k, (A, B) in AB.items():
        merged[k] = LB.lora_merge(w[k], A, B, lora_alpha=8, r=4)
    model = LlamaBiDense.load_from_lora(str(tmp_path / "lora_dense")).to("cuda").eval()
    out = model.doc_encode(input_ids=t_ids, attention_mask=t_mask).cpu().numpy()
    ref = LB.dense_encode(merged, cfg, ids, mask)
    assert rel(out, ref) < REL_TOL
    assert rel(out, z["left:dense"]) > 5 * REL_TOL            # the adapter really changed the model
    # sparse: LlamaBiForMNTP checkpoint (model.* + lm_head), keys base_model.model.model.layers...
    base_s = str(tmp_path / "base_sparse")
    _write_checkpoint(base_s, cfg, w, bare=False)
    AB = _write_adapter(str(tmp_path / "lora_sparse"), base_s, cfg, "base_model.model.model.", rng, base_cls="LlamaBiForMNTP")
    merged = dict(w)
    for k, (A, B) in AB.items():
        merged[k] = LB.lora_merge(w[k], A, B, lora_alpha=8, r=4)
    smodel = LlamaBiSparse.load_from_lora(str(tmp_path / "lora_sparse")).to("cuda").eval()
    out = smodel.encode(input_ids=t_ids, attention_mask=t_mask).cpu().numpy()
    assert rel(out, LB.sparse_encode(merged, cfg, ids, mask)) < REL_TOL
    # the dense loader rejects the MNTP key layout exactly like the reference's asserts
    with pytest.raises(AssertionError):
        LlamaBiDense.load_from_lora(str(tmp_path / "lora_sparse"))


def test_lora_merge_kernel_matches_oracle():
    L, lib = _lib()
    rng = np.random.default_rng(5)
    W = rng.standard_normal((96, 160)).astype(np.float32)
    A = rng.standard_normal((8, 160)).astype(np.float32)
    B = rng.standard_normal((96, 8)).astype(np.float32)
    dW, dA, dB = (torch.from_numpy(x).cuda() for x in (W, A, B))
    L.check(lib.sr_lora_merge(dW.data_ptr(), dA.data_ptr(), dB.data_ptr(), 96, 160, 8, 16 / 8, L.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(dW.cpu().numpy(), LB.lora_merge(W, A, B, 16, 8), rtol=1e-5, atol=1e-5)


def test_sparse_compact_matches_torch_nonzero():
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    reps = torch.relu(torch.randn((7, 1000), device="cuda", generator=g) - 1.0)
    reps[3] = 0
    row_ptr = torch.empty(8, dtype=torch.int64, device="cuda")
    cols = torch.empty(7000, dtype=torch.int32, device="cuda")
    vals = torch.empty(7000, dtype=torch.float32, device="cuda")
    n = ctypes.c_int64(0)
    L.check(lib.sr_sparse_compact(reps.data_ptr(), 7, 1000, row_ptr.data_ptr(), cols.data_ptr(), vals.data_ptr(), 7000,
                                  ctypes.byref(n), L.stream_ptr()))
    torch.cuda.synchronize()
    r, c = torch.nonzero(reps, as_tuple=True)
    assert n.value == len(r)
    assert torch.equal(cols[:n.value].long(), c) and torch.equal(vals[:n.value], reps[r, c])
    counts = torch.bincount(r, minlength=7)
    assert torch.equal(row_ptr[1:] - row_ptr[:-1], counts)
    rc = lib.sr_sparse_compact(reps.data_ptr(), 7, 1000, row_ptr.data_ptr(), cols.data_ptr(), vals.data_ptr(), 3,
                               ctypes.byref(n), L.stream_ptr())
    assert rc == L.SR_ERR_NOMEM and n.value == len(r)


@pytest.mark.parametrize("tile", ["128", "256", "split:512"])
def test_gemm_both_tile_configs_all_epilogues(tile, monkeypatch):
    """The 256 x 256 (8-wave) and 128 x 128 (4-wave) configurations of the GEMM template and the row cut between them
    (first 512 token rows on 256^2 tiles, the rest on 128^2), forced through the SR_GEMM_TILE switch, on ragged shapes."""
    monkeypatch.setenv("SR_GEMM_TILE", tile)
    g = torch.Generator(device="cuda").manual_seed(len(tile))
    M, N, K = 700, 800, 256
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    ref = A.float() @ W.float().T
    torch.testing.assert_close(_gemm(A, W, 4), ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(_gemm(A, W, 0).float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    X = torch.randn((M, N), device="cuda", generator=g)
    want = X + ref
    _gemm(A, W, 1, C=X)
    torch.testing.assert_close(X, want, rtol=1e-4, atol=1e-4)
    I = N // 2
    Wg, Wu = W[:I].contiguous(), W[I:].contiguous()
    Wgu = torch.stack([Wg.reshape(I // 16, 16, K), Wu.reshape(I // 16, 16, K)], dim=1).reshape(2 * I, K).contiguous()
    out = _gemm(A, Wgu, 2)
    sw = torch.nn.functional.silu(A.float() @ Wg.float().T) * (A.float() @ Wu.float().T)
    torch.testing.assert_close(out.float(), sw, rtol=2e-2, atol=2e-2)
    eye = torch.eye(K, device="cuda").bfloat16()
    Wa = (torch.arange(512 * K, device="cuda").reshape(512, K) % 251).float().bfloat16()
    assert torch.equal(_gemm(eye, Wa, 4), Wa.float().T.contiguous())


def test_gemm_tile_plan_does_not_change_the_result(monkeypatch):
    """Every output element is one k-ordered MFMA chain whatever tile computes it: the automatic plan (which cuts
    9600 x 2048 into 8192 rows of 256^2 tiles + 1408 rows of 128^2 tiles) is bit-identical to either forced tiling."""
    g = torch.Generator(device="cuda").manual_seed(77)
    M, N, K = 9600, 2048, 1024
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    auto = _gemm(A, W, 4)
    torch.testing.assert_close(auto, A.float() @ W.float().T, rtol=1e-4, atol=1e-4)
    for tile in ("128", "256", "split:4096"):
        monkeypatch.setenv("SR_GEMM_TILE", tile)
        assert torch.equal(_gemm(A, W, 4), auto), tile


@pytest.mark.parametrize("M,N,K", [(16384, 2048, 512), (9600, 3072, 256), (5000, 768, 256), (70000, 2048, 128), (2048, 16384, 256)])
def test_gemm_xcd_aware_tile_order_covers_every_tile_once(M, N, K, monkeypatch):
    """The XCD-aware order only changes WHICH workgroup computes a tile (compact blocks of tiles per XCD, padded slots at the
    ragged edge skipped): every output element must come out, bit for bit, as with the plain order - for tile grids whose
    feature-tile count is a multiple of 8, of 4 (12), odd (3), and for more rounds than one."""
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    L, lib = _lib()

    def run():       # into a NaN-filled buffer: a tile nobody computed cannot go unnoticed
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, 4, C.data_ptr(), None, L.stream_ptr()), "sr_gemm_bf16")
        torch.cuda.synchronize()
        return C

    for tile in ("256", "128", ""):
        monkeypatch.setenv("SR_GEMM_TILE", tile)
        monkeypatch.setenv("SR_GEMM_XCD", "0")
        plain = run()
        assert not torch.isnan(plain).any()
        monkeypatch.setenv("SR_GEMM_XCD", "1")
        assert torch.equal(run(), plain), tile
    torch.testing.assert_close(plain[:512], A[:512].float() @ W.float().T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M", [1, 9, 16, 17, 33, 64])
def test_gemm_few_token_rows_streaming_configuration(M, monkeypatch):
    """Up to 64 token rows (online queries) run single-wave workgroups that stream W; every output element is the same
    MFMA chain as in the tiled configurations: bit-identical for the store, residual, SwiGLU and per-sequence-max epilogues."""
    g = torch.Generator(device="cuda").manual_seed(M)
    N, K = 640, 512
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    X = torch.randn((M, N), device="cuda", generator=g)
    seq = (torch.arange(M) // 5).int().cuda()
    n_seq = int(seq.max().item()) + 1

    def run_all():
        return [_gemm(A, W, 4), _gemm(A, W, 0), _gemm(A, W, 2), _gemm(A, W, 1, C=X.clone()),
                _gemm(A, W, 3, seq_of=seq, n_seq=n_seq)]
    got = run_all()
    monkeypatch.setenv("SR_GEMM_SKINNY", "0")
    want = run_all()
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    torch.testing.assert_close(got[0], A.float() @ W.float().T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(700, 800, 256), (16384, 2048, 320), (9000, 3072, 2048), (2100, 16384, 256), (66000, 512, 1024)])
def test_gemm_four_wave_loop_equals_eight_wave_loop(M, N, K, monkeypatch):
    """The 256 x 256 tile runs on FOUR waves of 128 x 128 by default (the vendor library's configuration: hand-ordered k-step,
    accumulators in AGPRs, LDS-DMA through buffer descriptors two k-steps ahead); SR_GEMM_BIG=8w selects the 8-wave loop.  Every
    output element is the same k-ordered MFMA chain in both: all epilogues bit-identical, on ragged shapes (rows past the matrix
    read as zeros through the descriptor's bounds), on 4 k-steps (the loop's minimum), an odd count, and on more tiles than
    resident workgroups (the k-loop runs across tile boundaries: the last two k-steps fetch the next tile's first two)."""
    monkeypatch.setenv("SR_GEMM_TILE", "256")
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    seq_of = torch.sort(torch.randint(0, 37, (M,), device="cuda", generator=g)).values.to(torch.int32)
    seq_of[5:9] = -2
    X0 = torch.randn((M, N), device="cuda", generator=g)

    def all_epilogues():
        X = X0.clone()
        _gemm(A, W, 1, C=X)
        return [_gemm(A, W, 4), _gemm(A, W, 0), X, _gemm(A, W, 2), _gemm(A, W, 3, seq_of=seq_of, n_seq=37)]
    monkeypatch.setenv("SR_GEMM_BIG", "8w")
    ref = all_epilogues()
    monkeypatch.delenv("SR_GEMM_BIG")
    for _ in range(2):
        got = all_epilogues()
        for name, a, b in zip(("f32", "bf16", "residual", "swiglu", "segmax"), got, ref):
            assert torch.equal(a, b), name
    torch.testing.assert_close(got[0][:300], A[:300].float() @ W.float().T, rtol=1e-4, atol=1e-4)
    # a NaN-filled output: a tile nobody computed cannot go unnoticed
    L, lib = _lib()
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, 4, C.data_ptr(), None, L.stream_ptr()), "sr_gemm_bf16")
    assert torch.equal(C, ref[0])
    # QKV + RoPE (bf16 out): the four-wave loop's staged epilogue (rotation on the accumulators, round 6) against the 8-wave loop's
    # direct stores, for both head sizes; and the fp32 regime's scaled fp16 GEMM (+= into fp32) on the four-wave loop (SR_GEMM_BIG=4w)
    if N % 128 == 0:
        pos = torch.randint(0, 300, (M,), device="cuda", generator=g).to(torch.int32)
        for hd in (64, 128):
            ang = torch.arange(512, device="cuda")[:, None].float() * (1.0 / 10000 ** (torch.arange(hd // 2, device="cuda").float() / (hd // 2)))[None, :]
            cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
            n_rope = (N // hd) * hd * 3 // 4 // hd * hd

            def rope():
                out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
                L.check(lib.sr_gemm_qkv_rope(A.data_ptr(), W.data_ptr(), M, N, K, out.data_ptr(), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                             n_rope, hd, L.stream_ptr()), "sr_gemm_qkv_rope")
                return out
            monkeypatch.setenv("SR_GEMM_BIG", "8w")
            want = rope()
            monkeypatch.delenv("SR_GEMM_BIG")
            assert torch.equal(rope(), want), hd
    Ah, Wh = A.float().half(), W.float().half()
    sa = torch.exp2(torch.randint(-3, 4, (M,), device="cuda", generator=g).float())
    sw = torch.exp2(torch.randint(-3, 4, (N,), device="cuda", generator=g).float())

    def scaled():
        X = X0.clone()
        L.check(lib.sr_gemm_f16_scaled(Ah.data_ptr(), Wh.data_ptr(), M, N, K, sa.data_ptr(), sw.data_ptr(), X.data_ptr(), L.stream_ptr()), "sr_gemm_f16_scaled")
        return X
    monkeypatch.setenv("SR_GEMM_BIG", "8w")
    want = scaled()
    monkeypatch.setenv("SR_GEMM_BIG", "4w")
    assert torch.equal(scaled(), want)
    monkeypatch.delenv("SR_GEMM_BIG")
