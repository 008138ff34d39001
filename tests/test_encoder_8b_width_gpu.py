"""GPU parity at Lion-DS-8B WIDTH (BASELINE.json configs[4]; reference base model
/root/reference/train_configs/mntp/meta_llama3_8b_msmarco.json:2 -> llama-3-8b): hidden 4096, 32 heads / 8 kv heads of
128, MLP 14 336, vocabulary 128 256, untied lm_head, no rope scaling; 2 layers so the numpy oracle finishes in seconds.
What only this width exercises: head_dim 128 on the production tile configurations (QKV + RoPE epilogue over 128-wide
heads, the chunked attention kernel), K = 14 336 down_proj (224 k-steps), N = 28 672 gate/up, the 6 144-wide QKV GEMM,
and a dense search at H = 4096.
Checked against oracle/llama_bi.py (fp32) with the tolerances of tests/test_encoder_gpu.py (bf16 regime) and
tests/test_fp32_regime_gpu.py (fp32 regime)."""
import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB
from oracle import scoring as SC

pytestmark = pytest.mark.gpu

REL_TOL = 1.5e-2
CFG = {"hidden_size": 4096, "intermediate_size": 14336, "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 128,
       "num_hidden_layers": 2, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
       "tie_word_embeddings": False, "max_position_embeddings": 8192}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def weights():
    return make_weights(CFG, 4321, embed_std=0.05)


def _batch(n, lo, hi, seed, side="left"):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=n)
    lens[0] = hi
    S = int(lens.max())
    ids = rng.integers(3, CFG["vocab_size"], size=(n, S)).astype(np.int64)
    mask = np.zeros((n, S), dtype=np.int64)
    for i, l in enumerate(lens):
        if side == "left":
            mask[i, S - l:] = 1
        else:
            mask[i, :l] = 1
    ids[mask == 0] = 0
    return ids, mask


def _cuda(ids, mask):
    return torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()


def _doc_encode(model, t_ids, t_mask):
    """As store_embs / SparseIndexer.index call it (indexer.py:46-52, :255-256)."""
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        return model.doc_encode(input_ids=t_ids, attention_mask=t_mask)


def test_dense_encode_at_8b_width_batch_64(weights):
    """The reference's document regime (autocast bf16), lengths up to the 192-token passage cap."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    model = LlamaBiDense.from_weights(CFG, weights, max_batch_tokens=16384, max_batch_seqs=128).to("cuda").eval()
    ids, mask = _batch(64, 8, 192, 5)
    ref = LB.dense_encode(weights, CFG, ids, mask)
    t_ids, t_mask = _cuda(ids, mask)
    first = None
    for _ in range(2):
        out = _doc_encode(model, t_ids, t_mask)
        first = out.clone() if first is None else first
        assert torch.equal(out, first)                       # bitwise reproducible
        out = out.cpu().numpy()
        per_row = np.linalg.norm(out - ref, axis=1) / np.linalg.norm(ref, axis=1)
        print(f"8B width dense bf16 regime: rel L2 {_rel(out, ref):.2e}, worst row {per_row.max():.2e}")
        assert per_row.max() < 2 * REL_TOL, per_row.max()
        assert _rel(out, ref) < REL_TOL


def test_dense_encode_at_8b_width_512_tokens(weights):
    """BEIR lengths (scripts/beir/eval_beir_dense.sh:23-24): one batch reaching 512 tokens -> the chunked attention
    kernel at head_dim 128 walks 8 key chunks."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    model = LlamaBiDense.from_weights(CFG, weights, max_batch_tokens=16384, max_batch_seqs=128).to("cuda").eval()
    ids, mask = _batch(12, 40, 512, 9)
    ref = LB.dense_encode(weights, CFG, ids, mask)
    out = _doc_encode(model, *_cuda(ids, mask)).cpu().numpy()
    print(f"8B width dense, 512 tokens: rel L2 {_rel(out, ref):.2e}")
    assert _rel(out, ref) < REL_TOL


def test_sparse_encode_at_8b_width(weights):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    model = LlamaBiSparse.from_weights(CFG, weights, max_batch_tokens=8192, max_batch_seqs=64).to("cuda").eval()
    ids, mask = _batch(16, 8, 128, 6, side="right")
    ref = LB.sparse_encode(weights, CFG, ids, mask)
    t_ids, t_mask = _cuda(ids, mask)
    out_t = _doc_encode(model, t_ids, t_mask)
    assert torch.equal(out_t, _doc_encode(model, t_ids, t_mask))
    out = out_t.cpu().numpy()
    assert out.shape == ref.shape == (16, CFG["vocab_size"])
    print(f"8B width sparse bf16 regime: rel L2 {_rel(out, ref):.2e}")
    assert _rel(out, ref) < REL_TOL
    flips = (out > 0) != (ref > 0)
    assert np.all(np.maximum(out, ref)[flips] < 0.05)


@pytest.mark.fp32_regime
def test_dense_query_encode_fp32_at_8b_width(weights):
    """The reference's dense QUERY regime (no autocast, eval_dense.py:94-106) at 8B width: query lengths up to 64."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    model = LlamaBiDense.from_weights(CFG, weights, max_batch_tokens=8192, max_batch_seqs=128).to("cuda").eval()
    ids, mask = _batch(64, 4, 64, 15)
    ref = LB.dense_encode(weights, CFG, ids, mask)
    with torch.no_grad():
        out = model.query_encode(input_ids=_cuda(ids, mask)[0], attention_mask=_cuda(ids, mask)[1]).cpu().numpy()
    e = _rel(out, ref)
    print(f"8B width dense fp32 regime: rel L2 {e:.2e}")
    assert e < 2e-5


def test_dense_search_at_h4096_bit_exact():
    """sr_dense_search at the 8B embedding width vs the oracle's k-ordered fmaf chain: ids and scores bit-exact, both the
    tiled MFMA kernel (200 queries) and the streaming kernel (8 queries)."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(7)
    H, N = 4096, 30000
    D = (rng.standard_normal((N, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
    idx = DenseIndexHIP(H)
    idx.add_host_rows(D)
    for nq, korder in ((200, "mfma_korder"), (8, "mfma_korder16")):
        Q = (rng.standard_normal((nq, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
        s, i = idx.search(torch.from_numpy(Q).cuda(), 100)
        es, ei = SC.topk_rows(SC.dense_scores_fma(Q, D, getattr(SC, korder)(H)), 100)
        assert np.array_equal(i.cpu().numpy(), ei), nq
        assert np.array_equal(s.cpu().numpy(), es), nq
