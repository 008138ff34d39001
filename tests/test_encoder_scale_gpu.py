"""GPU parity at PRODUCTION width: Lion-1B layer dimensions (hidden 2048, 32 heads / 8 kv heads of 64, MLP 8192,
vocabulary 128 256; 2 layers so the numpy oracle finishes in seconds), the reference's batch of 128 passages.
At this size every GEMM runs its large-tile, software-pipelined, persistent configuration with all 256 CUs
streaming - the regime the tiny golden configs never reach (a missed LDS-DMA wait only shows up here).
Checked against oracle/llama_bi.py (fp32) with the tolerance of tests/test_encoder_gpu.py."""
import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB

pytestmark = pytest.mark.gpu

REL_TOL = 1.5e-2
CFG = {"hidden_size": 2048, "intermediate_size": 8192, "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 64,
       "num_hidden_layers": 2, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
       "tie_word_embeddings": True, "max_position_embeddings": 512}


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def weights():
    return make_weights(CFG, 1234, embed_std=0.05)


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


def test_dense_encode_at_1b_width_batch_128(weights):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    model = LlamaBiDense.from_weights(CFG, weights, precision="bf16").to("cuda").eval()
    ids, mask = _batch(128, 8, 160, 5)
    ref = LB.dense_encode(weights, CFG, ids, mask)
    for _ in range(2):        # twice: the second pass runs with warm caches and different timing
        out = model.doc_encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda()).cpu().numpy()
        assert out.shape == ref.shape
        per_row = np.linalg.norm(out - ref, axis=1) / np.linalg.norm(ref, axis=1)
        assert per_row.max() < 2 * REL_TOL, per_row.max()
        assert _rel(out, ref) < REL_TOL, _rel(out, ref)


def test_sparse_encode_at_1b_width(weights):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    model = LlamaBiSparse.from_weights(CFG, weights, precision="bf16").to("cuda").eval()
    ids, mask = _batch(24, 8, 128, 6, side="right")
    ref = LB.sparse_encode(weights, CFG, ids, mask)
    out = model.doc_encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda()).cpu().numpy()
    assert out.shape == ref.shape == (24, CFG["vocab_size"])
    assert _rel(out, ref) < REL_TOL, _rel(out, ref)
    flips = (out > 0) != (ref > 0)
    assert np.all(np.maximum(out, ref)[flips] < 0.05)


def test_encode_is_bitwise_reproducible_under_load(weights):
    """Races in the pipelined kernels (a missed wait, a stage refilled too early) show up as run-to-run differences long
    before they move a tolerance: the same batch encoded five times, with other batches in between, gives identical bits."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    ids, mask = _batch(128, 8, 160, 7)
    ids2, mask2 = _batch(96, 8, 192, 8)
    for cls, n in ((LlamaBiDense, 5), (LlamaBiSparse, 3)):
        model = cls.from_weights(CFG, weights, precision="bf16").to("cuda").eval()
        a = (torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda())
        b = (torch.from_numpy(ids2).cuda(), torch.from_numpy(mask2).cuda())
        first = model.doc_encode(input_ids=a[0], attention_mask=a[1]).clone()
        for _ in range(n - 1):
            model.doc_encode(input_ids=b[0], attention_mask=b[1])
            again = model.doc_encode(input_ids=a[0], attention_mask=a[1])
            assert torch.equal(again, first)
        del model
        torch.cuda.empty_cache()


def test_four_wave_gemm_loop_gives_the_eight_wave_bits_through_the_whole_encoder(weights, monkeypatch):
    """The bf16 / fp32 store, residual, SwiGLU and SwiGLU -> fp16-plane epilogues have staged-output versions for the four-wave
    256 x 256 loop.  Both regimes, dense head: SR_GEMM_BIG=8w (the 8-wave loop, direct stores), SR_GEMM_BIG=4w (four waves wherever a
    staged epilogue exists) and the default (four waves for the bf16 regime) give identical bits - every output element is the
    same k-ordered MFMA chain and the same epilogue arithmetic."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    ids, mask = _batch(128, 8, 160, 9)
    t_ids, t_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for prec in ("bf16", "fp32"):
        model = LlamaBiDense.from_weights(CFG, weights, precision=prec).to("cuda").eval()
        monkeypatch.setenv("SR_GEMM_BIG", "8w")
        want = model.doc_encode(input_ids=t_ids, attention_mask=t_mask).clone()
        monkeypatch.setenv("SR_GEMM_BIG", "4w")        # also the fp16-plane GEMMs of the fp32 regime (8-wave by default: 2 % faster there)
        got = model.doc_encode(input_ids=t_ids, attention_mask=t_mask)
        assert torch.equal(got, want), prec
        monkeypatch.delenv("SR_GEMM_BIG")
        assert torch.equal(model.doc_encode(input_ids=t_ids, attention_mask=t_mask), want), prec
        del model
        torch.cuda.empty_cache()
