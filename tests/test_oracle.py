"""CPU: the oracle (numpy + C restatement) against the golden vectors produced by the
reference's own code (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from golden_weights import make_weights
from oracle import llama_bi as LB
from oracle import scoring as SC

ENC = ["enc_tiny_a", "enc_hd64", "enc_hd128", "enc_toy_q", "enc_toy_d"]


@pytest.mark.parametrize("name", ENC)
@pytest.mark.parametrize("side", ["left", "right"])
def test_encoder_oracle_matches_reference_heads(golden_dir, name, side):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, int(z["weight_seed"]))
    ids, mask = z[f"{side}:input_ids"], z[f"{side}:attention_mask"]
    hs = LB.forward_hidden(w, cfg, ids, mask)
    np.testing.assert_allclose(hs, z[f"{side}:last_hidden_state"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(LB.dense_encode(w, cfg, ids, mask), z[f"{side}:dense"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(LB.sparse_encode(w, cfg, ids, mask), z[f"{side}:sparse"], atol=1e-5, rtol=1e-5)


def test_toy_config_scores(golden_dir):
    """BASELINE config 1 shape: 2 queries x 2 passages, scores = q @ d.T (README.md:50-52)."""
    zq = np.load(os.path.join(golden_dir, "enc_toy_q.npz"))
    zd = np.load(os.path.join(golden_dir, "enc_toy_d.npz"))
    cfg = json.loads(str(zq["config_json"]))
    w = make_weights(cfg, int(zq["weight_seed"]))
    q = LB.dense_encode(w, cfg, zq["left:input_ids"], zq["left:attention_mask"])
    d = LB.dense_encode(w, cfg, zd["left:input_ids"], zd["left:attention_mask"])
    np.testing.assert_allclose(q @ d.T, zq["left:dense"] @ zd["left:dense"].T, atol=1e-6)


def test_bf16_emulation_tracks_reference_autocast(golden_dir):
    """Tolerance calibration: the oracle with bf16-rounded GEMMs stays as close to the reference's
    bf16-autocast outputs as those are to fp32 (~0.5 % relative L2)."""
    z = np.load(os.path.join(golden_dir, "enc_hd64.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, int(z["weight_seed"]))
    ids, mask = z["left:input_ids"], z["left:attention_mask"]
    d16 = LB.dense_encode(w, cfg, ids, mask, LB.Hooks(bf16=True))
    ref16, ref32 = z["left:dense_bf16autocast"], z["left:dense"]
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(ref16, ref32) < 0.01
    assert rel(d16, ref16) < 0.01


def test_left_pad_shift_invariance(golden_dir):
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, int(z["weight_seed"]))
    ids, mask = z["left:input_ids"], z["left:attention_mask"]
    full = LB.dense_encode(w, cfg, ids, mask)
    n = int(mask[1].sum())
    alone = LB.dense_encode(w, cfg, ids[1:2, -n:], mask[1:2, -n:])
    np.testing.assert_allclose(alone[0], full[1], atol=2e-6)


def test_lora_merge_equals_unmerged_forward():
    rng = np.random.default_rng(0)
    W = rng.standard_normal((24, 16)).astype(np.float32)
    A = rng.standard_normal((4, 16)).astype(np.float32)
    B = rng.standard_normal((24, 4)).astype(np.float32)
    x = rng.standard_normal((5, 16)).astype(np.float32)
    merged = LB.lora_merge(W, A, B, lora_alpha=8, r=4)
    np.testing.assert_allclose(x @ merged.T, x @ W.T + 2.0 * (x @ A.T) @ B.T, rtol=1e-5, atol=1e-5)


def test_sparse_oracle_bit_exact_vs_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "sparse_score.npz"))
    indptr, ids, vals, N = z["indptr"], z["doc_ids"], z["vals"], int(z["N"])
    for q in range(int(z["nq"])):
        cols, v = z[f"q{q}:cols"], z[f"q{q}:vals"]
        thr, k = float(z[f"q{q}:threshold"]), int(z[f"q{q}:k"])
        fi, neg = SC.numba_score_float(indptr, ids, vals, cols, v, thr, N)
        assert np.array_equal(fi, z[f"q{q}:filtered"]) and fi.dtype == np.int64
        assert np.array_equal(neg, z[f"q{q}:neg_scores"]) and neg.dtype == np.float32
        ti, ts = SC.select_topk(fi, neg, k)
        o = np.argsort(ti, kind="stable")
        assert np.array_equal(ti[o], z[f"q{q}:topk_idx_sorted"])
        assert np.array_equal(ts[o], z[f"q{q}:topk_score_sorted"])
        # C restatement
        oi, os_, oc = SC.sparse_retrieve_c(indptr, ids, vals, np.array([0, len(cols)]), cols, v, k, thr, N,
                                           q_threads=1, inner_threads=2)
        assert oc[0] == len(ti) and np.array_equal(oi[0, :oc[0]], ti) and np.array_equal(os_[0, :oc[0]], ts)


def test_dense_oracle_variants_agree():
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((6, 64), dtype=np.float32)
    D = rng.standard_normal((500, 64), dtype=np.float32)
    ref = np.argsort(-(Q.astype(np.float64) @ D.astype(np.float64).T), axis=1, kind="stable")[:, :20]
    s1, i1 = SC.flat_ip_search(Q, D, 20, block=128)
    s2, i2 = SC.flat_ip_search_fast(Q, D, 20, block=100)
    F = SC.dense_scores_fma(Q, D, SC.mfma_korder(64))
    s3, i3 = SC.topk_rows(F, 20)
    assert np.array_equal(i1, ref) and np.array_equal(i2, ref) and np.array_equal(i3, ref)
    np.testing.assert_allclose(s3, s1, rtol=1e-5, atol=1e-5)
    s4, i4 = SC.flat_ip_search(Q, D[:10], 20)
    assert (i4[:, 10:] == -1).all() and (s4[:, 10:] == np.float32(-3.402823466e38)).all()
    assert sorted(SC.mfma_korder(64).tolist()) == list(range(64))


def test_plan_file_order(golden_dir):
    z = np.load(os.path.join(golden_dir, "plan_files.npz"))
    exp_v = [f"embs_{i}_{j}.npy" for i in range(int(z["nranks"])) for j in range(int(z["num_chunks"]))]
    assert list(z["vec"]) == exp_v


def test_blas_heap_flat_ip_equals_the_brute_force_restatement():
    """The cpu_baseline's faiss-style search (host BLAS sgemm blocks + one heap per query in C) returns what the numpy
    restatement of IndexFlatIP.search returns (/root/reference/scaling_retriever/indexer.py:210-214): exact inner products,
    descending, ties by ascending index, (-FLT_MAX, -1) pads when k > N."""
    rng = np.random.default_rng(4)
    Q = rng.standard_normal((37, 64), dtype=np.float32)
    D = rng.standard_normal((3000, 64), dtype=np.float32)
    D[100:130] = D[7]
    for k, db, qb in ((10, 512, 16), (200, 4096, 64)):
        s, i = SC.flat_ip_search_blas_heap(Q, D, k, d_block=db, q_block=qb)
        es, ei = SC.flat_ip_search(Q, D, k)
        assert np.array_equal(i, ei) and np.allclose(s, es, rtol=1e-6, atol=1e-6)
    s, i = SC.flat_ip_search_blas_heap(Q[:3], D[:5], 8)
    assert (i[:, 5:] == -1).all() and (s[:, 5:] < -3e38).all()
