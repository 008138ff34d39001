#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE code.

Runs ONLY in the build container (needs /root/reference, read-only).  Nothing
from the reference (source or bytecode) is copied: this script imports the
reference's own head classes and exec's its own scoring functions in-process,
feeds them seeded inputs, and stores inputs + outputs as .npz data.

What is pinned:
  * enc_*.npz     LlamaBiDense / LlamaBiSparse .doc_encode outputs
                  (scaling_retriever/modeling/llm_encoder.py:186-196,424-443)
                  wrapped around a stock HF Llama driven with the explicit
                  bidirectional key-padding mask of
                  scaling_retriever/modeling/bidirectional_llama.py:138-161.
                  (The reference's own LlamaBiModel is silently causal under
                  the installed transformers 5.x, so the backbone arithmetic is
                  the stock HF LlamaModel [3P] with the reference's mask.)
  * sparse_score.npz  SparseRetrieval.numba_score_float / select_topk
                  (scaling_retriever/indexer.py:315-344), exec'd with the numba
                  decorators stripped (numba is not installed).
  * plan_files.npz    obtain_doc_vec_dir_files ordering
                  (scaling_retriever/utils/utils.py:26-43).
  * index_build.npz   SparseIndexer.index (scaling_retriever/indexer.py:239-308) over
                  IndexDictOfArray.add_batch_document / save
                  (scaling_retriever/utils/inverted_index.py:67-105) and merge_indexes
                  (:108-170), imported and RUN (stubs: h5py -> tests/h5py_double.py,
                  ujson -> json, faiss / numba -> empty modules; torch.distributed's rank
                  and world size patched) with a fake model that emits fixed [B, V] reps,
                  all-zero rows included, at world sizes 1 and 2: the per-term posting
                  arrays, doc_ids, nb_docs(), L0_d and the merged index are recorded.

Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    sys.path.insert(0, REF)
    peft = types.ModuleType("peft")
    for n in ["LoraConfig", "TaskType", "get_peft_model", "PeftModel"]:
        setattr(peft, n, type(n, (), {}))
    sys.modules["peft"] = peft
    uj = types.ModuleType("ujson")
    uj.load, uj.dump, uj.loads, uj.dumps = json.load, json.dump, json.loads, json.dumps
    sys.modules["ujson"] = uj
    import transformers.models.llama.modeling_llama as ml
    import transformers.models.qwen2.modeling_qwen2 as mq
    for n in ["LlamaFlashAttention2", "LlamaSdpaAttention"]:
        if not hasattr(ml, n):
            setattr(ml, n, ml.LlamaAttention)
    for n in ["Qwen2FlashAttention2", "Qwen2SdpaAttention"]:
        if not hasattr(mq, n):
            setattr(mq, n, mq.Qwen2Attention)


class BiWrap(torch.nn.Module):
    """Test double: stock HF Llama + the reference's bidirectional mask.

    Builds the additive [B,1,L,L] mask exactly as bidirectional_llama.py:138-161
    does: zeros everywhere, finfo.min on padded KEY columns only.
    """

    def __init__(self, hf_model):
        super().__init__()
        self.m = hf_model
        self.config = hf_model.config

    def forward(self, input_ids=None, attention_mask=None, return_dict=True, **kw):
        B, L = input_ids.shape
        dtype = torch.float32
        mask4 = torch.zeros(B, 1, L, L, dtype=dtype)
        mask4 = mask4.masked_fill(attention_mask[:, None, None, :].eq(0), torch.finfo(dtype).min)
        return self.m(input_ids=input_ids, attention_mask=mask4, return_dict=True)


def _reinit(model, cfg_dict, seed):
    """Load the seeded numpy weights of golden_weights.make_weights into the HF model."""
    sys.path.insert(0, OUT)
    from golden_weights import make_weights
    w = make_weights(cfg_dict, seed)
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    if cfg_dict.get("tie_word_embeddings", False):
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing


def _make_batch(rng, V, L, lengths, side, pad_id):
    B = len(lengths)
    ids = np.full((B, L), pad_id, dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    for i, n in enumerate(lengths):
        toks = rng.integers(0, V, size=n)
        if side == "left":
            ids[i, L - n:] = toks
            mask[i, L - n:] = 1
        else:
            ids[i, :n] = toks
            mask[i, :n] = 1
    return ids, mask


ENC_CASES = {
    # name: (config kwargs, L, lengths)
    # head_dim 64, MHA (2 q heads : 2 kv heads), llama3 rope scaling, tied lm_head
    "enc_tiny_a": (dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=256,
                        rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=True,
                        rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                                      "high_freq_factor": 4.0, "original_max_position_embeddings": 64}),
                   12, [12, 9, 1, 5, 12]),
    # head_dim 64 (the 1B geometry), GQA 4:1, untied head, no rope scaling
    "enc_hd64": (dict(vocab_size=384, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=1, max_position_embeddings=512,
                      rope_theta=500000.0, rms_norm_eps=1e-6, tie_word_embeddings=False),
                 40, [40, 33, 1, 17, 2, 40, 25]),
    # head_dim 128 (the 8B geometry), GQA 2:1
    "enc_hd128": (dict(vocab_size=320, hidden_size=256, intermediate_size=384, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=512,
                       rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=False),
                  70, [70, 3, 64, 65, 31]),
    # README toy shape (config 1): queries lens {9,8}, passages lens {9,10}
    "enc_toy_q": (dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                       num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
                       rope_theta=500000.0, rms_norm_eps=1e-5, tie_word_embeddings=True,
                       rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                                     "high_freq_factor": 4.0, "original_max_position_embeddings": 64}),
                  9, [9, 8]),
    "enc_toy_d": (None, 10, [9, 10]),  # shares enc_toy_q weights
}


def gen_encoder():
    from transformers import LlamaConfig, LlamaForCausalLM
    from scaling_retriever.modeling import llm_encoder as le

    shared = {}
    for ci, (name, (ckw, L, lengths)) in enumerate(ENC_CASES.items()):
        if ckw is None:
            cfg, lm, ckw, seed = shared["enc_toy_q"]
        else:
            cfg = LlamaConfig(**ckw)
            cfg._attn_implementation = "eager"
            lm = LlamaForCausalLM(cfg).eval()
            seed = 100 + ci
            _reinit(lm, ckw, seed=seed)
            if cfg.tie_word_embeddings:
                lm.tie_weights()
            shared[name] = (cfg, lm, ckw, seed)
        V = cfg.vocab_size
        rng = np.random.default_rng(7 + ci)
        out = {"config_json": np.array(json.dumps(ckw)), "weight_seed": seed}
        dense = le.LlamaBiDense(BiWrap(lm.model))
        sparse = le.LlamaBiSparse(BiWrap(lm))
        for side in ["left", "right"]:
            ids, mask = _make_batch(rng, V, L, lengths, side, pad_id=V - 1)
            t_ids, t_mask = torch.from_numpy(ids), torch.from_numpy(mask)
            with torch.no_grad():
                hs = BiWrap(lm.model)(input_ids=t_ids, attention_mask=t_mask).last_hidden_state
                d = dense.doc_encode(input_ids=t_ids, attention_mask=t_mask)
                s = sparse.doc_encode(input_ids=t_ids, attention_mask=t_mask)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    d16 = dense.doc_encode(input_ids=t_ids, attention_mask=t_mask)
                    s16 = sparse.doc_encode(input_ids=t_ids, attention_mask=t_mask)
            out[f"{side}:input_ids"] = ids
            out[f"{side}:attention_mask"] = mask
            out[f"{side}:last_hidden_state"] = hs.numpy()
            out[f"{side}:dense"] = d.float().numpy()
            out[f"{side}:sparse"] = s.float().numpy()
            out[f"{side}:dense_bf16autocast"] = d16.float().numpy()
            out[f"{side}:sparse_bf16autocast"] = s16.float().numpy()
            assert d.dtype == torch.float32 and s.dtype == torch.float32
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "dense", out["left:dense"].shape, "sparse nnz",
              int((out["left:sparse"] > 0).sum()), "of", out["left:sparse"].size)


def _load_ref_sparse_fns():
    """exec indexer.py:315-344 with the numba decorators stripped."""
    src = open(os.path.join(REF, "scaling_retriever/indexer.py")).read().split("\n")
    body = src[314:344]
    assert "def select_topk" in body[1] and "return filtered_indexes, -scores[filtered_indexes]" in body[-1], body
    kept = [ln[4:] for ln in body if "@staticmethod" not in ln and "@numba" not in ln]
    code = "\n".join(kept)

    class _Numba:
        prange = range

        class typed:
            Dict = dict
    ns = {"np": np, "numba": _Numba}
    exec(code, ns)
    return ns["numba_score_float"], ns["select_topk"]


def gen_sparse_score():
    score, topk = _load_ref_sparse_fns()
    rng = np.random.default_rng(3)
    V, N = 60, 300
    ids_d, vals_d = {}, {}
    indptr = [0]
    all_ids, all_vals = [], []
    for t in range(V):
        df = int(rng.integers(0, 80)) if t % 7 else 0   # some empty posting lists
        docs = rng.choice(N, size=df, replace=False).astype(np.int32)  # unsorted, like a merged index
        vals = np.log1p(rng.uniform(0, 20, size=df)).astype(np.float32)
        ids_d[t], vals_d[t] = docs, vals
        all_ids.append(docs)
        all_vals.append(vals)
        indptr.append(indptr[-1] + df)
    out = {"V": V, "N": N, "indptr": np.array(indptr, dtype=np.int64),
           "doc_ids": np.concatenate(all_ids), "vals": np.concatenate(all_vals)}
    nq = 12
    for qi in range(nq):
        L0 = int(rng.integers(1, 12))
        cols = np.sort(rng.choice(V, size=L0, replace=False)).astype(np.int32)
        qv = np.log1p(rng.uniform(0, 20, size=L0)).astype(np.float32)
        thr = [0.0, 0.0, 2.5][qi % 3]
        k = [10, 1000, 25][qi % 3]
        fi, neg = score(ids_d, vals_d, cols, qv, threshold=thr, size_collection=N)
        ti, ts = topk(fi, neg, k=k)
        order = np.argsort(ti, kind="stable")
        out[f"q{qi}:cols"], out[f"q{qi}:vals"] = cols, qv
        out[f"q{qi}:threshold"], out[f"q{qi}:k"] = np.float32(thr), k
        out[f"q{qi}:filtered"], out[f"q{qi}:neg_scores"] = fi.astype(np.int64), neg.astype(np.float32)
        out[f"q{qi}:topk_idx_sorted"] = ti[order].astype(np.int64)
        out[f"q{qi}:topk_score_sorted"] = ts[order].astype(np.float32)
        assert fi.dtype == np.int64 and neg.dtype == np.float32
    out["nq"] = nq
    np.savez_compressed(os.path.join(OUT, "sparse_score.npz"), **out)
    print("sparse_score: V", V, "N", N, "nq", nq)


def gen_plan_files():
    from scaling_retriever.utils.utils import obtain_doc_vec_dir_files
    with tempfile.TemporaryDirectory() as d:
        plan = {"nranks": 3, "num_chunks": 2, "index_path": os.path.join(d, "model.index")}
        json.dump(plan, open(os.path.join(d, "plan.json"), "w"))
        for r in range(3):
            for c in range(2):
                np.save(os.path.join(d, f"embs_{r}_{c}.npy"), np.zeros((1, 2), np.float32))
                np.save(os.path.join(d, f"ids_{r}_{c}.npy"), np.zeros((1,), np.int64))
        vec, idf = obtain_doc_vec_dir_files(d)
        np.savez(os.path.join(OUT, "plan_files.npz"), nranks=3, num_chunks=2,
                 vec=np.array([os.path.basename(p) for p in vec]),
                 ids=np.array([os.path.basename(p) for p in idf]))
    print("plan_files ok")


def _install_index_stubs():
    """What scaling_retriever.indexer / utils.inverted_index import beyond _install_stubs(): h5py (the test double), faiss and
    numba (never called here: decorators pass functions through)."""
    sys.path.insert(0, os.path.dirname(OUT))
    import h5py_double
    if not hasattr(h5py_double.File, "close"):
        h5py_double.File.close = lambda self: None                  # the reference closes its files by hand (:41, :100)
    h5 = types.ModuleType("h5py")
    h5.File = h5py_double.File
    sys.modules["h5py"] = h5
    sys.modules["faiss"] = types.ModuleType("faiss")
    nb = types.ModuleType("numba")
    nb.__path__ = []                                                # a package: `from numba.typed import Dict` (indexer.py:15)
    nb.njit = lambda *a, **k: (lambda f: f)
    nb.prange = range
    nb.typed = types.ModuleType("numba.typed")
    nb.typed.Dict = dict
    nb.types = types.ModuleType("numba.types")
    sys.modules["numba"], sys.modules["numba.typed"], sys.modules["numba.types"] = nb, nb.typed, nb.types


class _OnAnyDevice:
    """A batch value the reference can `.to(device)` on a box without a GPU (its device is the integer rank, indexer.py:233-235)."""

    def __init__(self, t):
        self.t = t

    def to(self, *a, **k):
        return self


class _FixedReps(torch.nn.Module):
    """encode(rows=...) -> the given rows of a fixed [n_docs, V] fp32 matrix (what LlamaBiSparse.encode returns, llm_encoder.py:186-196)."""

    def __init__(self, reps):
        super().__init__()
        self.reps = reps

    def to(self, *a, **k):
        return self

    def encode(self, rows):
        return self.reps[rows.t]


def gen_index_build():
    _install_index_stubs()
    import pickle
    import torch.distributed as dist
    from scaling_retriever import indexer as ref_indexer
    from scaling_retriever.utils import inverted_index as ref_inv
    rng = np.random.default_rng(17)
    V, N, B = 37, 23, 5                                   # ragged last batch (23 = 4 * 5 + 3)
    reps = np.zeros((N, V), dtype=np.float32)
    for d in range(N):
        nz = rng.choice(V, size=int(rng.integers(1, 9)), replace=False)
        reps[d, nz] = np.log1p(rng.uniform(0, 20, size=len(nz))).astype(np.float32)
    reps[[3, 4, 11, 22]] = 0                              # documents without a posting (indexer.py:271-283), one of them the very last
    reps[:, [0, 5]] = 0                                   # terms without a posting
    pids = [f"p{100 + 3 * d}" for d in range(N)]          # string ids, as dataset.py:25-26 keeps them
    out = {"V": V, "N": N, "B": B, "reps": reps, "pids": np.array(pids)}
    real_rank, real_ws = dist.get_rank, dist.get_world_size
    with tempfile.TemporaryDirectory() as tmp:
        for W in (1, 2):
            for rank in range(W):
                dist.get_rank, dist.get_world_size = (lambda r=rank: r), (lambda w=W: w)
                mine = list(range(rank, N, W))            # DistributedSampler(shuffle=False) without its wrap-around pad (eval_sparse.py:96)
                loader = [{"ids": [pids[d] for d in mine[b0:b0 + B]], "rows": _OnAnyDevice(torch.tensor(mine[b0:b0 + B]))}
                          for b0 in range(0, len(mine), B)]
                idx_dir = os.path.join(tmp, f"W{W}", f"index_{rank}")
                ix = ref_indexer.SparseIndexer(_FixedReps(torch.from_numpy(reps)), idx_dir, device=rank, compute_stats=True, dim_voc=V,
                                               force_new=True)
                ix.index(loader)
                tag = f"W{W}r{rank}"
                si = ix.sparse_index
                terms = sorted(int(t) for t in si.index_doc_id.keys())
                lens = [len(si.index_doc_id[t]) for t in terms]
                out[f"{tag}:terms"] = np.array(terms, dtype=np.int64)
                out[f"{tag}:lens"] = np.array(lens, dtype=np.int64)
                out[f"{tag}:doc_id"] = np.concatenate([np.asarray(si.index_doc_id[t], dtype=np.int32) for t in terms])
                out[f"{tag}:value"] = np.concatenate([np.asarray(si.index_doc_value[t], dtype=np.float32) for t in terms])
                out[f"{tag}:nb_docs"] = si.nb_docs()
                doc_ids = pickle.load(open(os.path.join(idx_dir, "doc_ids.pkl"), "rb"))
                out[f"{tag}:doc_ids_keys"] = np.array(list(doc_ids.keys()), dtype=np.int64)       # insertion order
                out[f"{tag}:doc_ids_vals"] = np.array(list(doc_ids.values()))
                out[f"{tag}:L0_d"] = np.float64(json.load(open(os.path.join(idx_dir, "index_stats.json")))["L0_d"])
                dist_ = json.load(open(os.path.join(idx_dir, "index_dist.json")))
                assert {int(k): v for k, v in dist_.items()} == dict(zip(terms, lens))
                # what a reader of the saved file sees (inverted_index.py:22-55): n = len(doc_ids) for a list, max key + 1 for a dict
                try:
                    out[f"{tag}:nb_docs_reloaded"] = ref_inv.IndexDictOfArray(idx_dir, dim_voc=V).nb_docs()
                except AssertionError:                    # a shard of rank >= 1 alone: its smallest key is not 0 (:54) - only merged indexes reload
                    out[f"{tag}:nb_docs_reloaded"] = -1
        dist.get_rank, dist.get_world_size = real_rank, real_ws
        # merge_indexes over the two rank directories (inverted_index.py:108-170); it walks os.listdir order, so both orders are recorded
        model_dir = os.path.join(tmp, "model")
        os.makedirs(model_dir)
        json.dump({"vocab_size": V}, open(os.path.join(model_dir, "config.json"), "w"))
        real_listdir = os.listdir
        for order in ("01", "10"):
            names = [f"index_{c}" for c in order]
            ref_inv.os.listdir = lambda p, names=names: list(names)
            try:
                ref_inv.merge_indexes(model_dir, index_name="index", index_dir=os.path.join(tmp, "W2"))
            finally:
                ref_inv.os.listdir = real_listdir
            mdir = os.path.join(tmp, "W2", "index")
            m = ref_inv.IndexDictOfArray(mdir, dim_voc=V)
            tag = f"merge{order}"
            out[f"{tag}:lens"] = np.array([len(m.index_doc_id[t]) for t in range(V)], dtype=np.int64)
            out[f"{tag}:doc_id"] = np.concatenate([m.index_doc_id[t] for t in range(V)]).astype(np.int32)
            out[f"{tag}:value"] = np.concatenate([m.index_doc_value[t] for t in range(V)]).astype(np.float32)
            out[f"{tag}:nb_docs"] = m.nb_docs()
            md = pickle.load(open(os.path.join(mdir, "doc_ids.pkl"), "rb"))
            out[f"{tag}:doc_ids_keys"] = np.array(list(md.keys()), dtype=np.int64)
            out[f"{tag}:doc_ids_vals"] = np.array(list(md.values()))
            out[f"{tag}:L0_d"] = np.float64(json.load(open(os.path.join(mdir, "index_stats.json")))["L0_d"])
            for f_ in os.listdir(mdir):
                os.remove(os.path.join(mdir, f_))
            os.rmdir(mdir)
    np.savez_compressed(os.path.join(OUT, "index_build.npz"), **out)
    print("index_build: V", V, "N", N, "postings W1", len(out["W1r0:doc_id"]), "nb_docs W1", out["W1r0:nb_docs"],
          "W2", out["W2r0:nb_docs"], out["W2r1:nb_docs"], "reloaded", out["W2r0:nb_docs_reloaded"], out["W2r1:nb_docs_reloaded"],
          "merged", out["merge01:nb_docs"])


if __name__ == "__main__":
    _install_stubs()
    torch.manual_seed(0)
    if sys.argv[1:] == ["index_build"]:                   # one fixture only
        gen_index_build()
        sys.exit(0)
    gen_encoder()
    gen_sparse_score()
    gen_plan_files()
    gen_index_build()
