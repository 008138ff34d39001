"""CPU tests: adapter loading rules (what peft's PeftModel.from_pretrained + merge_and_unload would apply,
/root/reference/scaling_retriever/modeling/llm_encoder.py:105-150,474-520) and the BEIR folder reader / evaluate_beir
(/root/reference/scaling_retriever/utils/metrics.py:131-151)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from golden_weights import make_weights

CFG = {"vocab_size": 64, "hidden_size": 32, "intermediate_size": 64, "num_hidden_layers": 1, "num_attention_heads": 2,
       "num_key_value_heads": 1, "rms_norm_eps": 1e-5, "rope_theta": 10000.0, "tie_word_embeddings": True, "model_type": "llama"}


def _write(tmp, adapter_tensors, adapter_cfg=None, tie=True):
    from safetensors.numpy import save_file
    cfg = dict(CFG, tie_word_embeddings=tie)
    w = make_weights(cfg, 3)
    base, lora = os.path.join(tmp, "base"), os.path.join(tmp, "lora")
    os.makedirs(base, exist_ok=True), os.makedirs(lora, exist_ok=True)
    save_file(w, os.path.join(base, "model.safetensors"))
    json.dump(cfg, open(os.path.join(base, "config.json"), "w"))
    rng = np.random.default_rng(0)
    ad = {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": rng.standard_normal((4, 32)).astype(np.float32),
          "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": rng.standard_normal((32, 4)).astype(np.float32)}
    ad.update(adapter_tensors)
    save_file(ad, os.path.join(lora, "adapter_model.safetensors"))
    json.dump(dict({"base_model_name_or_path": base, "r": 4, "lora_alpha": 8, "peft_type": "LORA",
                    "auto_mapping": {"base_model_class": "LlamaBiForMNTP"}}, **(adapter_cfg or {})),
              open(os.path.join(lora, "adapter_config.json"), "w"))
    return lora, w


@pytest.mark.parametrize("key", ["base_model.model.lm_head.weight", "base_model.model.lm_head.modules_to_save.default.weight",
                                 "base_model.model.lm_head.modules_to_save.weight"])
def test_modules_to_save_lm_head_replaces_the_base_tensor(tmp_path, key):
    """A trained lm_head shipped inside the adapter (--lora_modules_to_save) must end up as the model's head - it used to
    be dropped silently."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    head = np.random.default_rng(5).standard_normal((64, 32)).astype(np.float32)
    lora, w = _write(str(tmp_path), {key: head, "base_model.model.lm_head.original_module.weight": np.zeros((64, 32), np.float32)})
    model = LlamaBiSparse.load_from_lora(lora)
    got = model.base_model._weights
    assert np.array_equal(np.asarray(got["lm_head.weight"]), head)
    assert model.base_model.config.tie_word_embeddings is False            # the head no longer aliases embed_tokens
    assert np.array_equal(np.asarray(got["model.embed_tokens.weight"]), w["model.embed_tokens.weight"])
    assert set(model.base_model._lora["A"]) == {"model.layers.0.self_attn.q_proj.weight"}


def test_modules_to_save_embed_tokens(tmp_path):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    emb = np.random.default_rng(6).standard_normal((64, 32)).astype(np.float32)
    lora, _ = _write(str(tmp_path), {"base_model.model.model.embed_tokens.weight": emb})
    model = LlamaBiSparse.load_from_lora(lora)
    assert np.array_equal(np.asarray(model.base_model._weights["model.embed_tokens.weight"]), emb)


@pytest.mark.parametrize("tensors,cfg,exc", [
    ({"base_model.model.model.layers.0.self_attn.q_proj.lora_magnitude_vector": np.ones(32, np.float32)}, {}, ValueError),
    ({"base_model.model.model.embed_tokens.lora_embedding_A": np.ones((4, 64), np.float32)}, {}, ValueError),
    ({"base_model.model.model.layers.0.self_attn.q_proj.bias": np.ones(32, np.float32)}, {}, ValueError),
    ({"base_model.model.model.layers.9.mlp.extra.weight": np.ones((2, 2), np.float32)}, {}, ValueError),
    ({}, {"use_dora": True}, NotImplementedError),
    ({}, {"rank_pattern": {"q_proj": 8}}, NotImplementedError),
    ({}, {"alpha_pattern": {"q_proj": 8}}, NotImplementedError),
    ({}, {"bias": "all"}, NotImplementedError),
])
def test_adapter_content_the_loader_cannot_apply_is_an_error(tmp_path, tensors, cfg, exc):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    lora, _ = _write(str(tmp_path), tensors, cfg)
    with pytest.raises(exc):
        LlamaBiSparse.load_from_lora(lora)


# ------------------------------------------------------------------------------------------------ BEIR
def _beir_folder(tmp):
    d = os.path.join(tmp, "scifact")
    os.makedirs(os.path.join(d, "qrels"))
    with open(os.path.join(d, "corpus.jsonl"), "w") as f:
        for i in range(6):
            f.write(json.dumps({"_id": f"d{i}", "title": "" if i % 2 else f"T{i}", "text": f"text {i}"}) + "\n")
    with open(os.path.join(d, "queries.jsonl"), "w") as f:
        for q in ("q1", "q2", "q3", "d1"):
            f.write(json.dumps({"_id": q, "text": f"query {q}"}) + "\n")
    with open(os.path.join(d, "qrels", "test.tsv"), "w") as f:
        f.write("query-id\tcorpus-id\tscore\n")
        f.write("q1\td0\t1\nq1\td3\t2\nq2\td1\t1\nd1\td2\t1\nq2\td5\t0\n")
    return tmp


def test_load_beir_and_dataset(tmp_path):
    from scaling_retriever_amd.dataset.dataset import BeirDataset
    from scaling_retriever_amd.utils.beir import load_beir
    corpus, queries, qrels = load_beir(_beir_folder(str(tmp_path)), "scifact", split="test")
    assert len(corpus) == 6 and corpus["d2"] == {"text": "text 2", "title": "T2"}
    assert set(queries) == {"q1", "q2", "d1"}                       # q3 has no qrel in the split: dropped, like GenericDataLoader
    assert qrels == {"q1": {"d0": 1, "d3": 2}, "q2": {"d1": 1, "d5": 0}, "d1": {"d2": 1}}
    ds = BeirDataset(corpus, information_type="document")
    assert ds[2] == ("d2", "title: T2 | context: text 2")
    assert BeirDataset(queries, information_type="query")[0] == ("q1", "query q1")
    with pytest.raises(FileNotFoundError):
        load_beir(str(tmp_path), "nope")


def test_evaluate_beir_hand_computed(tmp_path):
    from scaling_retriever_amd.utils.beir import load_beir
    from scaling_retriever_amd.utils.metrics import evaluate_beir
    _, _, qrels = load_beir(_beir_folder(str(tmp_path)), "scifact")
    out = str(tmp_path / "out")
    os.makedirs(out)
    run = {"q1": {"d3": 3.0, "d9": 2.0, "d0": 1.0},          # DCG = 2/log2(2) + 1/log2(4) = 2.5; IDCG = 2 + 1/log2(3)
           "q2": {"d5": 2.0, "d4": 1.0},                     # the only relevant doc d1 is missed
           "d1": {"d1": 9.0, "d2": 1.0}}                     # the hit equal to the query id is removed first -> d2 ranks 1st
    json.dump(run, open(os.path.join(out, "run.json"), "w"))
    res = evaluate_beir(SimpleNamespace(out_dir=out), qrels)
    ndcg_q1 = 2.5 / (2.0 + 1.0 / np.log2(3.0))
    assert res["NDCG@10"] == pytest.approx(round((ndcg_q1 + 0.0 + 1.0) / 3, 5))
    assert res["Recall@100"] == pytest.approx(round((1.0 + 0.0 + 1.0) / 3, 5))
    assert res["R_cap@100"] == res["Recall@100"]             # fewer than 100 relevant docs everywhere: the cap is inactive
    assert json.load(open(os.path.join(out, "perf.json"))) == res


def test_recall_cap():
    from scaling_retriever_amd.utils.metrics import recall_cap_k, recall_k
    qrel = {"q": {f"d{i}": 1 for i in range(5)}}
    run = {"q": {"d0": 5.0, "d1": 4.0, "x": 3.0, "d2": 2.0}}
    assert recall_k(run, qrel, 2) == pytest.approx(2 / 5)
    assert recall_cap_k(run, qrel, 2) == pytest.approx(2 / 2)      # denominator min(k, #relevant)
    assert recall_cap_k(run, qrel, 4) == pytest.approx(3 / 4)
    # beir averages over every query of the run and indexes the qrels with it: a query outside the qrels is an error there
    # too, not a silent skip
    run2 = dict(run, q2={"d0": 1.0})
    qrel2 = dict(qrel, q2={"d9": 1})
    assert recall_cap_k(run2, qrel2, 2) == pytest.approx((2 / 2 + 0.0) / 2)
    with pytest.raises(KeyError):
        recall_cap_k(dict(run, stray={"d0": 1.0}), qrel, 2)
