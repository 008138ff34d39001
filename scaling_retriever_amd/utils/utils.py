"""Process-group and artefact helpers of the eval path.

Same call surface as the reference's `scaling_retriever/utils/utils.py` for the names the hot path uses
(is_first_worker :20, to_list :23, obtain_doc_vec_dir_files :26-43, supports_bfloat16 :69-75); everything
else in that module (QA helpers, training-time reductions) is out of scope.
"""
import json
import os

import torch
import torch.distributed as dist


def _dist_ready():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if _dist_ready() else 0


def get_world_size():
    return dist.get_world_size() if _dist_ready() else 1


def is_first_worker():
    return get_rank() == 0


def to_list(tensor):
    return tensor.detach().cpu().tolist()


def obtain_doc_vec_dir_files(doc_embed_dir):
    """Shard manifest -> ordered file lists.

    `plan.json` = {"nranks", "num_chunks", "index_path"} (written by store_embs); the embedding and id files of
    encode rank r, chunk c are `embs_{r}_{c}.npy` / `ids_{r}_{c}.npy`, listed rank-major.  Missing files are an error.
    """
    with open(os.path.join(doc_embed_dir, "plan.json")) as fin:
        plan = json.load(fin)
    pairs = [(r, c) for r in range(int(plan["nranks"])) for c in range(int(plan["num_chunks"]))]
    vec_files = [os.path.join(doc_embed_dir, "embs_%d_%d.npy" % rc) for rc in pairs]
    id_files = [os.path.join(doc_embed_dir, "ids_%d_%d.npy" % rc) for rc in pairs]
    missing = [f for f in vec_files + id_files if not os.path.exists(f)]
    assert not missing, f"plan.json lists shard files that do not exist: {missing[:3]}"
    return vec_files, id_files


def supports_bfloat16():
    """True on MI355X.  The reference gates autocast on CUDA compute capability >= 8; on ROCm `major` is the gfx
    major (9 for gfx950).  The HIP encoder always uses bf16 GEMM inputs with fp32 accumulation, so this only
    exists for callers that branch on it."""
    return torch.cuda.is_available() and torch.cuda.get_device_properties(torch.cuda.current_device()).major >= 8


def batch_to_device(batch, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
