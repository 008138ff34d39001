"""Small helpers of the eval path: mirror of /root/reference/scaling_retriever/utils/utils.py
(is_first_worker :20, to_list :23, obtain_doc_vec_dir_files :26-43, supports_bfloat16 :69-75)."""
import json
import os

import torch
import torch.distributed


def is_first_worker():
    return (not torch.distributed.is_available() or not torch.distributed.is_initialized()
            or torch.distributed.get_rank() == 0)


def get_world_size():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_world_size()
    return 1


def get_rank():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank()
    return 0


def to_list(tensor):
    return tensor.detach().cpu().tolist()


def obtain_doc_vec_dir_files(doc_embed_dir):
    """plan.json {nranks, num_chunks, index_path} -> embs_{rank}_{chunk}.npy / ids_{rank}_{chunk}.npy
    lists in rank-major order."""
    with open(os.path.join(doc_embed_dir, "plan.json")) as fin:
        plan = json.load(fin)
    doc_vec_files, doc_id_files = [], []
    for i in range(plan["nranks"]):
        for j in range(plan["num_chunks"]):
            vec_file = os.path.join(doc_embed_dir, f"embs_{i}_{j}.npy")
            doc_id_file = os.path.join(doc_embed_dir, f"ids_{i}_{j}.npy")
            assert os.path.exists(vec_file) and os.path.exists(doc_id_file)
            doc_vec_files.append(vec_file)
            doc_id_files.append(doc_id_file)
    return doc_vec_files, doc_id_files


def supports_bfloat16():
    """The reference tests compute capability >= 8; on ROCm `major` is the gfx major (9 on gfx950).
    The HIP encoder always computes its GEMMs in bf16 with fp32 accumulation."""
    if torch.cuda.is_available():
        return torch.cuda.get_device_properties(torch.cuda.current_device()).major >= 8
    return False


def batch_to_device(batch, device):
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            batch[k] = v.to(device)
    return batch
