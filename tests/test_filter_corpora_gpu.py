"""The certified filter on corpora shaped like real embeddings: anisotropic rows (a common mean
component, rank-64 structure, norms spread over 0.3-0.9), the same with 2 % of the rows in near-duplicate clusters of
50-5 000 members, and queries drawn near documents - at >= 1 M documents.  Whatever the filter certifies or re-does, the
result must equal the exact kernel's (= faiss.IndexFlatIP.search restated, /root/reference/scaling_retriever/indexer.py:210-214)
bit for bit; the share of queries the filter answers itself is reported and held to a floor."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import dense_queries, dense_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("corpus,queries,min_certified", [("gauss", "gauss", 1.0), ("aniso", "aniso", 0.99), ("aniso_dup", "aniso", 0.9),
                                                          ("aniso_dup", "near_docs", 0.5)])
def test_filtered_equals_exact_on_realistic_corpora(corpus, queries, min_certified):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    N, H, nq, k = 1 << 20, 1024, 512, 1000
    dev = torch.device("cuda")
    D = dense_rows(corpus, N, H, dev, seed=5)
    Q = dense_queries(queries, nq, H, dev, seed=6, D=D)
    exact, filt = DenseIndexHIP(H), DenseIndexHIP(H)
    filt.set_precision("fp32_filtered")
    exact.add_device_rows(D)
    filt.add_device_rows(D)
    es, ei = exact.search(Q, k)
    fs, fi = filt.search(Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    cert, redone = filt.filter_query_stats()
    print(f"{corpus} / {queries}: {cert} of {nq} queries certified by the filter, {redone} re-done by the exact kernel")
    assert cert + redone == nq and cert >= min_certified * nq
