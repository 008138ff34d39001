"""GPU: the bf16x3 precision mode of sr_dense_search (fp32 operands split into bf16 hi + lo, three bf16 MFMA
products, fp32 accumulation) against the exact fp32 path.  Tolerance: |score - fp32 score| <= 2.5e-6 |q||d|
(measured on MI355X against float64: max 1.0e-6 at H = 256, 4.5e-7 at H = 2048; the exact fp32 chain itself is
at 2e-7) - four orders of magnitude inside north_star's bar (MRR@10 within 1e-3).  Ids must agree except where
neighbouring scores are closer than that tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,n,h,k", [(100, 5000, 256, 100), (300, 40000, 512, 1000), (65, 3001, 64, 10)])
def test_bf16x3_matches_fp32_exact(nq, n, h, k):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(nq + n)
    D = torch.randn((n, h), device="cuda", generator=g) * 0.5 / h ** 0.5
    Q = torch.randn((nq, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(D)
    s0, i0 = idx.search(Q, k)                       # exact fp32
    idx.set_precision("bf16x3")
    s1, i1 = idx.search(Q, k)
    idx.set_precision("fp32")
    s2, i2 = idx.search(Q, k)
    assert torch.equal(s0, s2) and torch.equal(i0, i2)          # switching back restores the exact path
    bound = 2.5e-6 * (Q.norm(dim=1)[:, None] * D.norm(dim=1).max())
    assert ((s1 - s0).abs() <= bound).all(), float(((s1 - s0).abs() / bound).max())
    assert (s1[:, :-1] >= s1[:, 1:]).all()
    mism = i0 != i1
    if mism.any():                                   # only near-ties may swap
        assert mism.float().mean() < 0.02
        assert ((s1 - s0).abs()[mism] <= bound.expand_as(s0)[mism]).all()
    # every returned id carries (to tolerance) its true fp32 score
    true = (Q.double() @ D.double().T).float()
    got = torch.gather(true, 1, i1)
    assert ((got - s1).abs() <= bound * 2).all()


def test_bf16x3_segments_added_after_switch_and_small_batches():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(7)
    h = 128
    D = torch.randn((6000, h), device="cuda", generator=g)
    Q = torch.randn((80, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.set_precision("bf16x3")
    idx.add_device_rows(D[:2500].contiguous())
    idx.add_device_rows(D[2500:].contiguous())
    s1, i1 = idx.search(Q, 50)
    ref = torch.topk(Q @ D.T, 50, dim=1)
    assert (i1 == ref.indices).float().mean() > 0.999
    torch.testing.assert_close(s1, ref.values, rtol=1e-5, atol=1e-5)
    s2, i2 = idx.search(Q[:8].contiguous(), 50)      # <= 64 queries: exact fp32 path regardless of the mode
    exact = DenseIndexHIP(h)
    exact.add_device_rows(D)
    s3, i3 = exact.search(Q[:8].contiguous(), 50)
    assert torch.equal(s2, s3) and torch.equal(i2, i3)
