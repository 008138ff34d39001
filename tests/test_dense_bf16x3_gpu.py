"""GPU: the split-bf16 precision modes of sr_dense_search against the exact fp32 path.
  bf16x3: fp32 operands split into 2 bf16 planes, 3 plane products, fp32 accumulation.
          Tolerance |score - fp32 score| <= 2.5e-6 |q||d| (measured on MI355X against float64: max 1.0e-6 at
          H = 256, 4.5e-7 at H = 2048) - four orders of magnitude inside north_star's bar (MRR@10 within 1e-3).
  bf16x6: 3 planes (the whole 24-bit significand), 6 plane products.  Tolerance 4e-7 |q||d| - the error class of
          the exact fp32 chain itself (measured: bf16x6 max 8.2e-8, exact fp32 chain 2.0e-7 against float64).
Ids must agree except where neighbouring scores are closer than the tolerance."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {"bf16x3": 2.5e-6, "bf16x6": 4e-7}


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
@pytest.mark.parametrize("nq,n,h,k", [(100, 5000, 256, 100), (300, 40000, 512, 1000), (65, 3001, 64, 10),
                                      (513, 70001, 2048, 100)])
def test_split_modes_match_fp32_exact(mode, nq, n, h, k):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(nq + n)
    D = torch.randn((n, h), device="cuda", generator=g) * 0.5 / h ** 0.5
    Q = torch.randn((nq, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(D)
    s0, i0 = idx.search(Q, k)                       # exact fp32
    idx.set_precision(mode)
    s1, i1 = idx.search(Q, k)
    s1b, i1b = idx.search(Q, k)
    assert torch.equal(s1, s1b) and torch.equal(i1, i1b)        # deterministic
    idx.set_precision("fp32")
    s2, i2 = idx.search(Q, k)
    assert torch.equal(s0, s2) and torch.equal(i0, i2)          # switching back restores the exact path
    bound = TOL[mode] * (Q.norm(dim=1)[:, None] * D.norm(dim=1).max())
    assert ((s1 - s0).abs() <= bound).all(), float(((s1 - s0).abs() / bound).max())
    assert (s1[:, :-1] >= s1[:, 1:]).all()
    mism = i0 != i1
    if mism.any():                                   # only near-ties may swap
        assert mism.float().mean() < 0.02
        assert ((s1 - s0).abs()[mism] <= bound.expand_as(s0)[mism]).all()
    # every returned id carries (to tolerance) its true score (float64 arithmetic)
    true = (Q.double() @ D.double().T).float()
    got = torch.gather(true, 1, i1)
    assert ((got - s1).abs() <= bound * 2).all()


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
def test_split_modes_segments_added_after_switch_and_small_batches(mode):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(7)
    h = 128
    D = torch.randn((6000, h), device="cuda", generator=g)
    Q = torch.randn((80, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.set_precision(mode)
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


def test_switching_between_split_modes_reuses_planes():
    """x3 -> x6 adds the third plane, x6 -> x3 keeps using the first two: results equal a fresh index in that mode."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(9)
    h = 256
    D = torch.randn((9000, h), device="cuda", generator=g)
    Q = torch.randn((130, h), device="cuda", generator=g)
    res = {}
    for mode in ("bf16x3", "bf16x6"):
        f = DenseIndexHIP(h)
        f.add_device_rows(D)
        f.set_precision(mode)
        res[mode] = f.search(Q, 20)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(D)
    for mode in ("bf16x3", "bf16x6", "bf16x3", "fp32", "bf16x6"):
        idx.set_precision(mode)
        s, i = idx.search(Q, 20)
        if mode in res:
            assert torch.equal(s, res[mode][0]) and torch.equal(i, res[mode][1])
