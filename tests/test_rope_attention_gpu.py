"""GPU: the QKV GEMM with the RoPE rotation fused into its epilogue, and the attention kernels that
consume already-rotated q/k (fast path for sequences <= 256 tokens, general path above) - each against a
plain PyTorch fp32 reference of the same op on the same bf16 inputs.
RoPE semantics: HF rotate_half, cos/sin in fp32 (transformers LlamaRotaryEmbedding [3P]); reference call
site /root/reference/scaling_retriever/modeling/bidirectional_llama.py:67-93 (LlamaBiModel layers)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from scaling_retriever_amd import _lib as L
    return L, L.load()


def _rope_tables(hd, max_pos, theta=500000.0):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float64) / hd))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv.float()[None, :]
    return torch.cos(ang).cuda().contiguous(), torch.sin(ang).cuda().contiguous()


def _rotate(x, pos, cos, sin, hd):
    """x fp32 [T, heads, hd] -> rotated (fp32)."""
    c = torch.cat([cos[pos.long()], cos[pos.long()]], -1)[:, None, :]
    s = torch.cat([sin[pos.long()], sin[pos.long()]], -1)[:, None, :]
    rot = torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    return x * c + rot * s


@pytest.mark.parametrize("tile", ["128", "256", "split:256"])
@pytest.mark.parametrize("nh,nkv,hd,M,K", [(4, 1, 64, 333, 256), (2, 1, 128, 200, 128), (32, 8, 64, 130, 256)])
def test_qkv_gemm_with_fused_rope(tile, nh, nkv, hd, M, K, monkeypatch):
    monkeypatch.setenv("SR_GEMM_TILE", tile)
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(nh * hd + M)
    N = (nh + 2 * nkv) * hd
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    pos = torch.randint(0, 500, (M,), device="cuda", generator=g).int()
    cos, sin = _rope_tables(hd, 512)
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    L.check(lib.sr_gemm_qkv_rope(A.data_ptr(), W.data_ptr(), M, N, K, out.data_ptr(), pos.data_ptr(), cos.data_ptr(),
                                 sin.data_ptr(), (nh + nkv) * hd, hd, L.stream_ptr()), "sr_gemm_qkv_rope")
    torch.cuda.synchronize()
    y = A.float() @ W.float().T
    qk = _rotate(y[:, :(nh + nkv) * hd].reshape(M, nh + nkv, hd), pos, cos, sin, hd).reshape(M, -1)
    ref = torch.cat([qk, y[:, (nh + nkv) * hd:]], dim=1)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)
    assert (out.float() - ref).abs().max() <= 0.01 * ref.abs().max() + 1e-3


def _attn_ref(qkv, lens, key_valid, nh, nkv, hd):
    T = qkv.shape[0]
    q = qkv[:, :nh * hd].float().reshape(T, nh, hd)
    k = qkv[:, nh * hd:(nh + nkv) * hd].float().reshape(T, nkv, hd)
    v = qkv[:, (nh + nkv) * hd:].float().reshape(T, nkv, hd)
    out = torch.zeros((T, nh, hd), device="cuda")
    t0 = 0
    for n in lens:
        sl = slice(t0, t0 + n)
        kk, vv = k[sl].repeat_interleave(nh // nkv, dim=1), v[sl].repeat_interleave(nh // nkv, dim=1)
        sc = torch.einsum("qhd,khd->hqk", q[sl], kk) / hd ** 0.5
        sc = sc.masked_fill(~key_valid[sl].bool()[None, None, :], float("-inf"))
        out[sl] = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), vv)
        t0 += n
    return out.reshape(T, nh * hd)


@pytest.mark.parametrize("nh,nkv,hd,lens", [
    (4, 1, 64, [75, 1, 32, 33, 192, 256]),     # fast path (all <= 256)
    (32, 8, 64, [80, 17, 64]),
    (2, 1, 128, [70, 3, 129, 255]),
    (8, 1, 64, [100, 31]),
    (2, 2, 64, [5]),
    (4, 1, 64, [300, 40]),                     # chunked online-softmax path (a sequence > 256), q/k pre-rotated
    (2, 1, 128, [257]),
    (32, 8, 64, [512, 300, 5, 256]),           # BEIR-length passages next to short ones
    (8, 2, 128, [700, 33]),                    # three key chunks
    (4, 4, 64, [513]),
])
def test_attention_prerotated_inputs(nh, nkv, hd, lens):
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(sum(lens) * nh + hd)
    T = sum(lens)
    qkv = torch.randn((T, (nh + 2 * nkv) * hd), device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    key_valid = torch.ones(T, dtype=torch.uint8, device="cuda")
    if lens[0] > 4:
        key_valid[2] = 0
    out = torch.empty((T, nh * hd), dtype=torch.bfloat16, device="cuda")
    L.check(lib.sr_attention_varlen(qkv.data_ptr(), out.data_ptr(), cu.data_ptr(), None, key_valid.data_ptr(), None, None,
                                    len(lens), nh, nkv, hd, L.stream_ptr()), "sr_attention_varlen")
    torch.cuda.synchronize()
    ref = _attn_ref(qkv, lens, key_valid, nh, nkv, hd)
    err = (out.float() - ref).norm() / ref.norm()
    assert err < 1e-2, err
    torch.testing.assert_close(out.float(), ref, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("order", ["0", "1"])
def test_gemm_tile_order_does_not_change_the_result(order, monkeypatch):
    """Feature-tile-fastest (default) and token-tile-fastest (chosen for weight matrices far larger than the caches)
    tile orders visit the same tiles: identical output, ragged edges included."""
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 1300, 2100 // 16 * 16, 256
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    out = []
    for tile in ("128", "256", "split:512"):
        monkeypatch.setenv("SR_GEMM_TILE", tile)
        monkeypatch.setenv("SR_GEMM_MFAST", order)
        C = torch.empty((M, N), dtype=torch.float32, device="cuda")
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, 4, C.data_ptr(), None, L.stream_ptr()))
        torch.cuda.synchronize()
        out.append(C)
    torch.testing.assert_close(out[0], A.float() @ W.float().T, rtol=1e-4, atol=1e-4)
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])


@pytest.mark.parametrize("tile", ["128", "256"])
def test_gemm_many_tiles_under_load(tile, monkeypatch):
    """1024 output tiles of 256 x 256 (4 per persistent workgroup) with K = 2048: the software-pipelined k-loop and the
    cross-tile prefetch only race when every CU streams (a single-tile case hides a missing LDS-DMA wait)."""
    monkeypatch.setenv("SR_GEMM_TILE", tile)
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    M = N = 8192
    K = 2048
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
    ref = A.float() @ W.float().T
    for _ in range(3):
        C = torch.empty((M, N), dtype=torch.float32, device="cuda")
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, 4, C.data_ptr(), None, L.stream_ptr()))
        torch.cuda.synchronize()
        assert (C - ref).abs().max().item() < 1e-3
