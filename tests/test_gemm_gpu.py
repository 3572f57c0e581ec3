"""GPU: the hand-written fp32 MFMA GEMMs (csrc/gemm_cl.hip) against fp64 torch references.
fp32 MFMA is an exact k-ordered fmaf chain, so the error bound is the usual fp32 dot-product bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(1000, 3, 64), (4096, 64, 64), (5000, 131, 128), (777, 259, 256), (2048, 128, 512), (130, 512, 128),
          (8192, 16, 8), (1, 5, 3), (33000, 64, 128),
          # tall shapes that take the row-streaming kernel (csrc/gemm_rs.hip) in fwd and/or dgrad
          (20011, 128, 256), (16400, 64, 131), (17000, 132, 40), (40000, 256, 96), (16384, 32, 256), (70000, 256, 128),
          # ... with a short last round of tiles, which the kernel cuts into column groups (8, 4 and 2 of them)
          (67601, 128, 256), (133003, 64, 64),
          # few input channels: the column-reduction wgrad (wgrad_smallk_kernel)
          (9001, 3, 64), (5000, 4, 40), (66000, 1, 128)]


def _lib():
    from graspbalance_amd import _lib
    return _lib


@pytest.mark.parametrize("P,K,N", SHAPES)
def test_gemm_fwd_stats_and_affine(P, K, N):
    L = _lib()
    torch.manual_seed(P + K + N)
    X = torch.randn(P, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    Y = torch.empty(P, N, device=DEV)
    stats = torch.zeros(2 * N, dtype=torch.float64, device=DEV)
    L.check(L.lib().gb_gemm_fwd(L.ptr(X), L.ptr(W), None, L.ptr(Y), L.ptr(stats), 1, P, K, N, None, None), "fwd")
    torch.cuda.synchronize()
    ref = X.double() @ W.double().t()
    scale = float(ref.abs().max()) + 1e-12
    assert float((Y.double() - ref).abs().max()) / scale < 2e-6
    assert torch.allclose(stats[:N], Y.double().sum(0), rtol=1e-6, atol=1e-6 * P ** 0.5)
    assert torch.allclose(stats[N:], (Y.double() ** 2).sum(0), rtol=1e-6, atol=1e-9)
    # fused BatchNorm+ReLU prologue of the next layer
    aff = torch.cat([torch.randn(K, device=DEV), torch.randn(K, device=DEV)])
    Y2 = torch.empty(P, N, device=DEV)
    L.check(L.lib().gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y2), None, 1, P, K, N, None, None), "fwd aff")
    torch.cuda.synchronize()
    ref2 = torch.relu(aff[:K] * X + aff[K:]).double() @ W.double().t()
    assert float((Y2.double() - ref2).abs().max()) / (float(ref2.abs().max()) + 1e-12) < 2e-6


@pytest.mark.parametrize("P,K,N", SHAPES)
def test_gemm_dgrad_and_wgrad(P, K, N):
    L = _lib()
    torch.manual_seed(P * 3 + K + N)
    X = torch.randn(P, K, device=DEV)
    W = torch.randn(N, K, device=DEV)
    dY = torch.randn(P, N, device=DEV)
    dX = torch.empty(P, K, device=DEV)
    L.check(L.lib().gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), None, None, None, 0, P, K, N, None, None, None, None), "dgrad")
    dW = torch.zeros(N, K, device=DEV)
    L.check(L.lib().gb_gemm_wgrad(L.ptr(dY), L.ptr(X), None, L.ptr(dW), P, K, N, None), "wgrad")
    torch.cuda.synchronize()
    rx = dY.double() @ W.double()
    rw = dY.double().t() @ X.double()
    assert float((dX.double() - rx).abs().max()) / (float(rx.abs().max()) + 1e-12) < 2e-6
    assert float((dW.double() - rw).abs().max()) / (float(rw.abs().max()) + 1e-12) < 1e-5


@pytest.mark.parametrize("P,K,N", SHAPES)
def test_gemm_fused_epilogues_slotted(P, K, N):
    """forward: prologue affine + statistics spread over slot rows; dgrad: BatchNorm-backward sums."""
    L = _lib()
    torch.manual_seed(P * 7 + K + N)
    slots = 4
    X = torch.randn(P, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    aff = torch.cat([torch.randn(K, device=DEV), torch.randn(K, device=DEV)])
    Y = torch.empty(P, N, device=DEV)
    stats = torch.zeros(slots, 2 * N, dtype=torch.float64, device=DEV)
    L.check(L.lib().gb_gemm_fwd(L.ptr(X), L.ptr(W), L.ptr(aff), L.ptr(Y), L.ptr(stats), slots, P, K, N, None, None), "fwd")
    torch.cuda.synchronize()
    ref = torch.relu(aff[:K] * X + aff[K:]).double() @ W.double().t()
    assert float((Y.double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12) < 2e-6
    st = stats.sum(0)
    assert torch.allclose(st[:N], Y.double().sum(0), rtol=1e-6, atol=1e-6 * P ** 0.5)
    assert torch.allclose(st[N:], (Y.double() ** 2).sum(0), rtol=1e-6, atol=1e-9)

    dY = torch.randn(P, N, device=DEV)
    yprev = torch.randn(P, K, device=DEV)
    ab = torch.cat([torch.randn(K, device=DEV), torch.randn(K, device=DEV), torch.randn(K, device=DEV),
                    torch.rand(K, device=DEV) + 0.5])
    dX = torch.empty(P, K, device=DEV)
    dst = torch.zeros(slots, 2 * K, dtype=torch.float64, device=DEV)
    L.check(L.lib().gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), L.ptr(yprev), L.ptr(ab), L.ptr(dst), slots, P, K, N,
                                  None, None, None, None), "dgrad bn")
    torch.cuda.synchronize()
    rx = dY.double() @ W.double()
    assert float((dX.double() - rx).abs().max()) / (float(rx.abs().max()) + 1e-12) < 2e-6
    a, b, mean, rstd = ab[:K], ab[K:2 * K], ab[2 * K:3 * K], ab[3 * K:]
    g = torch.where(a * yprev + b > 0, dX, torch.zeros_like(dX)).double()
    xhat = ((yprev - mean) * rstd).double()
    d = dst.sum(0)
    assert torch.allclose(d[:K], g.sum(0), rtol=1e-5, atol=1e-5 * P ** 0.5)
    assert torch.allclose(d[K:], (g * xhat).sum(0), rtol=1e-5, atol=1e-5 * P ** 0.5)


@pytest.mark.parametrize("P,K,N", [(1024, 1024, 256), (1024, 256, 1024), (4096, 512, 128), (2048, 1024, 256)])
def test_split_reduction_products_are_bit_reproducible(P, K, N):
    """Few-tile / long-reduction forward and dgrad products split the reduction over workgroups; the partial products
    are stored per chunk and added in chunk order, so repeated calls return identical bits (and stay within fp32
    rounding of the fp64 product)."""
    from graspbalance_amd import _lib as L
    lib = L.lib()
    g = torch.Generator(device=DEV).manual_seed(P + K)
    X = torch.randn(P, K, device=DEV, generator=g)
    W = torch.randn(N, K, device=DEV, generator=g)
    dY = torch.randn(P, N, device=DEV, generator=g)

    def fwd():
        Y = torch.full((P, N), float("nan"), device=DEV)
        L.check(lib.gb_gemm_fwd(L.ptr(X), L.ptr(W), None, L.ptr(Y), None, 1, P, K, N, None, None), "gb_gemm_fwd")
        return Y

    def dgrad():
        dX = torch.full((P, K), float("nan"), device=DEV)
        L.check(lib.gb_gemm_dgrad(L.ptr(dY), L.ptr(W), L.ptr(dX), None, None, None, 0, P, K, N, None, None, None, None),
                "gb_gemm_dgrad")
        return dX

    for fn, want in ((fwd, X.double() @ W.double().t()), (dgrad, dY.double() @ W.double())):
        first = fn()
        for _ in range(4):
            assert torch.equal(fn(), first)
        assert float((first.double() - want).norm() / want.norm()) < 2e-6
