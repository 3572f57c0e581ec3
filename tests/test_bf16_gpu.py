"""GPU: the bf16 MLP path of BASELINE configs[4] ("mixed bf16 MLP / fp32 geometry", fused_mlp.set_precision('bf16')):
v_mfma_f32_32x32x16_bf16 variants of the row-streaming and LDS-tiled GEMMs - operands rounded to bf16 on the way
into the matrix cores, fp32 accumulation, fp32 BatchNorm statistics, fp32 tensors in memory, fp32 geometry.

Truth for the GEMM tests: the same product in fp64 with both operands ROUNDED TO BF16 first (then the kernel must be
exact to fp32 accumulation noise: 1e-5), and the un-rounded product (bf16 rounding noise: ~2^-9 / sqrt(K) relative)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture()
def bf16():
    from graspbalance_amd import fused_mlp
    prev = fused_mlp.set_precision("bf16")
    yield
    fused_mlp.set_precision(prev)


def _r(t):
    return t.to(torch.bfloat16).double()


def _rel(a, b):
    return float((a.double() - b).norm() / b.norm())


@pytest.mark.parametrize("P,K,N", [(65536, 64, 128), (40000, 128, 256), (16384, 256, 256), (4096, 1024, 256),
                                   (4096, 256, 1024), (2048, 131, 128), (1000, 48, 40), (32768 + 17, 64, 64)])
def test_bf16_gemms_against_rounded_operand_truth(bf16, P, K, N):
    from graspbalance_amd import _lib, fused_mlp
    torch.manual_seed(P + K + N)
    X = torch.randn(P, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    dY = torch.randn(P, N, device=DEV)
    a = torch.rand(K, device=DEV) + 0.5
    b = torch.randn(K, device=DEV) * 0.3
    aff = torch.cat([a, b]).contiguous()
    st = fused_mlp._s(X)
    # forward with the BatchNorm+ReLU prologue and the column statistics
    Y = torch.empty(P, N, device=DEV)
    slots = fused_mlp.STAT_SLOTS if P >= 16384 else 1
    stats = torch.zeros(slots * 2 * N, dtype=torch.float64, device=DEV)
    opts = fused_mlp._opts(X.device, st, _lib.PREC_BF16)
    fused_mlp._call("gb_gemm_fwd", X.device, _lib.ptr(X), _lib.ptr(W), _lib.ptr(aff), _lib.ptr(Y), _lib.ptr(stats), slots,
                    P, K, N, None, opts, st)
    A = torch.relu(a * X + b)
    want = _r(A) @ _r(W).t()
    assert _rel(Y, want) < 1e-5
    assert _rel(Y, A.double() @ W.double().t()) < 2e-2 / K ** 0.25
    s = stats.view(slots, 2 * N).sum(0)
    assert _rel(s[:N], Y.double().sum(0)) < 1e-6 and _rel(s[N:], (Y.double() ** 2).sum(0)) < 1e-6
    # dgrad
    dX = torch.empty(P, K, device=DEV)
    fused_mlp._call("gb_gemm_dgrad", X.device, _lib.ptr(dY), _lib.ptr(W), _lib.ptr(dX), None, None, None, 0, P, K, N, None,
                    None, None, opts, st)
    assert _rel(dX, _r(dY) @ _r(W)) < 1e-5
    # wgrad with the prologue on X
    dW = torch.zeros(N, K, device=DEV)
    fused_mlp._call("gb_gemm_wgrad", X.device, _lib.ptr(dY), _lib.ptr(X), _lib.ptr(aff), _lib.ptr(dW), P, K, N, opts, st)
    assert _rel(dW, _r(dY).t() @ _r(A)) < 2e-5


def test_short_reductions_stay_fp32(bf16):
    """K = 3 (xyz-only first layers): the fp32 instruction, bit-identical to the fp32 mode."""
    from graspbalance_amd import _lib, fused_mlp
    torch.manual_seed(0)
    X = torch.randn(50000, 3, device=DEV)
    W = torch.randn(64, 3, device=DEV)
    out = []
    for mode in (_lib.PREC_BF16, _lib.PREC_F32):
        Y = torch.empty(50000, 64, device=DEV)
        st = fused_mlp._s(X)
        fused_mlp._call("gb_gemm_fwd", X.device, _lib.ptr(X), _lib.ptr(W), None, _lib.ptr(Y), None, 1, 50000, 3, 64, None,
                        fused_mlp._opts(X.device, st, mode), st)
        out.append(Y)
    assert torch.equal(out[0], out[1])


def test_sa_module_and_cylinder_head_bf16_vs_fp32(golden):
    """A set-abstraction level and a GraspWidthGrouping head: bf16 mode vs fp32 mode, forward and gradients at bf16
    tolerance; indices (geometry stays fp32) identical."""
    from graspbalance_amd import fused_mlp, fused_ops, pointnet2_modules as pm
    from graspbalance_amd.modules import GraspWidthGrouping
    from graspbalance_amd.scene import make_batch
    from tests.seeded import fill_by_key
    xyz = torch.from_numpy(make_batch([0, 1], 8192)).to(DEV)
    sa = fill_by_key(pm.PointnetSAModuleVotes(npoint=1024, radius=0.06, nsample=32, mlp=[16, 64, 64, 128], use_xyz=True,
                                              normalize_xyz=True), seed=2).to(DEV).train()
    torch.manual_seed(1)
    feat0 = torch.randn(2, 16, 8192, device=DEV)
    wg = fill_by_key(GraspWidthGrouping(64, 3, 0.06, -0.02, [0.01, 0.02, 0.03, 0.04]), seed=3).to(DEV).train()
    seeds = xyz[:, :256].contiguous()
    rot = torch.from_numpy(golden.load("g9_views")["rot"])[:256].unsqueeze(0).repeat(2, 1, 1, 1).contiguous().to(DEV)
    idx = fused_ops.cylinder_query_multi(xyz, seeds, rot, [0.06], -0.02, [0.01, 0.02, 0.03, 0.04], 64)
    rows = fused_mlp.cylinder_rows(idx, xyz, seeds, rot)[0]
    res = {}
    for mode in ("f32", "bf16"):
        fused_mlp.set_precision(mode)
        try:
            m1, m2 = copy.deepcopy(sa), copy.deepcopy(wg)
            f = feat0.clone().requires_grad_(True)
            _, o1, inds = m1(xyz, f)
            o2 = m2(seeds, xyz, rot, rows=rows)
            torch.manual_seed(5)
            ((o1 * torch.randn(o1.shape, device=o1.device)).sum() + (o2 * torch.randn(o2.shape, device=o2.device)).sum()).backward()
            res[mode] = (o1.detach(), o2.detach(), inds, f.grad.clone(),
                         [p.grad.clone() for p in list(m1.parameters()) + list(m2.parameters())])
        finally:
            fused_mlp.set_precision("f32")
    a, b = res["bf16"], res["f32"]
    assert torch.equal(a[2], b[2])
    assert _rel(a[0], b[0].double()) < 2e-2 and _rel(a[1], b[1].double()) < 2e-2
    assert _rel(a[0], b[0].double()) > 1e-5  # the mode really changed the arithmetic
    # gradients: bf16 noise (4e-3 per activation) re-routes max-pool / ReLU decisions, so they agree to ~0.1 only
    # (measured 0.13 on the input gradient); the tight statement about the kernels is the rounded-operand test above
    num = sum(float((x - y).norm()) ** 2 for x, y in zip(a[4], b[4])) ** 0.5
    den = sum(float(y.norm()) ** 2 for y in b[4]) ** 0.5
    print("bf16 vs fp32: fwd %.2e %.2e, dinput %.2e, dparams %.2e" % (_rel(a[0], b[0].double()), _rel(a[1], b[1].double()),
                                                                    _rel(a[3], b[3].double()), num / den))
    assert _rel(a[3], b[3].double()) < 0.3
    assert num / den < 0.3, num / den


def test_train_step_bf16_at_toy_size():
    from tests.test_model_cpu import _tiny_net
    from graspbalance_amd import fused_mlp
    from graspbalance_amd.synthetic import make_training_batch
    from graspbalance_amd.train import Trainer
    batch = make_training_batch(range(2), num_point=3000, num_objects=2, grasp_points_per_object=20, num_view=30,
                                device=DEV)
    try:
        tr = Trainer(DEV, num_view=30, model=_tiny_net(), steps_per_epoch=10, max_epoch=2, mlp_precision="bf16")
        losses = [float(tr.train_step(batch).detach()) for _ in range(3)]
        assert tr.mlp_precision == "bf16" and fused_mlp.get_precision() == "f32"   # the trainer's, not the thread's
    finally:
        fused_mlp.set_precision("f32")
    ref = Trainer(DEV, num_view=30, model=_tiny_net(), steps_per_epoch=10, max_epoch=2)
    l0 = float(ref.train_step(batch).detach())
    assert all(l == l for l in losses) and abs(losses[0] - l0) < 0.05 * abs(l0), (losses, l0)
    assert losses[0] != l0  # bf16 rounding was really applied inside the trainer's steps
