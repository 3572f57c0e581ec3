"""GPU parity: libgraspbal_hip.so (through the pointnet2._ext / pointnet2_batch_cuda shims, i.e.
through the C-ABI) against the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and at full size through size-independent properties.

Bar: bit-exact for every index output and for pure copies / un-fused fp32 arithmetic; scatter-add
gradients (atomic order is undefined, as in the reference) within 1e-5 relative.
"""
import numpy as np
import pytest
import torch

from graspbalance_amd.scene import make_scene, make_batch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ext():
    from graspbalance_amd.pointnet2 import _ext
    return _ext


@pytest.fixture(scope="module")
def pb():
    from graspbalance_amd import pointnet2_batch_cuda
    return pointnet2_batch_cuda


def _rot(golden, m, B):
    r = torch.from_numpy(golden.load("g9_views")["rot"])
    reps = (m + r.shape[0] - 1) // r.shape[0]
    return r.repeat(reps, 1, 1)[:m].unsqueeze(0).repeat(B, 1, 1, 1).contiguous()


# ------------------------------------------- FPS ----------------------------------------------
@pytest.mark.parametrize("B,N,m", [(2, 4096, 1024), (1, 20000, 1024), (3, 2048, 1024), (2, 1024, 512),
                                   (2, 512, 256), (2, 300, 100), (1, 77, 77), (2, 9, 4), (1, 1, 1),
                                   (1, 5000, 64), (2, 24576, 33), (1, 30000, 40),
                                   (1, 64512, 130), (1, 65000, 130), (1, 70000, 129)])  # around the pruned kernel's limit
def test_fps_matches_oracle_pn(ext, orc, B, N, m):
    torch.manual_seed(N + m)
    xyz = torch.rand(B, N, 3) + 0.1
    got = ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    want = orc.furthest_point_sampling(xyz, m, orc.FPS_SKIP_NEAR_ORIGIN | orc.FPS_TIE_TREE512)
    assert got.dtype == torch.int32 and torch.equal(got, want)


def test_fps_golden_c1_and_scene(ext, golden):
    torch.manual_seed(0)
    xyz = torch.rand(2, 4096, 3)
    mag = (xyz * xyz).sum(-1)
    xyz_dev = xyz.to(DEV)
    from graspbalance_amd import _lib
    idx = torch.zeros(2, 1024, dtype=torch.int32, device=DEV)
    # no skip (C1 has near-origin points, the fallback has no skip rule), every tie mode
    for tie in (_lib.FPS_TIE_LOWEST, _lib.FPS_TIE_TREE512, _lib.FPS_TIE_TREE1024):
        _lib.check(_lib.lib().gb_fps(_lib.ptr(xyz_dev), None, _lib.ptr(idx), 2, 4096, 1024, tie, None), "gb_fps")
        torch.cuda.synchronize()
        assert torch.equal(idx.cpu(), torch.from_numpy(golden.load("g1_fps_c1")["fps"]))
    assert bool((mag <= 1e-3).any())
    cloud = torch.from_numpy(make_scene(0, 20000))[None].to(DEV)
    idx = torch.zeros(1, 1024, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().gb_fps(_lib.ptr(cloud), None, _lib.ptr(idx), 1, 20000, 1024, _lib.FPS_TIE_LOWEST, None), "gb_fps")
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu(), torch.from_numpy(golden.load("g3_scene0")["fps"]))


@pytest.mark.parametrize("tie", ["lowest", "tree512", "tree1024"])
@pytest.mark.parametrize("skip", [0, 1])
def test_fps_ties_duplicates_and_skip(orc, tie, skip):
    """Heavy exact ties (integer lattice + duplicated points) and near-origin points: every rule."""
    from graspbalance_amd import _lib
    g = torch.Generator().manual_seed(7)
    for (B, N, m) in [(2, 700, 200), (1, 3000, 300), (1, 6000, 128), (2, 64, 64)]:
        xyz = torch.randint(0, 6, (B, N, 3), generator=g).float() * 0.25
        third = N // 3
        xyz[:, N - third:] = xyz[:, :third]  # exact duplicates
        flags = {"lowest": _lib.FPS_TIE_LOWEST, "tree512": _lib.FPS_TIE_TREE512, "tree1024": _lib.FPS_TIE_TREE1024}[tie]
        flags |= skip
        dev = xyz.to(DEV)
        idx = torch.zeros(B, m, dtype=torch.int32, device=DEV)
        temp = torch.full((B, N), 1e10, device=DEV)
        _lib.check(_lib.lib().gb_fps(_lib.ptr(dev), _lib.ptr(temp), _lib.ptr(idx), B, N, m, flags, None), "gb_fps")
        torch.cuda.synchronize()
        temp_o = torch.full((B, N), 1e10)
        want = orc.furthest_point_sampling(xyz, m, flags, temp=temp_o)
        assert torch.equal(idx.cpu(), want), (tie, skip, B, N, m)
        assert torch.equal(temp.cpu(), temp_o), "running min-distance state differs"


def test_fps_pb_wrapper_and_full_size_properties(pb, orc):
    # PB-ext: caller-allocated temp/idx, no skip, tree1024
    xyz = torch.from_numpy(make_batch([0, 1, 2, 3], 20000))
    dev = xyz.to(DEV)
    temp = torch.full((4, 20000), 1e10, device=DEV)
    idx = torch.zeros(4, 2048, dtype=torch.int32, device=DEV)
    assert pb.furthest_point_sampling_wrapper(4, 20000, 2048, dev, temp, idx) == 1
    idx = idx.cpu()
    want = orc.furthest_point_sampling(xyz, 2048, orc.FPS_TIE_TREE1024)
    assert torch.equal(idx, want)
    # size-independent properties: starts at 0, no index repeated unless its point is a duplicate,
    # running min distance to the chosen set is non-increasing along the sample order
    assert bool((idx[:, 0] == 0).all())
    pts = torch.gather(xyz, 1, idx.long()[:, :, None].expand(-1, -1, 3))
    for b in range(4):
        assert len(set(idx[b].tolist())) == 2048
        d = torch.cdist(pts[b].double(), pts[b].double())
        gaps = torch.stack([d[j, :j].min() for j in range(1, 2048)])
        assert bool((gaps[1:] <= gaps[:-1] + 1e-9).all())


# --------------------------------------- ball / cylinder ---------------------------------------
@pytest.mark.parametrize("B,N,m,r,ns", [(2, 4096, 512, 0.1, 32), (2, 4096, 512, 0.04, 32), (2, 4096, 512, 0.2, 64),
                                        (1, 20000, 1024, 0.04, 32), (3, 1000, 37, 0.15, 5), (1, 50, 50, 10.0, 64),
                                        (2, 130, 3, 0.3, 200), (1, 64, 1, 0.0, 4), (4, 20000, 2048, 0.04, 64),
                                        (8, 5000, 2048, 0.1, 16)])
def test_ball_query_matches_oracle(ext, pb, orc, B, N, m, r, ns):
    torch.manual_seed(N * 3 + m)
    if N == 20000:
        xyz = torch.from_numpy(make_batch(range(B), N))
    else:
        xyz = torch.rand(B, N, 3)
    new_xyz = xyz[:, torch.randperm(N)[:m]].contiguous()
    want, scanned = orc.ball_query(new_xyz, xyz, r, ns, return_scanned=True)
    got = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
    assert torch.equal(got, want)
    idx = torch.full((B, m, ns), -7, dtype=torch.int32, device=DEV)  # PB: caller allocated, garbage-filled
    assert pb.ball_query_wrapper(B, N, m, r, ns, new_xyz.to(DEV), xyz.to(DEV), idx) == 1
    assert torch.equal(idx.cpu(), want)
    # scanned-pair counts (roofline accounting) through the raw C-ABI
    from graspbalance_amd import _lib
    sc = torch.zeros(B, m, dtype=torch.int32, device=DEV)
    idx2 = torch.zeros(B, m, ns, dtype=torch.int32, device=DEV)
    nx, x = new_xyz.to(DEV), xyz.to(DEV)
    _lib.check(_lib.lib().gb_ball_query(_lib.ptr(nx), _lib.ptr(x), _lib.ptr(idx2), _lib.ptr(sc), B, N, m, r, ns, None), "bq")
    torch.cuda.synchronize()
    assert torch.equal(idx2.cpu(), want) and torch.equal(sc.cpu(), scanned)


def test_ball_query_golden(ext, golden):
    torch.manual_seed(0)
    xyz = torch.rand(2, 4096, 3)
    fps = torch.from_numpy(golden.load("g1_fps_c1")["fps"]).long()
    new_xyz = torch.gather(xyz, 1, fps[:, :512, None].expand(-1, -1, 3)).contiguous()
    for (r, ns) in [(0.1, 32), (0.04, 32), (0.2, 64)]:
        want = torch.from_numpy(golden.load("g2_ball_c1")["idx_r%g_ns%d" % (r, ns)]).clone()
        want[want == 4096] = 0
        assert torch.equal(ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu(), want)


@pytest.mark.parametrize("B,N,m,ns", [(2, 3000, 64, 16), (1, 20000, 1024, 64), (2, 500, 300, 7)])
def test_cylinder_query_matches_oracle(ext, orc, golden, B, N, m, ns):
    torch.manual_seed(N + 11)
    xyz = torch.from_numpy(make_batch(range(B), N)) if N == 20000 else torch.rand(B, N, 3) * 0.3
    new_xyz = xyz[:, :m].contiguous()
    rot = _rot(golden, m, B)
    for (r, hmin, hmax) in [(0.05, -0.02, 0.04), (0.02, -0.02, 0.01), (0.08, -0.02, 0.03)]:
        want = orc.cylinder_query(new_xyz, xyz, rot.view(B, m, 9), r, hmin, hmax, ns)
        got = ext.cylinder_query(new_xyz.to(DEV), xyz.to(DEV), rot.view(B, m, 9).to(DEV), r, hmin, hmax, ns).cpu()
        assert torch.equal(got, want)


def test_cylinder_query_multi_equals_16_single_queries(orc, golden):
    from graspbalance_amd import _lib
    import ctypes
    B, N, m, ns = 2, 20000, 1024, 64
    xyz = torch.from_numpy(make_batch([5, 6], N))
    new_xyz = xyz[:, :m].contiguous()
    rot = _rot(golden, m, B).view(B, m, 9)
    radii = [0.02, 0.04, 0.06, 0.08]
    hmaxs = [0.01, 0.02, 0.03, 0.04]
    out = torch.full((4, 4, B, m, ns), -1, dtype=torch.int32, device=DEV)
    nx, x, rt = new_xyz.to(DEV), xyz.to(DEV), rot.to(DEV)
    ra = (ctypes.c_float * 4)(*radii)
    ha = (ctypes.c_float * 4)(*hmaxs)
    _lib.check(_lib.lib().gb_cylinder_query_multi(_lib.ptr(nx), _lib.ptr(x), _lib.ptr(rt), _lib.ptr(out), B, N, m,
                                                  ctypes.cast(ra, ctypes.c_void_p), 4, -0.02,
                                                  ctypes.cast(ha, ctypes.c_void_p), 4, ns, None), "multi")
    torch.cuda.synchronize()
    out = out.cpu()
    for ir, r in enumerate(radii):
        for ih, h in enumerate(hmaxs):
            want = orc.cylinder_query(new_xyz, xyz, rot, r, -0.02, h, ns)
            assert torch.equal(out[ir, ih], want), (r, h)


# ------------------------------------- gather / group ------------------------------------------
@pytest.mark.parametrize("B,C,N,m,ns", [(2, 3, 4096, 512, 32), (2, 128, 2048, 2048, 64), (1, 5, 33, 7, 3), (3, 1, 10, 4, 1)])
def test_group_and_gather_fwd_bwd(ext, pb, orc, B, C, N, m, ns):
    torch.manual_seed(C + N)
    pts = torch.randn(B, C, N)
    idx = torch.randint(0, N, (B, m, ns), dtype=torch.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]  # padded rows -> heavy duplicate scatter targets
    got = ext.group_points(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, orc.group_points(pts, idx))
    out = torch.empty(B, C, m, ns, device=DEV)
    pb.group_points_wrapper(B, C, N, m, ns, pts.to(DEV), idx.to(DEV), out)
    assert torch.equal(out.cpu(), got)
    g = torch.randn(B, C, m, ns)
    gg = ext.group_points_grad(g.to(DEV), idx.to(DEV), N).cpu()
    ref = orc.group_points_grad(g.double().float(), idx, N)
    ref64 = torch.zeros(B, C, N, dtype=torch.float64).scatter_add_(
        2, idx.long().view(B, 1, -1).expand(-1, C, -1), g.double().view(B, C, -1))
    scale = ref64.abs().max().item() + 1e-12
    assert float((gg.double() - ref64).abs().max()) / scale < 1e-5
    assert float((ref.double() - ref64).abs().max()) / scale < 1e-5
    acc = torch.zeros(B, C, N, device=DEV)
    pb.group_points_grad_wrapper(B, C, N, m, ns, g.to(DEV), idx.to(DEV), acc)
    assert float((acc.cpu().double() - ref64).abs().max()) / scale < 1e-5
    # gather
    idx1 = idx[:, :, 0].contiguous()
    ga = ext.gather_points(pts.to(DEV), idx1.to(DEV)).cpu()
    assert torch.equal(ga, orc.gather_points(pts, idx1))
    g1 = torch.randn(B, C, m)
    gag = ext.gather_points_grad(g1.to(DEV), idx1.to(DEV), N).cpu()
    r1 = torch.zeros(B, C, N, dtype=torch.float64).scatter_add_(2, idx1.long().view(B, 1, -1).expand(-1, C, -1), g1.double())
    assert float((gag.double() - r1).abs().max()) / (r1.abs().max().item() + 1e-12) < 1e-5


# --------------------------------- three_nn / interpolate --------------------------------------
@pytest.mark.parametrize("B,n,m,C", [(2, 1024, 512, 16), (4, 512, 256, 256), (1, 20000, 1024, 8), (2, 10, 2, 3), (1, 7, 1, 2), (1, 3000, 2500, 4)])
def test_three_nn_and_interpolate(ext, pb, orc, B, n, m, C):
    torch.manual_seed(n + m)
    unknown, known = torch.rand(B, n, 3), torch.rand(B, m, 3)
    d2, idx = ext.three_nn(unknown.to(DEV), known.to(DEV))
    wd2, widx = orc.three_nn(unknown, known)
    assert torch.equal(idx.cpu(), widx) and torch.equal(d2.cpu(), wd2)
    d2b = torch.empty(B, n, 3, device=DEV)
    idxb = torch.empty(B, n, 3, dtype=torch.int32, device=DEV)
    pb.three_nn_wrapper(B, n, m, unknown.to(DEV), known.to(DEV), d2b, idxb)
    assert torch.equal(idxb.cpu(), widx) and torch.equal(d2b.cpu(), wd2)
    if m < 3:
        return
    feats = torch.randn(B, C, m)
    w = torch.rand(B, n, 3)
    w = w / w.sum(-1, keepdim=True)
    out = ext.three_interpolate(feats.to(DEV), idx, w.to(DEV)).cpu()
    assert torch.equal(out, orc.three_interpolate(feats, widx, w))  # same un-fused evaluation order
    g = torch.randn(B, C, n)
    gp = ext.three_interpolate_grad(g.to(DEV), idx, w.to(DEV), m).cpu()
    ref = orc.three_interpolate_grad(g, widx, w, m)
    assert float((gp - ref).abs().max()) / (ref.abs().max().item() + 1e-12) < 1e-5


def test_three_nn_golden(ext, golden):
    g = golden.load("g5_three_nn")
    torch.manual_seed(5)
    unknown, known = torch.rand(2, 1024, 3), torch.rand(2, 512, 3)
    d2, idx = ext.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(idx.cpu(), torch.from_numpy(g["idx"])) and torch.equal(d2.cpu(), torch.from_numpy(g["dist2"]))


def test_interpolation_weights_equal_the_reference_composition(ext):
    """gb_interp_weights (pointnet2_utils.three_nn_weights): the PointnetFPModule weights formed from three_nn's squared
    distances in one pass == sqrt, + 1e-8, 1/x, sum over the three, divide as IEEE float32 operations in that order
    (the CPU evaluation of the reference lines, pointnet2_modules.py:260-263), bit for bit; duplicates (distance 0) kept."""
    from graspbalance_amd import pointnet2_utils
    torch.manual_seed(11)
    unknown, known = torch.rand(2, 3000, 3), torch.rand(2, 700, 3)
    unknown[0, :50] = known[0, :50]          # exact hits: d = 0, r = 1e8
    w, idx = pointnet2_utils.three_nn_weights(unknown.to(DEV), known.to(DEV))
    d2, idx_ref = ext.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(idx, idx_ref)
    def compose(d2):
        recip = 1.0 / (torch.sqrt(d2) + 1e-8)
        return recip / ((recip[..., 0] + recip[..., 1]) + recip[..., 2]).unsqueeze(-1)
    assert torch.equal(w, compose(d2))                                   # the same launches' worth of torch on the GPU
    cpu = compose(d2.cpu())
    print("interp weights vs CPU torch: max abs diff %.2e" % float((w.cpu() - cpu).abs().max()))
    assert torch.allclose(w.cpu(), cpu, rtol=3e-7, atol=1e-12)
    assert float((w.sum(-1) - 1).abs().max()) < 1e-6


# ----------------------------------------- knn1 ------------------------------------------------
def test_knn1(orc, golden):
    from graspbalance_amd import _lib
    g = golden.load("g8_knn")
    torch.manual_seed(8)
    ref, query = torch.rand(1, 3, 300), torch.rand(1, 3, 300)
    ref2, query2 = torch.rand(2, 3, 700), torch.rand(2, 3, 1024)
    for (r, q, want) in [(ref, query, g["inds"]), (ref2, query2, g["inds2"])]:
        out = torch.zeros(r.shape[0], 1, q.shape[2], dtype=torch.int64, device=DEV)
        rd, qd = r.to(DEV), q.to(DEV)
        _lib.check(_lib.lib().gb_knn1(_lib.ptr(rd), _lib.ptr(qd), _lib.ptr(out), r.shape[0], 3, r.shape[2], q.shape[2], None), "knn")
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), torch.from_numpy(want))
        assert torch.equal(out.cpu(), orc.knn1(r, q))


# --------------------------------- error behaviour of the shims --------------------------------
def test_error_contract(ext):
    x = torch.rand(1, 8, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(x, 2)
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        ext.furthest_point_sampling(x.double().to(DEV), 2)
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        ext.gather_points(torch.rand(1, 8, 3, device=DEV).transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32, device=DEV))
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        ext.gather_points(torch.rand(1, 3, 8, device=DEV), torch.zeros(1, 2, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ext.gather_points(torch.rand(1, 3, 8, device=DEV), torch.zeros(1, 2, dtype=torch.int32))


def test_label_finish_matches_torch_composition():
    """gb_label_finish against the reference's element-wise formulation (label_generation.py:112-116)."""
    import torch
    from graspbalance_amd import _lib
    from graspbalance_amd.loss_utils import GRASP_MAX_WIDTH
    torch.manual_seed(0)
    B, Ns, V, A, D = 2, 37, 30, 12, 4
    labels = torch.rand(B, Ns, V, A, D, device="cuda:0") * 1.2 - 0.2
    labels[torch.rand_like(labels) < 0.3] = -1.0
    labels[torch.rand_like(labels) < 0.05] = 0.0
    offsets = torch.rand(B, Ns, V, A, D, 3, device="cuda:0") * 0.15
    u_max = labels.max()
    out = torch.empty_like(labels)
    vs = torch.empty(B, Ns, V, device="cuda:0")
    va = torch.empty(B, Ns, V, dtype=torch.int32, device="cuda:0")
    _lib.check(_lib.lib().gb_label_finish(_lib.ptr(labels), _lib.ptr(offsets), None, _lib.ptr(u_max),
                                          float(GRASP_MAX_WIDTH), _lib.ptr(out), _lib.ptr(vs), _lib.ptr(va), B * Ns * V,
                                          A * D, None), "label_finish")
    # the same from a contiguous width column (what gb_label_gather's out_col provides), offsets not read at all
    widths = offsets[..., 2].contiguous()
    out2, vs2, va2 = torch.empty_like(out), torch.empty_like(vs), torch.empty_like(va)
    _lib.check(_lib.lib().gb_label_finish(_lib.ptr(labels), None, _lib.ptr(widths), _lib.ptr(u_max),
                                          float(GRASP_MAX_WIDTH), _lib.ptr(out2), _lib.ptr(vs2), _lib.ptr(va2), B * Ns * V,
                                          A * D, None), "label_finish")
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.equal(vs2, vs) and torch.equal(va2, va)
    mask = (labels > 0) & (offsets[..., 2] <= GRASP_MAX_WIDTH)
    ref = labels.clone()
    ref[mask] = torch.log(u_max / ref[mask])
    ref[~mask] = 0
    ref_vs = ref.view(B, Ns, V, A * D).max(dim=-1)[0]
    assert torch.equal(out, ref)
    assert torch.equal(vs, ref_vs)
    assert torch.equal(va.long(), ref.view(B, Ns, V, A * D).argmax(dim=-1))  # first maximum, many exact ties (zeros)
    # ... which composes to the arg-max over all views of a seed (loss.py:31)
    top_view = vs.argmax(dim=2, keepdim=True)
    flat = top_view * (A * D) + torch.gather(va.long(), 2, top_view)
    assert torch.equal(flat.squeeze(2), ref.view(B, Ns, -1).argmax(dim=2))


@pytest.mark.parametrize("tie", ["lowest", "tree512", "tree1024"])
@pytest.mark.parametrize("skip", [0, 1])
def test_fps_pruned_is_the_same_sequence(orc, tie, skip):
    """gb_fps_pruned (wave-level skipping on a sorted visiting order) against the oracle: heavy exact ties
    (integer lattice + duplicated points), near-origin points, every tie rule, Morton / identity / random
    visiting orders, and the running min-distance state."""
    import torch
    from graspbalance_amd import _lib
    g = torch.Generator().manual_seed(11)
    flags = {"lowest": _lib.FPS_TIE_LOWEST, "tree512": _lib.FPS_TIE_TREE512, "tree1024": _lib.FPS_TIE_TREE1024}[tie] | skip
    cases = []
    for (B, N, m) in [(2, 9000, 300), (1, 20000, 256), (2, 4100, 128), (1, 70, 70), (1, 8200, 65), (2, 12345, 1), (1, 20480, 193)]:
        xyz = torch.randint(0, 9, (B, N, 3), generator=g).float() * 0.125
        xyz[:, N - N // 3:] = xyz[:, :N // 3]  # exact duplicates
        cases.append((xyz, m))
    cases.append((torch.from_numpy(make_batch([5, 6], 20000)), 512))  # the bench's kind of cloud
    # beyond one CU's registers: rows re-read from the sorted scratch copy (24 rows in registers, the rest in LDS)
    big = torch.randint(0, 40, (1, 30000, 3), generator=g).float() * 0.03125
    big[:, 20000:] = big[:, :10000]
    cases.append((big, 200))
    cases.append((torch.from_numpy(make_batch([9], 50000)), 384))    # BASELINE configs[4] cloud size
    cases.append((torch.from_numpy(make_batch([10], 64512)), 130))   # the largest cloud the kernel takes
    for xyz, m in cases:
        B, N = xyz.shape[:2]
        dev = xyz.to(DEV)
        scratch = torch.empty(B, N, 4, device=DEV) if N > 20480 else None
        keys = torch.empty(B, N, dtype=torch.int32, device=DEV)
        _lib.check(_lib.lib().gb_fps_morton_keys(_lib.ptr(dev), _lib.ptr(keys), B, N, None), "morton")
        perms = {"morton": torch.argsort(keys, dim=1).int(),
                 "identity": torch.arange(N, device=DEV, dtype=torch.int32).repeat(B, 1),
                 "random": torch.stack([torch.randperm(N, generator=g) for _ in range(B)]).int().to(DEV)}
        temp_o = torch.full((B, N), 1e10)
        want = orc.furthest_point_sampling(xyz, m, flags, temp=temp_o)
        # every way the library spreads a register-resident cloud over its CU (round 5: 4 / 8 / 12 / 16 waves with
        # run-time indexed row registers, and the round-4 kernel) returns the same samples; larger clouds have one form
        layouts = _lib.FPS_LAYOUT.items() if N <= 20480 else [("auto", 0), ("r4", _lib.FPS_LAYOUT["r4"])]
        for name, perm in perms.items():
            for lname, layout in layouts:
                idx = torch.full((B, m), -7, dtype=torch.int32, device=DEV)
                temp = torch.full((B, N), 1e10, device=DEV)
                _lib.check(_lib.lib().gb_fps_pruned(_lib.ptr(dev), _lib.ptr(perm.contiguous()), _lib.ptr(temp), _lib.ptr(idx),
                                                    B, N, m, flags | layout, _lib.ptr(scratch), None), "gb_fps_pruned")
                torch.cuda.synchronize()
                assert torch.equal(idx.cpu(), want), (tie, skip, name, lname, B, N, m)
                assert torch.equal(temp.cpu(), temp_o), ("running min-distance state differs", name, lname)


@pytest.mark.parametrize("tie", ["lowest", "tree512", "tree1024"])
@pytest.mark.parametrize("skip", [0, 1])
def test_fps_guarded_prefix_check(orc, tie, skip):
    """gb_fps_guarded = gb_fps in every case: inputs already in farthest-point order (the check passes and the
    sequential loop is skipped), the same with exact ties (lattice / duplicated points: the check may fail and the
    loop runs), arbitrary clouds (fails at once), near-origin points; samples and running min-distances."""
    import torch
    from graspbalance_amd import _lib
    g = torch.Generator().manual_seed(21)
    flags = {"lowest": _lib.FPS_TIE_LOWEST, "tree512": _lib.FPS_TIE_TREE512, "tree1024": _lib.FPS_TIE_TREE1024}[tie] | skip

    def fps_order(xyz, m0):  # reorder every cloud into its own farthest-point order (oracle)
        idx = orc.furthest_point_sampling(xyz, m0, flags)
        return torch.gather(xyz, 1, idx.long()[:, :, None].expand(-1, -1, 3)).contiguous()

    scene = torch.from_numpy(make_batch([7, 8], 6000))
    lattice = torch.randint(0, 7, (2, 3000, 3), generator=g).float() * 0.2
    lattice[:, 2000:] = lattice[:, :1000]
    cases = [(fps_order(scene, 2048), 1024, True), (fps_order(scene, 1024), 512, True), (fps_order(scene, 300), 300, True),
             (fps_order(lattice, 1500), 700, False), (scene[:, :3000].contiguous(), 500, False),
             (torch.rand(3, 100, 3, generator=g) * 0.02, 64, False)]
    passed = 0
    for xyz, m, expect_identity in cases:
        B, N = xyz.shape[:2]
        dev = xyz.to(DEV)
        idx = torch.full((B, m), -7, dtype=torch.int32, device=DEV)
        temp = torch.full((B, N), 1e10, device=DEV)
        ws = torch.empty(B * (m + N), device=DEV)
        ok = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        _lib.check(_lib.lib().gb_fps_guarded(_lib.ptr(dev), _lib.ptr(temp), _lib.ptr(idx), B, N, m, flags, _lib.ptr(ws),
                                             _lib._c.c_void_p(ws.data_ptr() + 4 * B * m), _lib.ptr(ok), None), "guarded")
        torch.cuda.synchronize()
        temp_o = torch.full((B, N), 1e10)
        want = orc.furthest_point_sampling(xyz, m, flags, temp=temp_o)
        assert torch.equal(idx.cpu(), want), (tie, skip, N, m)
        assert torch.equal(temp.cpu(), temp_o), "running min-distance state differs"
        if expect_identity and not skip:
            assert torch.equal(want, torch.arange(m, dtype=torch.int32).repeat(B, 1))
            assert bool((ok == 1).all()), "prefix check should have passed on farthest-point-ordered input"
        passed += int((ok == 1).sum())
    assert passed > 0


@pytest.mark.parametrize("tie", ["lowest", "tree512", "tree1024"])
def test_fps_segments_equals_per_segment_fps(orc, tie):
    """gb_fps_segments (one workgroup per segment of a packed list) == the oracle's FPS of every segment as a cloud
    of its own: segment sizes 1 .. 5000 (different reference block sizes, hence tie rules), exact ties (lattice +
    duplicates), near-origin points, more samples than points, empty sample counts."""
    import torch
    from graspbalance_amd import _lib
    g = torch.Generator().manual_seed(31)
    flags = {"lowest": _lib.FPS_TIE_LOWEST, "tree512": _lib.FPS_TIE_TREE512, "tree1024": _lib.FPS_TIE_TREE1024}[tie] | 1
    sizes = [1, 7, 64, 300, 513, 1024, 1500, 5000, 90]
    counts = [1, 7, 20, 128, 128, 0, 300, 146, 120]
    segs = []
    for n in sizes:
        pts = torch.randint(0, 6, (n, 3), generator=g).float() * 0.25 + 0.5
        if n >= 64:
            pts[n // 2:n // 2 + n // 4] = pts[:n // 4]     # duplicates
            pts[-3:] = 0.001                               # near the origin: skipped
        segs.append(pts)
    packed = torch.cat(segs, 0).contiguous()
    offs = [0]
    outs = [0]
    for n, m in zip(sizes, counts):
        offs.append(offs[-1] + n)
        outs.append(outs[-1] + m)
    dev = packed.to(DEV)
    seg_off = torch.tensor(offs, dtype=torch.int32, device=DEV)
    out_off = torch.tensor(outs, dtype=torch.int32, device=DEV)
    temp = torch.empty(offs[-1], device=DEV)
    idx = torch.full((outs[-1],), -5, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().gb_fps_segments(_lib.ptr(dev), _lib.ptr(seg_off), _lib.ptr(out_off), _lib.ptr(temp), _lib.ptr(idx),
                                          len(sizes), max(sizes), flags, None), "gb_fps_segments")
    torch.cuda.synchronize()
    got = idx.cpu()
    for s, (pts, m) in enumerate(zip(segs, counts)):
        if m == 0:
            continue
        want = orc.furthest_point_sampling(pts.unsqueeze(0).contiguous(), m, flags)[0]
        assert torch.equal(got[outs[s]:outs[s + 1]], want), (tie, s, sizes[s], m)


def test_known_answers_and_degenerate_shapes(ext, pb, orc):
    """The hand-derived known answers of tests/test_oracle_golden.py (G10: d2 == r2 boundary, empty ball -> zeros,
    first-hit padding, nsample = 1 early exit, FPS tie rules and the near-origin skip) on the HIP path itself, a
    lattice cloud where most pair distances hit r2 exactly, and the degenerate shapes (no centres, no samples, one
    point) through the raw C-ABI."""
    import torch
    from graspbalance_amd import _lib
    xyz = torch.tensor([[[0.0, 0, 1], [0.5, 0, 1], [0.25, 0, 1], [0.1, 0, 1], [3.0, 0, 1]]])
    new_xyz = torch.tensor([[[0.0, 0, 1], [10.0, 0, 1], [0.25, 0, 1]]])
    idx = ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 0.25, 4).cpu()
    assert idx[0].tolist() == [[0, 3, 0, 0], [0, 0, 0, 0], [2, 3, 2, 2]]
    assert ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), 0.25, 1).cpu()[0, :, 0].tolist() == [0, 0, 2]
    # lattice: every coordinate a multiple of 1/8, r = 0.25 -> many pairs with d2 == r2 exactly (excluded: strict <)
    g = torch.Generator().manual_seed(5)
    lat = torch.randint(0, 6, (2, 700, 3), generator=g).float() * 0.125 + 1.0
    cen = lat[:, :90].contiguous()
    for r, ns in ((0.25, 16), (0.125, 8), (0.2165063509461097, 32)):   # the last one: r2 = 3/64 up to rounding
        want = orc.ball_query(cen, lat, r, ns)
        assert torch.equal(ext.ball_query(cen.to(DEV), lat.to(DEV), r, ns).cpu(), want), (r, ns)
    rot = torch.eye(3).repeat(2, 90, 1, 1).view(2, 90, 9).contiguous()
    for radius, hmax in ((0.25, 0.125), (0.125, 0.25)):
        want = orc.cylinder_query(cen, lat, rot, radius, -0.125, hmax, 16)
        got = ext.cylinder_query(cen.to(DEV), lat.to(DEV), rot.to(DEV), radius, -0.125, hmax, 16).cpu()
        assert torch.equal(got, want), (radius, hmax)
    # FPS: square corners + duplicate; tie rules; near-origin skip
    sq = torch.tensor([[[1.0, 1, 1], [2.0, 1, 1], [1.0, 2, 1], [2.0, 2, 1], [2.0, 2, 1]]]).to(DEV)
    out = torch.zeros(1, 3, dtype=torch.int32, device=DEV)
    for flags, first_two in ((_lib.FPS_TIE_LOWEST, [0, 3]), (_lib.FPS_TIE_TREE512, [0, 4])):
        _lib.check(_lib.lib().gb_fps(_lib.ptr(sq), None, _lib.ptr(out), 1, 5, 3, flags, None), "gb_fps")
        torch.cuda.synchronize()
        assert out[0].tolist()[:2] == first_two
    og = torch.tensor([[[1.0, 0, 0], [0.0, 0, 0], [1.1, 0, 0], [0.01, 0.01, 0.01]]]).to(DEV)
    out2 = torch.zeros(1, 2, dtype=torch.int32, device=DEV)
    for flags, want2 in ((_lib.FPS_TIE_LOWEST, [0, 1]), (_lib.FPS_TIE_LOWEST | _lib.FPS_SKIP_NEAR_ORIGIN, [0, 2])):
        _lib.check(_lib.lib().gb_fps(_lib.ptr(og), None, _lib.ptr(out2), 1, 4, 2, flags, None), "gb_fps")
        torch.cuda.synchronize()
        assert out2[0].tolist() == want2
    # degenerate shapes: nothing to do is not an error and writes nothing
    L = _lib.lib()
    guard = torch.full((8,), -3, dtype=torch.int32, device=DEV)
    one = torch.tensor([[[0.5, 0.5, 0.5]]], device=DEV)
    assert L.gb_fps(_lib.ptr(one), None, _lib.ptr(guard), 0, 1, 4, _lib.FPS_TIE_LOWEST, None) == 0      # b = 0
    assert L.gb_fps(_lib.ptr(one), None, _lib.ptr(guard), 1, 1, 0, _lib.FPS_TIE_LOWEST, None) == 0      # m = 0
    assert L.gb_ball_query(_lib.ptr(one), _lib.ptr(one), _lib.ptr(guard), None, 1, 1, 0, 0.1, 4, None) == 0  # no centres
    torch.cuda.synchronize()
    assert bool((guard == -3).all())
    # one point, several samples: index 0 every time (m > n is legal in the reference's loop)
    assert L.gb_fps(_lib.ptr(one), None, _lib.ptr(guard), 1, 1, 4, _lib.FPS_TIE_TREE512, None) == 0
    torch.cuda.synchronize()
    assert guard[:4].tolist() == [0, 0, 0, 0]


def test_fps_cell_order_is_a_coherent_permutation(orc):
    """gb_fps_cell_order: every cloud's output is a permutation of its indices, grouped by grid cell (points of one
    32^3 cell are contiguous), and gb_fps_pruned on it returns the oracle's samples."""
    import torch
    from graspbalance_amd import _lib
    for (B, N, m) in [(3, 20000, 300), (1, 777, 128), (2, 50000, 150)]:
        xyz = torch.from_numpy(make_batch(list(range(20, 20 + B)), N))
        xyz[0, : N // 10] = xyz[0, 0]  # many points in one cell
        dev = xyz.to(DEV)
        perm = torch.full((B, N), -1, dtype=torch.int32, device=DEV)
        _lib.check(_lib.lib().gb_fps_cell_order(_lib.ptr(dev), _lib.ptr(perm), B, N, None), "cell_order")
        torch.cuda.synchronize()
        p = perm.cpu().long()
        assert torch.equal(torch.sort(p, dim=1)[0], torch.arange(N).repeat(B, 1))
        lo, hi = xyz.min(1, keepdim=True)[0], xyz.max(1, keepdim=True)[0]
        q = ((xyz - lo) * (31.0 / (hi - lo))).clamp(0, 31).floor().long()
        cell = q[..., 0] * 1024 + q[..., 1] * 32 + q[..., 2]           # any injective cell id
        walked = torch.gather(cell, 1, p)
        changes = (walked[:, 1:] != walked[:, :-1]).sum(1) + 1
        distinct = torch.tensor([len(torch.unique(c)) for c in cell])
        # (a point on a cell boundary may round differently here and in the kernel: allow a few extra runs)
        assert bool(((changes >= distinct) & (changes <= distinct + 3)).all()), (changes, distinct)
        flags = _lib.FPS_TIE_TREE512 | _lib.FPS_SKIP_NEAR_ORIGIN
        idx = torch.zeros(B, m, dtype=torch.int32, device=DEV)
        scratch = torch.empty(B, N, 4, device=DEV) if N > 20480 else None
        _lib.check(_lib.lib().gb_fps_pruned(_lib.ptr(dev), _lib.ptr(perm), None, _lib.ptr(idx), B, N, m, flags,
                                            _lib.ptr(scratch), None), "gb_fps_pruned")
        torch.cuda.synchronize()
        assert torch.equal(idx.cpu(), orc.furthest_point_sampling(xyz, m, flags))


def test_fps_row_order_is_a_permutation_with_compact_rows(orc):
    """gb_fps_row_order (round 5): every cloud's output is a permutation of its indices; its structure is the one the
    kernel documents - the cloud sorted along its widest axis and cut into K slabs at multiples of 64 points (so slab
    membership is monotone in that coordinate up to one bin width), every slab sorted along ITS widest axis (monotone up
    to a bin) - rows of 64 consecutive points are at least twice as compact as in gb_fps_cell_order's order, and
    gb_fps_pruned on it returns the oracle's samples.  Degenerate inputs: all points equal, n < 64, n = 1."""
    import torch
    from graspbalance_amd import _lib
    cases = [(torch.from_numpy(make_batch([30, 31, 32], 20000)), 300, True), (torch.from_numpy(make_batch([33], 777)), 128, False),
             (torch.from_numpy(make_batch([34, 35], 24576)), 100, True), (torch.zeros(2, 5000, 3) + 0.25, 64, False),
             (torch.rand(1, 50, 3, generator=torch.Generator().manual_seed(3)), 50, False), (torch.rand(1, 1, 3) + 1, 1, False),
             (torch.from_numpy(make_batch([36], 50000)), 150, False),   # > 24576 points without a workspace: the cell order
             (torch.from_numpy(make_batch([37, 38], 50000)), 150, True),  # ... with one (the sampling's scratch): compact rows
             (torch.from_numpy(make_batch([39], 65536)), 130, True)]
    for case, (xyz, m, check_rows) in enumerate(cases):
        xyz = xyz.clone()
        B, N = xyz.shape[:2]
        dev = xyz.to(DEV)
        perm = torch.full((B, N), -1, dtype=torch.int32, device=DEV)
        if N > 24576 and check_rows:
            ws = torch.empty(B, N, 4, device=DEV)
            _lib.check(_lib.lib().gb_fps_row_order_ws(_lib.ptr(dev), _lib.ptr(perm), _lib.ptr(ws), B, N, None), "row_order_ws")
        else:
            _lib.check(_lib.lib().gb_fps_row_order(_lib.ptr(dev), _lib.ptr(perm), B, N, None), "row_order")
        torch.cuda.synchronize()
        p = perm.cpu().long()
        assert torch.equal(torch.sort(p, dim=1)[0], torch.arange(N).repeat(B, 1)), N
        walked = torch.gather(xyz, 1, p[:, :, None].expand(-1, -1, 3))
        if check_rows:
            nrow = (N + 63) // 64
            K = min(int(round(nrow ** 0.5)), 32 if N > 24576 else 20)
            ext = xyz.max(1)[0] - xyz.min(1)[0]
            for b in range(B):
                a1 = int(ext[b].argmax())
                tol1 = float(ext[b, a1]) / 4095 * 1.01
                bounds = [(i * nrow) // K * 64 for i in range(K)] + [N]
                prev_hi = -float("inf")
                for i in range(K):
                    slab = walked[b, bounds[i]:bounds[i + 1]]
                    assert float(slab[:, a1].min()) >= prev_hi - tol1, (N, b, i)   # slabs are ordered along the first axis
                    prev_hi = float(slab[:, a1].max())
                    e2 = slab.max(0)[0] - slab.min(0)[0]
                    a2 = int(e2.argmax())
                    tol2 = float(e2[a2]) / 1023 * 1.01
                    c = slab[:, a2]
                    assert float((c[:-1] - c[1:]).max()) <= tol2, (N, b, i)        # ... and sorted along their own
            # the point of it: tighter rows than the grid-cell order (sum of the rows' box diagonals)
            pc = torch.empty_like(perm)
            _lib.check(_lib.lib().gb_fps_cell_order(_lib.ptr(dev), _lib.ptr(pc), B, N, None), "cell_order")
            torch.cuda.synchronize()

            def diag_sum(pp):
                w = torch.gather(xyz, 1, pp.cpu().long()[:, :, None].expand(-1, -1, 3))[:, :N // 64 * 64].view(B, -1, 64, 3)
                return float((w.max(2)[0] - w.min(2)[0]).norm(dim=-1).sum())
            # (20 000 points: 0.57; at 50 000+ the 32^3 cells are finer relative to a row and the margin is smaller: 0.80)
            assert diag_sum(perm) < (0.7 if N <= 24576 else 0.9) * diag_sum(pc), (diag_sum(perm), diag_sum(pc))
        flags = _lib.FPS_TIE_TREE512 | _lib.FPS_SKIP_NEAR_ORIGIN
        idx = torch.zeros(B, m, dtype=torch.int32, device=DEV)
        scratch = torch.empty(B, N, 4, device=DEV) if N > 20480 else None
        _lib.check(_lib.lib().gb_fps_pruned(_lib.ptr(dev), _lib.ptr(perm), None, _lib.ptr(idx), B, N, m, flags,
                                            _lib.ptr(scratch), None), "gb_fps_pruned")
        torch.cuda.synchronize()
        assert torch.equal(idx.cpu(), orc.furthest_point_sampling(xyz, m, flags)), N


@pytest.mark.parametrize("k", [1, 2, 3, 5, 8, 16])
def test_knn_any_k_equals_oracle_and_the_reference_build(orc, k):
    """gb_knn (KNN/Pytorch_CUDA_KNN/knn.h:11-59 for k <= 16; the model itself only uses k = 1): indices of the k nearest
    references, nearest first, equal distances in index order - against the C oracle on random clouds, on a lattice full
    of exact ties and on duplicated points, in 3 and 5 dimensions; and, where oracle/_ref/knn_ref.so exists (the
    reference's own vision.cpp + cpu/knn_cpu.cpp compiled by oracle/build_ref.py), against the reference itself."""
    from graspbalance_amd.knn_modules import myknn
    g = torch.Generator().manual_seed(31 + k)
    cases = [(torch.rand(2, 3, 300, generator=g), torch.rand(2, 3, 97, generator=g)),
             (torch.randint(0, 5, (1, 3, 400), generator=g).float() * 0.25, torch.randint(0, 5, (1, 3, 64), generator=g).float() * 0.25),
             (torch.rand(1, 5, 1500, generator=g), torch.rand(1, 5, 300, generator=g)),
             (torch.rand(1, 3, 40, generator=g).repeat(1, 1, 3), torch.rand(1, 3, 33, generator=g))]
    try:
        from oracle.build_ref import load_knn_ref
        ref_mod = load_knn_ref()
    except Exception:
        ref_mod = None
    for ref, query in cases:
        if k > ref.shape[2]:
            continue
        got = myknn(ref.to(DEV), query.to(DEV), k=k).cpu()
        want = orc.knn(ref, query, k)
        assert got.shape == (ref.shape[0], k, query.shape[2]) and got.dtype == torch.int64
        assert torch.equal(got, want), (k, tuple(ref.shape))
        if ref_mod is not None and ref.shape[2] <= 400:   # (the reference sorts every row completely: O(nref^2) per query)
            idx = torch.empty(ref.shape[0], k, query.shape[2], dtype=torch.long)
            ref_mod.knn(ref.contiguous(), query.contiguous(), idx)
            assert torch.equal(idx, want), ("reference build", k, tuple(ref.shape))
