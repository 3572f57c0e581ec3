"""Dense-torch restatement of the reference's CPU-runnable fallback algorithm (TrainModel/pointnet2_util.py):
a full distance matrix per call, sort-based neighbour selection, an iterative arg-max FPS on whole tensors.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): the stand-in for "the reference's CPU fallback path" in
bench.py's ``cpu_baseline_dense`` leg (leg (i)) and a second, independent checker in the tests.
The algorithm is the fallback's; the three documented divergences from the CUDA kernels are resolved the kernels'
way so that outputs are index-identical to the HIP path and the C oracle:

  * FPS starts at index 0 (the fallback draws a random start, pointnet2_util.py:34) and, like the CUDA kernel, the
    optional skip of points with |p|^2 <= 1e-3 (sampling_gpu.cu:105-106);
  * ball query accepts d^2 <  r^2 (ball_query_gpu.cu:33; the fallback keeps d^2 == r^2, :51) and marks an empty ball
    with index 0 (the fallback writes N, :51-55).
"""
import torch


def square_distance(src, dst):
    """(B,S,3),(B,N,3) -> (B,S,N): ((dx*dx + dy*dy) + dz*dz), the fallback's broadcast-subtract-square-sum
    (pointnet2_util.py:18-19) with the summation order every side of this repo pins."""
    d = src[:, :, None, :] - dst[:, None, :, :]
    d = d * d
    return (d[..., 0] + d[..., 1]) + d[..., 2]


def farthest_point_sample(xyz, npoint, skip_near_origin=True):
    """pointnet2_util.py:29-42: running min-distance vector + arg-max per iteration, whole-batch tensor ops.
    Ties go to the lowest index (torch.max) - the fallback's rule, equal to the kernel's whenever no exact tie occurs."""
    B, N, _ = xyz.shape
    picks = torch.zeros(B, npoint, dtype=torch.long)
    running = torch.full((B, N), 1e10)
    eligible = (xyz * xyz).sum(-1) > 1e-3 if skip_near_origin else None
    current = torch.zeros(B, dtype=torch.long)
    rows = torch.arange(B)
    for i in range(npoint):
        picks[:, i] = current
        d = xyz - xyz[rows, current, :].view(B, 1, 3)
        d = d * d
        running = torch.minimum(running, (d[..., 0] + d[..., 1]) + d[..., 2])
        score = running if eligible is None else torch.where(eligible, running, torch.full_like(running, -1.0))
        current = torch.max(score, -1)[1]
    return picks.to(torch.int32)


def query_ball_point(radius, nsample, xyz, new_xyz):
    """pointnet2_util.py:45-57: index grid, mask by the distance matrix, sort, keep the first nsample, pad with the
    first hit.  -> (B,S,nsample) int32."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    grid = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    grid[~(square_distance(new_xyz, xyz) < radius * radius)] = N
    grid = grid.sort(dim=-1)[0][:, :, :nsample]
    first = grid[:, :, :1].expand(-1, -1, nsample)
    grid = torch.where(grid == N, first, grid)
    return torch.where(grid == N, torch.zeros_like(grid), grid).to(torch.int32)


def sa_layer_forward(xyz, npoint, radius, nsample, weights, normalize_xyz=True, eps=1e-5):
    """BASELINE configs[1]: one set-abstraction layer forward in train mode (FPS -> ball query -> group ->
    [1x1 conv (no bias) -> BatchNorm (batch statistics) -> ReLU] x L -> max over nsample), dense torch throughout
    (sample_and_group :59-88 + the conv/BN/ReLU/max of PointNetSetAbstraction :100-128).
    weights: list of (W (Cout,Cin), gamma, beta).  -> (inds (B,npoint) i32, idx (B,npoint,ns) i32, features (B,C,npoint))."""
    B = xyz.shape[0]
    inds = farthest_point_sample(xyz, npoint)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped = torch.gather(xyz, 1, idx.long().reshape(B, -1, 1).expand(-1, -1, 3)).view(B, npoint, nsample, 3)
    x = grouped - new_xyz.unsqueeze(2)
    if normalize_xyz:
        x = x / radius
    x = x.reshape(-1, 3)
    for W, gamma, beta in weights:
        y = x @ W.t()
        mean, var = y.mean(0), y.var(0, unbiased=False)
        x = torch.relu((y - mean) / torch.sqrt(var + eps) * gamma + beta)
    feats = x.view(B, npoint, nsample, -1).max(2)[0]
    return inds, idx, feats.transpose(1, 2).contiguous()


def cylinder_query(radius, hmin, hmax, nsample, xyz, new_xyz, rot):
    """A second, independent restatement of cylinder_query_gpu.cu:20-78 as whole-tensor operations (the C oracle loops;
    this one masks and sorts like the fallback's query_ball_point): a point belongs to centre j's cylinder when, in the
    frame rot[j] (columns = gripper axes), its offset has x in (hmin, hmax) and y^2 + z^2 < radius^2; the first
    `nsample` members in index order, padded with the first, zeros when there is none.  Products and sums are single
    rounded operations in the kernel's left-to-right order.  rot (B,S,9) or (B,S,3,3) -> (B,S,nsample) int32."""
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    r = rot.reshape(B, S, 9)
    d = xyz[:, None, :, :] - new_xyz[:, :, None, :]                       # (B,S,N,3)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    c = lambda i: r[:, :, i].unsqueeze(-1)
    x_rot = (c(0) * x + c(3) * y) + c(6) * z
    y_rot = (c(1) * x + c(4) * y) + c(7) * z
    z_rot = (c(2) * x + c(5) * y) + c(8) * z
    inside = ((y_rot * y_rot + z_rot * z_rot) < radius * radius) & (x_rot > hmin) & (x_rot < hmax)
    grid = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    grid[~inside] = N
    grid = grid.sort(dim=-1)[0][:, :, :nsample]
    if grid.shape[-1] < nsample:
        grid = torch.cat([grid, grid.new_full((B, S, nsample - grid.shape[-1]), N)], dim=-1)
    first = grid[:, :, :1].expand(-1, -1, nsample)
    grid = torch.where(grid == N, first, grid)
    return torch.where(grid == N, torch.zeros_like(grid), grid).to(torch.int32)
