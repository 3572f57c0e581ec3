"""Constants and small geometry / loss helpers (reference loss_utils.py: constants :6-9,
transform_point_cloud :11, generate_grasp_views :23, batch_viewpoint_params_to_matrix :33,
huber_loss :51)."""
import numpy as np
import torch
from torch.nn import functional as F

GRASP_MAX_WIDTH = 0.1
GRASP_MAX_TOLERANCE = 0.05
THRESH_GOOD = 0.7
THRESH_BAD = 0.1


def transform_point_cloud(cloud, transform, format='4x4'):
    """cloud (N,3); transform (3,3) | (3,4) | (4,4)."""
    if format not in ('3x3', '4x4', '3x4'):
        raise ValueError('Unknown transformation format, only support \'3x3\' or \'4x4\' or \'3x4\'.')
    if format == '3x3':
        return torch.matmul(transform, cloud.T).T
    homo = torch.cat([cloud, cloud.new_ones(cloud.size(0), 1)], dim=1)
    return torch.matmul(transform, homo.T).T[:, :3]


_VIEW_CACHE = {}


def generate_grasp_views(N=300, phi=(np.sqrt(5) - 1) / 2, center=np.zeros(3), r=1):
    """N approach directions on a Fibonacci sphere, float32 (N,3) (computed in float64 like the reference)."""
    key = (N, float(phi), tuple(np.asarray(center, dtype=np.float64).tolist()), float(r))
    if key not in _VIEW_CACHE:
        i = np.arange(N, dtype=np.float64)
        zi = (2 * i + 1) / N - 1
        rad = np.sqrt(1 - zi ** 2)
        views = r * np.stack([rad * np.cos(2 * i * np.pi * phi), rad * np.sin(2 * i * np.pi * phi), zi], 1) + center
        _VIEW_CACHE[key] = torch.from_numpy(views.astype(np.float32))
    return _VIEW_CACHE[key].clone()


_VIEW_DEVICE_CACHE = {}


def grasp_views_on(device, N=300):
    """generate_grasp_views(N) resident on `device`, uploaded once (a per-step ``.to(device)`` of the host
    tensor is a synchronous copy that stalls the launch queue).  Read-only: do not modify in place."""
    key = (N, str(device))
    if key not in _VIEW_DEVICE_CACHE:
        _VIEW_DEVICE_CACHE[key] = generate_grasp_views(N).to(device)
    return _VIEW_DEVICE_CACHE[key]


_VIEW_ROT_CACHE = {}


def grasp_view_rotations_on(device, N=300):
    """batch_viewpoint_params_to_matrix(-views, 0) of the N template views, (N,3,3), resident on `device` and computed
    once: a constant of the model (every seed's approach rotation is one of these rows - modules.py:74-79 builds it per
    seed from the picked template view, label_generation.py:52-54 per object).  Row i equals what the per-seed call
    returns for template i (the function is row-wise).  Read-only."""
    key = (N, str(device))
    if key not in _VIEW_ROT_CACHE:
        views = grasp_views_on(device, N)
        _VIEW_ROT_CACHE[key] = batch_viewpoint_params_to_matrix(-views, torch.zeros(N, dtype=views.dtype, device=views.device))
    return _VIEW_ROT_CACHE[key]


def batch_viewpoint_params_to_matrix(batch_towards, batch_angle):
    """Approach vectors (N,3) + in-plane angles (N,) -> rotation matrices (N,3,3) whose first column
    is the normalised approach direction."""
    axis_x = batch_towards
    ones = torch.ones(axis_x.shape[0], dtype=axis_x.dtype, device=axis_x.device)
    zeros = torch.zeros(axis_x.shape[0], dtype=axis_x.dtype, device=axis_x.device)
    axis_y = torch.stack([-axis_x[:, 1], axis_x[:, 0], zeros], dim=-1)
    degenerate = torch.norm(axis_y, dim=-1) == 0
    axis_y[degenerate, 1] = 1
    axis_x = axis_x / torch.norm(axis_x, dim=-1, keepdim=True)
    axis_y = axis_y / torch.norm(axis_y, dim=-1, keepdim=True)
    axis_z = torch.cross(axis_x, axis_y, dim=-1)
    sin, cos = torch.sin(batch_angle), torch.cos(batch_angle)
    R1 = torch.stack([ones, zeros, zeros, zeros, cos, -sin, zeros, sin, cos], dim=-1).reshape([-1, 3, 3])
    R2 = torch.stack([axis_x, axis_y, axis_z], dim=-1)
    return torch.matmul(R2, R1)


def huber_loss(error, delta=1.0):
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def l1_loss_clamp(error, thresh=0.01):
    return F.relu(torch.abs(error) - thresh)
