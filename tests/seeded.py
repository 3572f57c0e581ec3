"""Test-only helpers shared by tests/golden/make_golden_r2.py (build container) and the parity tests: parameters that
are a function of the state_dict KEY, so the reference's modules and this repo's modules hold identical weights
without shipping a checkpoint, and slices / norms that summarise a large tensor in a small fixture."""
import zlib

import numpy as np
import torch


def fill_by_key(module, seed=0):
    """Overwrite every parameter / buffer of `module` with values drawn from a generator seeded by the entry's
    state_dict key (+ seed): conv / linear weights ~ N(0, 2/fan_in), BatchNorm weights 1 + 0.1 N, biases 0.05 N,
    running_mean 0.1 N, running_var U(0.5, 1.5), counters 0.  Same key + same shape -> same values in any module."""
    with torch.no_grad():
        for key, t in module.state_dict().items():
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7fffffff)
            if key.endswith("num_batches_tracked"):
                t.zero_()
            elif key.endswith("running_var"):
                t.copy_(torch.rand(t.shape, generator=g) + 0.5)
            elif key.endswith("running_mean"):
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif key.endswith("bias"):
                t.copy_(torch.randn(t.shape, generator=g) * 0.05)
            elif t.dim() == 1:  # BatchNorm weight
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, generator=g))
            else:
                fan_in = int(np.prod(t.shape[1:]))
                t.copy_(torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5)
    return module


def summary(t, max_elems=16384):
    """Small, deterministic summary of a tensor for a fixture: shape, L2 norm, sum (fp64) and an evenly strided
    sample of at most `max_elems` entries of the flattened tensor."""
    a = t.detach().cpu().contiguous()
    flat = a.reshape(-1)
    step = max(1, -(-flat.numel() // max_elems))
    d = flat.double()
    return {"shape": np.asarray(a.shape, dtype=np.int64), "norm": np.float64(d.norm().item()),
            "sum": np.float64(d.sum().item()), "step": np.int64(step), "sample": flat[::step].numpy().copy()}


def save_summaries(path, tensors, extra=None):
    out = {}
    for name, t in tensors.items():
        for k, v in summary(t).items():
            out[name.replace("/", "__") + "::" + k] = v
    for k, v in (extra or {}).items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(path, **out)


def check_summary(npz, name, t, rtol, what=None):
    """Compare tensor `t` with the stored summary of `name`: identical shape (asserted); returns the relative L2
    error of the strided sample (or a quarter of the relative norm error, if larger) and asserts it <= rtol unless
    rtol is None."""
    key = name.replace("/", "__")
    shape = tuple(int(v) for v in npz[key + "::shape"])
    assert tuple(t.shape) == shape, (what or name, tuple(t.shape), shape)
    flat = t.detach().cpu().contiguous().reshape(-1)
    got = flat[::int(npz[key + "::step"])].double().numpy()
    want = npz[key + "::sample"].astype(np.float64)
    scale = np.linalg.norm(want) + 1e-30
    err = float(np.linalg.norm(got - want) / scale)
    norm = float(flat.double().norm())
    want_norm = float(npz[key + "::norm"])
    err = max(err, abs(norm - want_norm) / (want_norm + 1e-30) / 4)  # the whole tensor's norm guards what the stride skips
    if rtol is not None:
        assert err <= rtol, (what or name, "relative L2 of the sample / norm", err, rtol)
    return err


def assert_errors(errs, tol, default):
    """errs {key: measured}, tol {key prefix: bound}: every error within the bound of its longest matching prefix."""
    bad = {}
    for k, e in errs.items():
        match = [p for p in tol if k.startswith(p)]
        bound = tol[max(match, key=len)] if match else default
        if not e <= bound:
            bad[k] = (e, bound)
    assert not bad, bad
