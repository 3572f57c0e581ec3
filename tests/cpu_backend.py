"""Test-only: bind the package's extension hooks to the CPU oracle so the python layers (modules,
models, train step, gloo data parallel) can be exercised without a GPU.  The product never does this."""
from oracle import oracle as orc


def install(monkeypatch=None):
    from graspbalance_amd import knn_modules, pointnet2_utils
    from graspbalance_amd.modified_net_tools import group, subsample, upsampling
    targets = [(pointnet2_utils, "_ext", orc.ExtBackend()),
               (group, "pointnet2_cuda", orc.PBBackend()),
               (subsample, "pointnet2_cuda", orc.PBBackend()),
               (upsampling, "pointnet2_cuda", orc.PBBackend())]
    for mod, name, val in targets:
        if monkeypatch is not None:
            monkeypatch.setattr(mod, name, val)
        else:
            setattr(mod, name, val)
    # knn: bypass the CUDA check
    if monkeypatch is not None:
        monkeypatch.setattr(knn_modules, "knn", orc.knn_into)
    else:
        knn_modules.knn = orc.knn_into
