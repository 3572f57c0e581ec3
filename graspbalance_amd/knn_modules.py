"""``myknn`` with the interface of the reference's KNN/knn_modules.py:11-18 over the HIP 1-NN kernel
(``gb_knn1``).  The reference calls it with k = 1 only (label_generation.py:58,84)."""
import torch

from . import _lib


def knn(ref, query, idx):
    """KNN._C.knn(ref (B,dim,nref), query (B,dim,nq), idx (B,k,nq) int64 out) -> 1; 1-based indices."""
    if idx.size(1) != 1:
        raise NotImplementedError("only k = 1 is implemented (the only k GraspBalance uses)")
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    for t in (ref, query, idx):
        if not t.is_contiguous():
            raise RuntimeError("knn: tensors must be contiguous")
    B, dim, nref = ref.shape
    nq = query.size(2)
    with _lib.device_ctx(ref.device):
        _lib.check(_lib.lib().gb_knn1(_lib.ptr(ref), _lib.ptr(query), _lib.ptr(idx), B, dim, nref, nq,
                                      _lib.current_stream(ref.device)), "knn")
    return 1


def myknn(ref, query, k=1):
    """Indices (1-based, int64, (B,1,nq)) of the nearest `ref` column of every `query` column."""
    device = ref.device
    ref = ref.float().to(device).contiguous()
    query = query.float().to(device).contiguous()
    inds = torch.empty(query.shape[0], 1, query.shape[2], dtype=torch.long, device=device)
    knn(ref, query, inds)
    return inds
