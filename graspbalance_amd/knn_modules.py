"""``myknn`` with the interface of the reference's KNN/knn_modules.py:11-18 over the HIP k-NN kernels (``gb_knn``: k <= 16;
k = 1 - the only k the reference itself calls, label_generation.py:58,84 - on the one-wave-per-4-queries kernel)."""
import torch

from . import _lib


def knn(ref, query, idx):
    """KNN._C.knn(ref (B,dim,nref), query (B,dim,nq), idx (B,k,nq) int64 out) -> 1; 1-based indices."""
    k = idx.size(1)
    if k < 1 or k > 16:
        raise NotImplementedError("k = %d: gb_knn keeps a query's k best in registers, k <= 16" % k)
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    for t in (ref, query, idx):
        if not t.is_contiguous():
            raise RuntimeError("knn: tensors must be contiguous")
    B, dim, nref = ref.shape
    nq = query.size(2)
    with _lib.device_ctx(ref.device):
        _lib.check(_lib.lib().gb_knn(_lib.ptr(ref), _lib.ptr(query), _lib.ptr(idx), B, dim, nref, nq, k,
                                     _lib.current_stream(ref.device)), "knn")
    return 1


def myknn(ref, query, k=1):
    """Indices (1-based, int64, (B,k,nq)) of the k nearest `ref` columns of every `query` column, nearest first."""
    device = ref.device
    ref = ref.float().to(device).contiguous()
    query = query.float().to(device).contiguous()
    inds = torch.empty(query.shape[0], k, query.shape[2], dtype=torch.long, device=device)
    knn(ref, query, inds)
    return inds
