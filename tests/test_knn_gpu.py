"""GPU: gb_knn1 (wave-per-4-queries dim-3 kernel and the generic-dim kernel) vs the CPU oracle, with
exact ties (duplicated reference points: the lowest index must win) and odd sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,dim,nref,nq", [(1, 3, 300, 300), (4, 3, 2400, 1024), (32, 3, 300, 300), (2, 3, 1, 7),
                                           (3, 3, 65, 3), (2, 2, 500, 130), (1, 5, 700, 260), (2, 8, 64, 64)])
def test_knn1_matches_oracle_with_ties(orc, B, dim, nref, nq):
    from graspbalance_amd import _lib
    g = torch.Generator().manual_seed(B * 1000 + nref)
    ref = torch.randint(0, 5, (B, dim, nref), generator=g).float() * 0.5  # lattice -> many exact ties
    ref[:, :, nref // 2:] = ref[:, :, : nref - nref // 2]                  # duplicated columns
    query = torch.randint(0, 5, (B, dim, nq), generator=g).float() * 0.5 + 0.25 * (torch.rand(B, dim, nq, generator=g) > 0.5)
    want = orc.knn1(ref, query)
    out = torch.zeros(B, 1, nq, dtype=torch.int64, device=DEV)
    r, q = ref.to(DEV), query.to(DEV)
    _lib.check(_lib.lib().gb_knn1(_lib.ptr(r), _lib.ptr(q), _lib.ptr(out), B, dim, nref, nq, None), "knn")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)
    # python-level wrapper (KNN/knn_modules.py:11 myknn)
    from graspbalance_amd.knn_modules import myknn
    assert torch.equal(myknn(r, q).cpu(), want)
