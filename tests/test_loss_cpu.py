"""Host-side loss helpers that were rewritten to avoid device->host synchronisations: same values as the
reference's formulation (TrainModel/loss.py:29-41 masked-assignment loop; boolean-index means)."""
import numpy as np
import torch

from graspbalance_amd.loss import ScalePrior, _masked_fraction


def _lookup_loop(prior, widths):
    idx = torch.zeros(widths.shape, dtype=torch.long)
    for b in range(len(prior.intervals) - 1):
        idx[(prior.intervals[b] < widths) & (prior.intervals[b + 1] > widths)] = b
    return prior.weights[idx]


def test_scale_prior_lookup_matches_masked_assignment_loop():
    rng = np.random.default_rng(0)
    prior = ScalePrior(rng.integers(1, 1000, 32), np.linspace(0.1 / 33, 0.1, 33))
    edges = torch.tensor(prior.intervals, dtype=torch.float32)
    widths = torch.cat([torch.from_numpy(rng.uniform(-0.01, 0.12, 5000).astype(np.float32)),
                        edges, torch.nextafter(edges, edges + 1), torch.nextafter(edges, edges - 1),
                        torch.tensor([0.0, -1.0, 1.0])]).view(2, -1)
    assert torch.equal(prior.lookup(widths), _lookup_loop(prior, widths))


def test_masked_fraction_equals_boolean_index_mean():
    torch.manual_seed(0)
    flags = torch.rand(4, 100, 3) > 0.4
    mask = torch.rand(4, 100, 3) > 0.7
    assert torch.equal(_masked_fraction(flags, mask), flags[mask].float().mean())
    empty = torch.zeros_like(mask)
    assert torch.isnan(_masked_fraction(flags, empty)) and torch.isnan(flags[empty].float().mean())
