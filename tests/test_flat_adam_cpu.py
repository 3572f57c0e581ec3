"""FlatAdam (one flat parameter buffer, one update launch) follows torch.optim.Adam's trajectory, also under the
OneCycleLR schedule of the reference's train.py (:107-108, which cycles lr AND beta1), and keeps Adam's state_dict
layout."""
import torch
import torch.nn as nn
from torch.optim.lr_scheduler import OneCycleLR


def _net(seed):
    torch.manual_seed(seed)
    return nn.Sequential(nn.Linear(7, 16), nn.BatchNorm1d(16), nn.ReLU(), nn.Linear(16, 5), nn.Linear(5, 3, bias=False))


def _run(opt_cls, steps, wd=0.0, **kw):
    net = _net(0)
    opt = opt_cls(net.parameters(), lr=1e-2, weight_decay=wd, **kw)
    sched = OneCycleLR(opt, max_lr=1e-2, steps_per_epoch=4, epochs=3)
    torch.manual_seed(1)
    xs = [torch.randn(9, 7) for _ in range(steps)]
    for x in xs:
        loss = net(x).square().mean()
        for p in net.parameters():
            p.grad = None
        loss.backward()
        opt.step()
        sched.step()
    return net, opt


def test_flat_adam_follows_torch_adam():
    from graspbalance_amd.flat_adam import FlatAdam
    for wd in (0.0, 0.01):
        a, _ = _run(torch.optim.Adam, 10, wd)
        b, opt = _run(FlatAdam, 10, wd)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
        # the parameters live in one flat buffer
        base = opt._flat_p.data_ptr()
        assert all(base <= p.data_ptr() < base + opt._flat_p.numel() * 4 for p in b.parameters())


def test_flat_adam_state_dict_layout_and_resume():
    from graspbalance_amd.flat_adam import FlatAdam
    ref_net, ref_opt = _run(torch.optim.Adam, 4)
    net, opt = _run(FlatAdam, 4)
    sd, ref_sd = opt.state_dict(), ref_opt.state_dict()
    assert sd['state'].keys() == ref_sd['state'].keys()
    for k in sd['state']:
        assert set(sd['state'][k]) == {'step', 'exp_avg', 'exp_avg_sq'}
        assert torch.allclose(sd['state'][k]['exp_avg'], ref_sd['state'][k]['exp_avg'], rtol=1e-5, atol=1e-8)
        assert float(sd['state'][k]['step']) == float(ref_sd['state'][k]['step']) == 4.0
    # resume a fresh FlatAdam from torch.optim.Adam's state: same next step as torch's own continuation
    net2 = _net(0)
    net2.load_state_dict(ref_net.state_dict())
    opt2 = FlatAdam(net2.parameters(), lr=1e-2)
    opt2.load_state_dict(ref_sd)
    x = torch.randn(9, 7)
    for n, o in ((ref_net, ref_opt), (net2, opt2)):
        for p in n.parameters():
            p.grad = None
        n(x).square().mean().backward()
        o.step()
    for pa, pb in zip(ref_net.parameters(), net2.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
