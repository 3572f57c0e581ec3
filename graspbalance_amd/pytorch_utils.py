"""Layer containers with the names, signatures and ``state_dict`` keys of the reference's
PointNet/pytorch_utils.py (SharedMLP :5, _BNBase :33, _ConvBase :61, Conv1d/2d/3d :115-226, FC :229,
BNMomentumScheduler :263), so reference checkpoints load unchanged:

    SharedMLP([3, 64, 128], bn=True).state_dict() ->
        layer0.conv.weight, layer0.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}, ...

Own implementation: one table-driven block builder instead of per-dimension classes.
"""
import torch
import torch.nn as nn


def max_over_samples(x, keepdim=False):
    """Max over the neighbour axis (last) of a grouped (B,C,npoint,nsample) tensor: F.max_pool2d(kernel=[1,nsample])
    of the reference (pointnet2_modules.py:165-169, modules.py:121, drp.py:23).  The dim-reduction kernel is ~10x
    faster than the NCHW pooling kernel on ROCm and gives the same values.  One function for every plain-composition
    max-pool of the package, so a test can observe / freeze the routing in one place (tests/routing_tape.py)."""
    return torch.max(x, dim=-1, keepdim=keepdim)[0]


_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_NORM = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


class _BNBase(nn.Sequential):
    """Sequential holding one BatchNorm under the key ``<name>bn`` (weight = 1, bias = 0)."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        norm = batch_norm(in_size)
        nn.init.constant_(norm.weight, 1.0)
        nn.init.constant_(norm.bias, 0)
        self.add_module(name + "bn", norm)


def _bn_class(ndim):
    class _BN(_BNBase):
        def __init__(self, in_size, *, name=""):
            super().__init__(in_size, batch_norm=_NORM[ndim], name=name)
    _BN.__name__ = _BN.__qualname__ = "BatchNorm%dd" % ndim
    return _BN


BatchNorm1d, BatchNorm2d, BatchNorm3d = _bn_class(1), _bn_class(2), _bn_class(3)
_BN_WRAPPER = {1: BatchNorm1d, 2: BatchNorm2d, 3: BatchNorm3d}


class _ConvBase(nn.Sequential):
    """conv (+ bn) (+ activation) in post-activation order, or bn/activation first when ``preact``.
    The conv has a bias only without batch norm; submodule keys are ``conv``, ``bn``, ``activation``."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bias and not bn
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                         bias=use_bias)
        init(conv_unit.weight)
        if use_bias:
            nn.init.constant_(conv_unit.bias, 0)
        pre_or_post = []
        if bn:
            pre_or_post.append((name + "bn", batch_norm(in_size if preact else out_size)))
        if activation is not None:
            pre_or_post.append((name + "activation", activation))
        order = pre_or_post + [(name + "conv", conv_unit)] if preact else [(name + "conv", conv_unit)] + pre_or_post
        for key, mod in order:
            self.add_module(key, mod)


def _conv_class(ndim):
    ones, zeros = (1,) * ndim, (0,) * ndim
    k_default, s_default, p_default = (1, 1, 0) if ndim == 1 else (ones, ones, zeros)

    class _Conv(_ConvBase):
        def __init__(self, in_size, out_size, *, kernel_size=k_default, stride=s_default,
                     padding=p_default, activation=nn.ReLU(inplace=True), bn=False,
                     init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
            super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                             conv=_CONV[ndim], batch_norm=_BN_WRAPPER[ndim], bias=bias, preact=preact,
                             name=name)
    _Conv.__name__ = _Conv.__qualname__ = "Conv%dd" % ndim
    return _Conv


Conv1d, Conv2d, Conv3d = _conv_class(1), _conv_class(2), _conv_class(3)


class SharedMLP(nn.Sequential):
    """Stack of 1x1 ``Conv2d`` blocks ``layer0..layerK`` over ``args = [C0, C1, ..., CK+1]``."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False,
                 name=""):
        super().__init__()
        for i, (cin, cout) in enumerate(zip(args[:-1], args[1:])):
            plain_input = first and preact and i == 0  # the very first pre-activation layer sees raw input
            self.add_module(name + "layer{}".format(i),
                            Conv2d(cin, cout, bn=bn and not plain_input,
                                   activation=None if plain_input else activation, preact=preact))


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None,
                 preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        extras = []
        if bn:
            extras.append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            extras.append((name + "activation", activation))
        order = extras + [(name + "fc", fc)] if preact else [(name + "fc", fc)] + extras
        for key, mod in order:
            self.add_module(key, mod)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Sets every BatchNorm's momentum to ``bn_lambda(epoch)`` on ``step()`` (train.py:110-113,136)."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
