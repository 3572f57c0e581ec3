"""conv / norm / act block factories (reference ModifiedNetTools/conv.py:8-140): blocks are plain
``nn.Sequential`` so their state_dict keys are positional (``0.weight`` conv, ``1.*`` norm)."""
import torch.nn as nn

from .activation import create_act
from .norm import create_norm


class Conv2d(nn.Conv2d):
    """nn.Conv2d defaulting to a 1x1 kernel when only (in, out) are given."""

    def __init__(self, *args, **kwargs):
        if len(args) == 2 and 'kernel_size' not in kwargs.keys():
            super().__init__(*args, (1, 1), **kwargs)
        else:
            super().__init__(*args, **kwargs)


class Conv1d(nn.Conv1d):
    def __init__(self, *args, **kwargs):
        if len(args) == 2 and 'kernel_size' not in kwargs.keys():
            super().__init__(*args, 1, **kwargs)
        else:
            super().__init__(*args, **kwargs)


def _block(make_main, in_channels, out_channels, dimension, norm_args, act_args, order, bias):
    """Assemble [main, norm, act] in the requested order; the main layer loses its bias when a
    norm layer is present."""
    if order not in ('conv-norm-act', 'norm-act-conv', 'conv-act-norm'):
        raise NotImplementedError(f"{order} is not supported")
    norm_layer = create_norm(norm_args, in_channels if order == 'norm-act-conv' else out_channels,
                             dimension=dimension)
    main = make_main(False if norm_layer is not None else bias)
    act_layer = create_act(act_args) if act_args is not None else None
    parts = {'conv': main, 'norm': norm_layer, 'act': act_layer}
    return nn.Sequential(*[parts[k] for k in order.split('-') if parts[k] is not None])


def create_convblock2d(*args, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
    bias = kwargs.pop('bias', True)
    return _block(lambda b: Conv2d(*args, bias=b, **kwargs), args[0], args[1], '2d', norm_args, act_args,
                  order, bias)


def create_convblock1d(*args, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
    bias = kwargs.pop('bias', True)
    return _block(lambda b: Conv1d(*args, bias=b, **kwargs), args[0], args[1], '1d', norm_args, act_args,
                  order, bias)


def create_linearblock(*args, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
    bias = kwargs.pop('bias', True)
    return _block(lambda b: nn.Linear(*args, bias=b, **kwargs), args[0], args[1], '1d', norm_args, act_args,
                  order, bias)


class CreateResConvBlock2D(nn.Module):
    def __init__(self, mlps, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
        super().__init__()
        self.convs = nn.Sequential()
        for i in range(len(mlps) - 2):
            self.convs.add_module(f'conv{i}', create_convblock2d(mlps[i], mlps[i + 1], norm_args=norm_args,
                                                                 act_args=act_args, order=order, **kwargs))
        self.convs.add_module(f'conv{len(mlps) - 1}',
                              create_convblock2d(mlps[-2], mlps[-1], norm_args=norm_args, act_args=None, **kwargs))
        self.act = create_act(act_args)

    def forward(self, x, res=None):
        return self.act(self.convs(x) + (x if res is None else res))
