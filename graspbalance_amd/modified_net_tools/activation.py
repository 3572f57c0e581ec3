"""Activation factory and feature-channel map (reference ModifiedNetTools/activation.py:4-66)."""
import copy

from torch import nn

_ACT_LAYER = dict(
    silu=nn.SiLU, swish=nn.SiLU, mish=nn.Mish, relu=nn.ReLU, relu6=nn.ReLU6, leaky_relu=nn.LeakyReLU,
    leakyrelu=nn.LeakyReLU, elu=nn.ELU, prelu=nn.PReLU, celu=nn.CELU, selu=nn.SELU, gelu=nn.GELU,
    sigmoid=nn.Sigmoid, tanh=nn.Tanh, hard_sigmoid=nn.Hardsigmoid, hard_swish=nn.Hardswish,
)

# input channels of the first aggregation conv for each feature recipe, given C feature channels
CHANNEL_MAP = {
    'fj': lambda c: c,
    'df': lambda c: c,
    'assa': lambda c: c * 3,
    'assa_dp': lambda c: c * 3 + 3,
    'dp_fj': lambda c: 3 + c,
    'pj': lambda c: c,
    'dp': lambda c: 3,
    'pi_dp': lambda c: c + 3,
    'pj_dp': lambda c: c + 3,
    'dp_fj_df': lambda c: c * 2 + 3,
    'dp_fi_df': lambda c: c * 2 + 3,
    'pi_dp_fj_df': lambda c: c * 2 + 6,
    'pj_dp_fj_df': lambda c: c * 2 + 6,
    'pj_dp_df': lambda c: c + 6,
    'dp_df': lambda c: c + 3,
}


def create_act(act_args):
    """``{'act': 'relu', ...}`` or a name -> module (in-place by default); None -> None."""
    if act_args is None:
        return None
    act_args = copy.deepcopy(act_args)
    if isinstance(act_args, str):
        act_args = {"act": act_args}
    act = act_args.pop('act', None)
    if act is None:
        return None
    if isinstance(act, str):
        act = act.lower()
        assert act in _ACT_LAYER.keys(), f"input {act} is not supported"
        act_layer = _ACT_LAYER[act]
    inplace = act_args.pop('inplace', True)
    if act in ('gelu', 'sigmoid'):
        return act_layer(**act_args)
    return act_layer(inplace=inplace, **act_args)
