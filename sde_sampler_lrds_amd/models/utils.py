"""Near-zero last-layer initialisers (mirror of ``sde_sampler/models/utils.py:7-49``)."""
import math

import torch

init_weight_scale = 1e-6


def _neg_slope():
    # kaiming gain sqrt(2/(1+a^2)) -> uniform bound init_weight_scale * sqrt(3/fan_in) ... i.e. ~1e-6 weights
    return math.sqrt((6.0 / init_weight_scale ** 2) - 1)


def kaiming_uniform_zeros_(m):
    return torch.nn.init.kaiming_uniform_(m, a=_neg_slope())


def kaiming_normal_zeros_(m):
    return torch.nn.init.kaiming_normal_(m, a=_neg_slope())


def _fan_in(weight):
    return torch.nn.init._calculate_fan_in_and_fan_out(weight)[0]


def init_bias_uniform_zeros(m, weight):
    return init_bias_uniform_constant(m, weight, val=0.0)


def init_bias_normal_zeros(m, weight):
    return init_bias_normal_constant(m, weight, val=0.0)


def init_bias_uniform_constant(m, weight, val=1.0):
    fan = _fan_in(weight)
    if fan > 0:
        bound = init_weight_scale / math.sqrt(fan)
        return torch.nn.init.uniform_(m, val - bound, val + bound)
    return torch.nn.init.constant_(m, val)


def init_bias_normal_constant(m, weight, val=1.0):
    fan = _fan_in(weight)
    if fan > 0:
        return torch.nn.init.normal_(m, mean=val, std=init_weight_scale / math.sqrt(fan))
    return torch.nn.init.constant_(m, val)
