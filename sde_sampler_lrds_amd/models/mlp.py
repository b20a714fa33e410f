"""Drift nets (mirror of ``sde_sampler/models/mlp.py``: TimeEmbed :57-96, FourierMLP :99-143).

Parameter and buffer names equal the reference's, so ``state_dict``s move both ways.  ``forward`` is a
torch implementation for host-side use (training with autograd, ad-hoc calls); the simulate path reads the
parameters through ``engine.net_desc`` and evaluates the net with FP32 MFMA in csrc/sim_device.hpp."""
from __future__ import annotations

import inspect
from typing import Callable

import torch
from torch import nn


def _init_linear(layer: nn.Linear, bias_init=None, weight_init=None):
    if weight_init:
        weight_init(layer.weight)
    if bias_init:
        fn = getattr(bias_init, "func", bias_init)
        if "weight" in inspect.signature(fn).parameters:
            bias_init(layer.bias, weight=layer.weight)
        else:
            bias_init(layer.bias)


class TimeEmbed(nn.Module):
    def __init__(self, dim_out: int, activation: Callable, num_layers: int = 2, channels: int = 64,
                 last_bias_init=None, last_weight_init=None):
        super().__init__()
        self.dim, self.dim_in, self.dim_out = 1, 2, dim_out
        self.channels, self.activation = channels, activation
        self.register_buffer("timestep_coeff", torch.linspace(start=0.1, end=100, steps=channels).unsqueeze(0), persistent=False)
        self.timestep_phase = nn.Parameter(torch.randn(1, channels))
        self.hidden_layer = nn.ModuleList([nn.Linear(2 * channels, channels)] +
                                          [nn.Linear(channels, channels) for _ in range(num_layers - 2)])
        self.out_layer = nn.Linear(channels, dim_out)
        _init_linear(self.out_layer, last_bias_init, last_weight_init)

    def forward(self, t, *args):
        t = t.view(-1, 1).float()
        ang = (self.timestep_coeff * t) + self.timestep_phase
        h = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
        for layer in self.hidden_layer:
            h = self.activation(layer(h))
        return self.out_layer(h)


class FourierMLP(nn.Module):
    def __init__(self, dim: int, activation: Callable, num_layers: int = 4, channels: int = 64, last_bias_init=None,
                 last_weight_init=None, use_angle_encoding: bool = False, dim_out=None, **kwargs):
        super().__init__()
        if use_angle_encoding:
            raise NotImplementedError("use_angle_encoding is not part of the engine (conf/model/base/fouriermlp.yaml: False)")
        self.dim, self.dim_in, self.dim_out = dim, dim + 1, dim_out or dim
        self.channels, self.activation = channels, activation
        self.input_embed = nn.Linear(dim, channels)
        self.timestep_embed = TimeEmbed(dim_out=channels, activation=activation, num_layers=2, channels=channels)
        self.hidden_layer = nn.ModuleList([nn.Linear(channels, channels) for _ in range(num_layers - 2)])
        self.out_layer = nn.Linear(channels, self.dim_out)
        _init_linear(self.out_layer, last_bias_init, last_weight_init)

    def forward(self, t, x):
        t = t.view(-1, 1).expand(x.shape[0], 1).float()
        h = self.input_embed(x) + self.timestep_embed(t)
        for layer in self.hidden_layer:
            h = layer(self.activation(h))
        return self.out_layer(self.activation(h))
