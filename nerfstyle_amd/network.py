"""`Network`: the stand-in for `tinycudann.Network` as the reference uses it
(networks/style_nerf.py:44-98): constructed as Network(n_input_dims, n_output_dims,
network_config, seed), exposes a flat fp32 `params` Parameter and `n_output_dims`.

tinycudann is an un-vendored, unpinned third-party dependency of the reference (README.md:26),
so its numerics and its internal parameter order are not available ("parity unpinned"); this
module defines the contract instead (see include/nsr.h, nsr_mlp_forward):
  * bias-free; ReLU hidden layers of width 64; output activation None | Sigmoid;
  * params = row-major [out, in] matrices, layers concatenated, last layer padded to 16 rows;
  * inputs / weights / hidden activations rounded to `dtype` (float16 like tcnn, or bfloat16),
    fp32 accumulation on the MFMA units; outputs fp32.
Initialisation: Xavier-uniform per layer from torch.Generator(seed) (tcnn also uses a seeded
Xavier-uniform, but the streams differ).
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_ACT = {'None': L.NSR_ACT_NONE, 'Sigmoid': L.NSR_ACT_SIGMOID}


def mlp_layer_shapes(n_in, n_out, n_neurons=64, n_hidden_layers=1):
    pad16 = lambda v: (v + 15) // 16 * 16
    shapes, d = [], pad16(n_in)
    for _ in range(n_hidden_layers):
        shapes.append((n_neurons, d))
        d = n_neurons
    shapes.append((pad16(n_out), d))
    return shapes


def init_mlp_params(n_in, n_out, n_neurons, n_hidden_layers, seed):
    g = torch.Generator().manual_seed(int(seed))
    chunks = []
    for (o, i) in mlp_layer_shapes(n_in, n_out, n_neurons, n_hidden_layers):
        a = math.sqrt(6.0 / (o + i))
        chunks.append(((torch.rand(o, i, generator=g) * 2 - 1) * a).reshape(-1))
    return torch.cat(chunks)


class _mlp(Function):
    @staticmethod
    def forward(ctx, x, params, cfg):
        n_in, n_out, n_neurons, n_hidden, act, cd = cfg
        x = x.detach().to(torch.float32).contiguous()
        p = params.detach().contiguous()
        M = x.shape[0]
        y = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
        L.check(L.lib().nsr_mlp_forward(L.p(x), L.p(p), M, n_in, n_out, n_neurons, n_hidden, act, cd, L.p(y),
                                        L.stream()), 'mlp_forward')
        ctx.save_for_backward(x, p, y)
        ctx.cfg = cfg
        ctx.need_dx = True
        return y

    @staticmethod
    def backward(ctx, dy):
        x, p, y = ctx.saved_tensors
        n_in, n_out, n_neurons, n_hidden, act, cd = ctx.cfg
        dy = dy.to(torch.float32).contiguous()
        M = x.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dp = torch.zeros_like(p) if ctx.needs_input_grad[1] else None
        L.check(L.lib().nsr_mlp_backward(L.p(x), L.p(p), L.p(y), L.p(dy), M, n_in, n_out, n_neurons, n_hidden, act, cd,
                                         L.p(dx), L.p(dp), L.stream()), 'mlp_backward')
        return dx, dp, None


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337, dtype=torch.float16):
        super().__init__()
        assert network_config.get('otype', 'FullyFusedMLP') == 'FullyFusedMLP'
        assert network_config.get('activation', 'ReLU') == 'ReLU'
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.n_neurons = int(network_config['n_neurons'])
        self.n_hidden_layers = int(network_config['n_hidden_layers'])
        self.output_activation = network_config.get('output_activation', 'None')
        self.seed = seed
        self.compute_dtype = dtype
        self.params = nn.Parameter(init_mlp_params(n_input_dims, n_output_dims, self.n_neurons,
                                                   self.n_hidden_layers, seed))

    def cfg(self):
        return (self.n_input_dims, self.n_output_dims, self.n_neurons, self.n_hidden_layers,
                _ACT[self.output_activation], L.dt(self.compute_dtype))

    def forward(self, x):
        return _mlp.apply(x.view(-1, self.n_input_dims), self.params, self.cfg())
