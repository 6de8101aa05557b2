"""Learned safe set: host side of reference src/safe_mpc/safe_set.py.

``NeuralNetwork`` has the reference's architecture and ``state_dict`` keys (safe_set.py:26-43) so its checkpoints
(``{'model','mean','std'}``, safe_set.py:76-85) load unchanged.  PyTorch only *holds* the weights here; evaluation and
differentiation inside the OCP happen in the HIP engine (``smpc_set_mlp``), replacing l4casadi (safe_set.py:89-94).

The reference's checkpoints are not distributed (README.md:5, .gitignore:10).  ``network_path: 'synthetic:<seed>'``
builds a seeded stand-in with the default ``nn.Linear`` initialisation and an output bias that keeps the set non-empty
(SURVEY 8d, C1); normalisation is the mid-range / uniform std of the joint box.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn


def activation(name):
    return {'relu': nn.ReLU(), 'elu': nn.ELU(), 'tanh': nn.Tanh(), 'gelu': nn.GELU(approximate='tanh'),
            'silu': nn.SiLU()}[name]


class NeuralNetwork(nn.Module):
    """input -> hidden -> hidden -> hidden -> output with the activation between (safe_set.py:26-43)."""

    def __init__(self, input_size, hidden_size, output_size, activation=nn.ReLU()):
        super().__init__()
        self.linear_stack = nn.Sequential(
            nn.Linear(input_size, hidden_size), activation,
            nn.Linear(hidden_size, hidden_size), activation,
            nn.Linear(hidden_size, hidden_size), activation,
            nn.Linear(hidden_size, output_size))

    def forward(self, x):
        return self.linear_stack(x)


class SafeSetNet:
    """Weights + normalisation of the safe-set network, ready to hand to the engine."""

    ACT_CODES = {'gelu': 0, 'relu': 1, 'elu': 2, 'tanh': 3, 'silu': 4}      # SMPC_ACT_* of include/smpc.h (parser.py:95-102)

    def __init__(self, model: NeuralNetwork, mean, std, act='gelu'):
        if act not in self.ACT_CODES:
            raise ValueError(f'unknown activation {act!r} (parser.py:95-102 offers {sorted(self.ACT_CODES)})')
        self.act = act
        self.model = model.float().eval()
        self.mean = np.asarray(mean, np.float64).reshape(-1)
        self.std = np.asarray(std, np.float64).reshape(-1)
        lin = [m for m in self.model.linear_stack if isinstance(m, nn.Linear)]
        self.weights = [l.weight.detach().cpu().numpy().astype(np.float32).copy() for l in lin]
        self.biases = [l.bias.detach().cpu().numpy().astype(np.float32).copy() for l in lin]
        self.dims = [self.weights[0].shape[1]] + [w.shape[0] for w in self.weights]

    @classmethod
    def from_params(cls, params, x_min=None, x_max=None):
        path = params.net_path
        if path.startswith('synthetic'):
            seed = int(path.split(':')[1]) if ':' in path else 0
            return cls.synthetic(params.net_size, params.n_dof_safe_set, x_min, x_max, seed, params.act)
        if not os.path.isabs(path):
            path = os.path.join(params.ROOT_DIR, 'src', path) if not os.path.exists(path) else path
        data = torch.load(path, map_location='cpu')           # safe_set.py:76-77
        net = NeuralNetwork(*params.net_size, activation(params.act))
        net.load_state_dict(data['model'])
        mean = np.asarray(data['mean'], float).reshape(-1)
        std = np.asarray(data['std'], float).reshape(-1)
        n = params.n_dof_safe_set
        if mean.size == 1:
            mean, std = np.full(n, mean.item()), np.full(n, std.item())
        return cls(net, mean, std, params.act)

    @classmethod
    def synthetic(cls, net_size, n_dof, x_min, x_max, seed=0, act='gelu', out_bias=3.0):
        torch.manual_seed(seed)
        net = NeuralNetwork(int(net_size[0]), int(net_size[1]), int(net_size[2]), activation(act))
        with torch.no_grad():
            net.linear_stack[-1].bias.fill_(out_bias)
        q_lo, q_hi = np.asarray(x_min)[:n_dof], np.asarray(x_max)[:n_dof]
        mean = 0.5 * (q_lo + q_hi)
        std = (q_hi - q_lo) / np.sqrt(12.0)
        return cls(net, mean, std, act)

    def torch_value_and_grad(self, s):
        """fp32 value and input gradient through torch autograd (what l4casadi computes)."""
        t = torch.tensor(np.asarray(s, np.float32), requires_grad=True)
        y = self.model(t).reshape(-1)
        g, = torch.autograd.grad(y.sum(), t)
        return y.detach().numpy(), g.numpy()
