"""Array backend of the policy layer: the same controller / closed-loop code runs on numpy arrays (host path of the engine,
CPU tests with the oracle double) or on ROCm torch tensors (device path: every per-instance state array of the policy
automata lives in HBM and no step of the loop copies it to the host).  Only the handful of operations the policies use.

On the device the per-step automata themselves run as engine kernels (smpc_policy_step / smpc_loop_*, kernels_policy.hpp);
what goes through TorchOps there is the set-up (initialize, setGuess, checkGuess), the abort-event bookkeeping and the
result assembly -- and any step() of a controller driven by a solver object that does not offer those kernels.
"""
from __future__ import annotations

import numpy as np


class InPlaceState:
    """Mixin: with ``_inplace`` set, assigning a tensor to an attribute that already holds a tensor of the same shape and dtype
    copies INTO the existing storage instead of rebinding the name.  The policy code keeps its plain ``self.fails = ...``
    statements, and every piece of per-instance state stays at a fixed address in HBM -- which is what lets a whole
    closed-loop step be captured once as a hipGraph and replayed (closed_loop.py)."""
    _inplace = False

    def __setattr__(self, name, value):
        if self.__dict__.get('_inplace') and not name.startswith('_'):
            cur = self.__dict__.get(name)
            if (cur is not None and hasattr(cur, 'copy_') and hasattr(value, 'dtype') and cur is not value
                    and tuple(cur.shape) == tuple(getattr(value, 'shape', ())) and cur.dtype == value.dtype):
                cur.copy_(value)
                return
        object.__setattr__(self, name, value)


class NumpyOps:
    name = 'numpy'
    on_device = False
    f64, i64, i32, bool_, u8 = np.float64, np.int64, np.int32, np.bool_, np.uint8

    def asarray(self, a, dtype=None):
        if type(a).__module__.startswith('torch'):
            a = a.detach().cpu().numpy()
        return np.array(a, dtype=dtype, copy=True)

    def zeros(self, shape, dtype=np.float64):
        return np.zeros(shape, dtype)

    def full(self, shape, value, dtype=np.float64):
        return np.full(shape, value, dtype)

    def arange(self, n):
        return np.arange(n)

    def copy(self, a):
        return np.copy(a)

    def where(self, c, a, b):
        return np.where(c, a, b)

    def cast(self, a, dtype):
        return a.astype(dtype)

    def repeat_nodes(self, x, n):
        """x[B, nx] -> [B, n, nx]"""
        return np.repeat(x[:, None, :], n, axis=1)

    def all_tail(self, m):
        """all() over every axis but the first"""
        return np.all(m.reshape(m.shape[0], -1), axis=1)

    def any(self, m):
        return bool(np.any(m))

    def last_true(self, m):
        """index of the last True along axis 1, -1 if none"""
        n = m.shape[1]
        return np.where(m.any(1), n - 1 - np.argmax(m[:, ::-1], axis=1), -1)

    def take_rows(self, a, idx):
        """a[b, idx[b]] for every b"""
        return a[np.arange(a.shape[0]), idx]

    def norm_tail(self, a):
        return np.linalg.norm(a.reshape(a.shape[0], -1), axis=1)

    def clip_max(self, a, hi):
        return np.minimum(a, hi)

    def swap_last(self, a):
        """[c, B, n] -> [B, n, c]"""
        return np.moveaxis(a, 0, -1)

    def host(self, a):
        return np.asarray(a)

    # step-indexed logs: `j` is a python int here, a one-element device tensor on the torch backend
    def step_index(self, j0=0):
        return [j0]

    def put_row(self, log, j, value, offset=0):
        log[j[0] + offset] = value

    def step_vec(self, j, n, offset=0):
        return np.full(n, j[0] + offset, np.int64)

    def step_advance(self, j):
        j[0] += 1


class TorchOps:
    name = 'torch'
    on_device = True

    def __init__(self, device):
        import torch
        self.t = torch
        self.device = torch.device('cuda', int(device)) if not isinstance(device, torch.device) else device
        self.f64, self.i64, self.i32, self.bool_, self.u8 = torch.float64, torch.int64, torch.int32, torch.bool, torch.uint8

    def asarray(self, a, dtype=None):
        t = self.t
        if isinstance(a, t.Tensor):
            return a.to(device=self.device, dtype=dtype, copy=True)
        return t.tensor(np.asarray(a), dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=None):
        return self.t.zeros(shape, dtype=dtype or self.f64, device=self.device)

    def full(self, shape, value, dtype=None):
        return self.t.full(shape if isinstance(shape, tuple) else (shape,), value, dtype=dtype or self.f64, device=self.device)

    def arange(self, n):
        return self.t.arange(n, device=self.device)

    def copy(self, a):
        return a.clone()

    def where(self, c, a, b):
        return self.t.where(c, a, b)

    def cast(self, a, dtype):
        return a.to(dtype)

    def repeat_nodes(self, x, n):
        return x[:, None, :].repeat(1, n, 1).contiguous()

    def all_tail(self, m):
        return m.reshape(m.shape[0], -1).all(dim=1)

    def any(self, m):
        return bool(m.any().item())          # the ONE host synchronisation a caller may ask for

    def last_true(self, m):
        n = m.shape[1]
        idx = self.t.arange(n, device=m.device)[None, :]
        return self.t.where(m, idx, self.t.full_like(idx, -1)).max(dim=1).values

    def take_rows(self, a, idx):
        return a[self.t.arange(a.shape[0], device=a.device), idx]

    def norm_tail(self, a):
        return self.t.linalg.vector_norm(a.reshape(a.shape[0], -1), dim=1)

    def clip_max(self, a, hi):
        return self.t.clamp(a, max=hi)

    def swap_last(self, a):
        return a.permute(1, 2, 0)

    def host(self, a):
        return a.detach().cpu().numpy()

    def step_index(self, j0=0):
        return self.t.full((1,), j0, dtype=self.i64, device=self.device)

    def put_row(self, log, j, value, offset=0):
        log.index_copy_(0, j + offset if offset else j, value[None])

    def step_vec(self, j, n, offset=0):
        return (j + offset).expand(n)

    def step_advance(self, j):
        j.add_(1)
