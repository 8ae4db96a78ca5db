"""Network *views*: the reference's Actor / Critic / SACActorModel (src/model.py) as named
windows onto the engine's flat parameter vectors.

The arithmetic of these networks lives in csrc (batched fp32-MFMA GEMM + fused epilogues);
what this module keeps from the reference is the checkpoint surface: `state_dict()` /
`load_state_dict()` with the reference's key names and [out, in] weight layout
(`base_net.{0,2,..}.weight`, `net.*`, `mean_head.*`, BatchNorm buffers), `save` / `load` of
`.pth` files that the reference can read back, and the no-op `train()` / `eval()` switches the
trainer calls (src/env.py:641).
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from .. import _ffi
from .._ffi import lib


class NetView:
    kind = "mlp"       # "actor" | "critic" | "sac_actor"
    prefix = "net"

    def __init__(self, agent_handle_getter, name: str, in_dim: int, hidden_dim: int, out_dim: int,
                 layer_stack: int):
        self._agent = agent_handle_getter
        self.name = name
        self.in_dim, self.hidden_dim, self.out_dim, self.layer_stack = in_dim, hidden_dim, out_dim, layer_stack
        self.training = True
        self.num_batches_tracked = 0

    # ---- layout: (key, shape) in module.parameters() order = order of the flat vector
    def _param_layout(self):
        L, H = self.layer_stack, self.hidden_dim
        out = []
        if self.kind == "sac_actor":
            for l in range(L):
                k = self.in_dim if l == 0 else H
                out += [(f"base_net.{3 * l}.weight", (H, k)), (f"base_net.{3 * l}.bias", (H,)),
                        (f"base_net.{3 * l + 1}.weight", (H,)), (f"base_net.{3 * l + 1}.bias", (H,))]
            out += [("mean_head.weight", (self.out_dim, H)), ("mean_head.bias", (self.out_dim,)),
                    ("log_std_head.weight", (self.out_dim, H)), ("log_std_head.bias", (self.out_dim,))]
        else:
            for l in range(L + 1):
                k = self.in_dim if l == 0 else H
                n = H if l < L else self.out_dim
                out += [(f"{self.prefix}.{2 * l}.weight", (n, k)), (f"{self.prefix}.{2 * l}.bias", (n,))]
        return out

    def numel(self) -> int:
        return int(sum(int(np.prod(s)) for _, s in self._param_layout()))

    # ---- flat vector I/O through the C ABI
    def _get(self, name: str) -> np.ndarray:
        h = self._agent()
        n = lib.gcrl_agent_numel(h, name.encode())
        if n < 0:
            raise KeyError(name)
        buf = np.empty(n, dtype=np.float32)
        _ffi.check(lib.gcrl_agent_get(h, name.encode(), buf.ctypes.data, n))
        return buf

    def _set(self, name: str, arr: np.ndarray):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        _ffi.check(lib.gcrl_agent_set(self._agent(), name.encode(), arr.ctypes.data, arr.size))

    def flat(self) -> np.ndarray:
        return self._get(self.name)

    def set_flat(self, arr):
        self._set(self.name, arr)

    def grad_flat(self) -> np.ndarray:
        return self._get("grad:" + self.name)

    def split(self, flat: np.ndarray) -> "OrderedDict[str, np.ndarray]":
        out, off = OrderedDict(), 0
        for key, shape in self._param_layout():
            n = int(np.prod(shape))
            out[key] = flat[off:off + n].reshape(shape)
            off += n
        return out

    def join(self, named) -> np.ndarray:
        parts = []
        for key, shape in self._param_layout():
            t = named[key]
            t = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{key}: expected shape {shape}, got {tuple(t.shape)}")
            parts.append(np.asarray(t, dtype=np.float32).reshape(-1))
        return np.concatenate(parts)

    # ---- torch.nn.Module-like surface
    def state_dict(self):
        sd = OrderedDict()
        named = self.split(self.flat())
        if self.kind != "sac_actor":
            for k, v in named.items():
                sd[k] = torch.from_numpy(v.copy())
            return sd
        H, L = self.hidden_dim, self.layer_stack
        rm = self._get("bn_running_mean").reshape(L, H)
        rv = self._get("bn_running_var").reshape(L, H)
        for k, v in named.items():
            sd[k] = torch.from_numpy(v.copy())
            if k.startswith("base_net.") and k.endswith(".bias") and int(k.split(".")[1]) % 3 == 1:
                l = int(k.split(".")[1]) // 3
                sd[f"base_net.{3 * l + 1}.running_mean"] = torch.from_numpy(rm[l].copy())
                sd[f"base_net.{3 * l + 1}.running_var"] = torch.from_numpy(rv[l].copy())
                sd[f"base_net.{3 * l + 1}.num_batches_tracked"] = torch.tensor(self.num_batches_tracked)
        return sd

    def load_state_dict(self, sd):
        self.set_flat(self.join(sd))
        if self.kind == "sac_actor":
            H, L = self.hidden_dim, self.layer_stack
            keys = [f"base_net.{3 * l + 1}.running_mean" for l in range(L)]
            if all(k in sd for k in keys):
                rm = np.stack([np.asarray(sd[f"base_net.{3 * l + 1}.running_mean"], dtype=np.float32) for l in range(L)])
                rv = np.stack([np.asarray(sd[f"base_net.{3 * l + 1}.running_var"], dtype=np.float32) for l in range(L)])
                self._set("bn_running_mean", rm)
                self._set("bn_running_var", rv)
                nb = sd.get(f"base_net.1.num_batches_tracked")
                if nb is not None:
                    self.num_batches_tracked = int(nb)

    def load(self, weights: str, device: str = "cuda"):
        self.load_state_dict(torch.load(weights, map_location="cpu"))

    def save(self, path: str):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.state_dict(), path)

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [torch.from_numpy(v.copy()) for v in self.split(self.flat()).values()]


class Actor(NetView):
    """Linear+LeakyReLU x L, Linear, Tanh (src/model.py:7-45)."""
    kind, prefix = "actor", "base_net"


class Critic(NetView):
    """Linear+LeakyReLU x L, Linear -> 1 (src/model.py:48-83)."""
    kind, prefix = "critic", "net"


class SACActorModel(NetView):
    """Linear+BatchNorm1d+ReLU x L, mean/log_std heads, tanh-Gaussian (src/model.py:86-156)."""
    kind, prefix = "sac_actor", "base_net"
