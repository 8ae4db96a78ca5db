"""Host-side configuration and normaliser, field-compatible with the reference's src/utils.py.

Only what the hot path consumes is mirrored: the hyper-parameter models whose FIELD NAMES the
engine reads (reference src/utils.py:10-65), the YAML loaders (:177-194), the running
normaliser that sits in front of push() (:68-117) and set_seed (:197-208, minus the gym env).
The gym wrappers of the reference are env-side and out of scope.
"""
from __future__ import annotations

import os
import random
from typing import Union

import numpy as np
import yaml
from pydantic import BaseModel, Field


class BaseAgentConfig(BaseModel):
    """Same 23 required fields as the reference (src/utils.py:10-33); unknown YAML keys are
    ignored exactly as pydantic's default does there."""
    hidden_dim: int = Field(..., ge=1)
    layer_count: int = Field(..., ge=1)
    actor_lr: float = Field(..., gt=0)
    actor_lr_min: float = Field(..., gt=0)
    ac_scheduler_steps: int = Field(..., ge=1)
    critic_lr: float = Field(..., gt=0)
    critic_lr_min: float = Field(..., gt=0)
    cr_scheduler_steps: int = Field(..., ge=1)
    buffer_type: str
    max_len: int = Field(..., ge=1)
    alpha: float = Field(..., ge=0)
    batch_size: int = Field(..., ge=1)
    gamma: float = Field(..., ge=0, le=1)
    ac_update_freq: int = Field(..., ge=1)
    noise_std: float = Field(..., ge=0)
    noise_clamp: float = Field(..., ge=0)
    policy_noise: float = Field(..., ge=0)
    grad_clip: float = Field(..., ge=0)
    beta: float = Field(..., ge=0)
    beta_end: int = Field(..., ge=1)
    k_future: int = Field(..., ge=0)
    max_eps_len: int = Field(..., ge=1)
    tau: float = Field(..., ge=0)


class SACAgentConfig(BaseAgentConfig):
    alpha_lr: float = Field(default=0.0003, gt=0)
    alpha_min: float = Field(default=0.05, gt=0)
    alpha_min_steps: float = Field(..., ge=0)


class HERConfig(BaseModel):
    max_episode: int = Field(..., ge=1)
    max_cycle: int = Field(..., ge=1)
    max_epoch: int = Field(..., ge=1)
    save_freq: int = Field(..., ge=1)
    video_freq: int = Field(..., ge=1)
    window_size: int = Field(..., ge=1)
    gradient_step: int = Field(..., ge=1)
    reset_freq: int = Field(..., ge=1)
    g_normalize: bool = Field(default=False)
    obs_normalize: bool = Field(default=True)
    agent: Union[BaseAgentConfig, SACAgentConfig]


class Config(BaseModel):
    max_frames: int = Field(..., ge=1)
    save_freq: int = Field(..., ge=1)
    video_freq: int = Field(..., ge=1)
    window_size: int = Field(..., ge=1)
    gradient_step: int = Field(..., ge=1)
    reset_freq: int = Field(..., ge=1)
    g_normalize: bool = Field(default=True)
    obs_normalize: bool = Field(default=True)
    agent: Union[BaseAgentConfig, SACAgentConfig]


def _agent_model(agent_type: str):
    return SACAgentConfig if agent_type in ("SAC", "TQC") else BaseAgentConfig


def load_config(path: str, agent_type: str) -> Config:
    with open(path, "r") as fh:
        raw = yaml.safe_load(fh)
    raw["agent"] = _agent_model(agent_type)(**raw["agent"])
    return Config(**raw)


def load_her_config(path: str, agent_type: str) -> HERConfig:
    with open(path, "r") as fh:
        raw = yaml.safe_load(fh)
    raw["agent"] = _agent_model(agent_type)(**raw["agent"])
    return HERConfig(**raw)


class RunningNormalizer:
    """Streaming mean/variance with the parallel-merge (Chan et al.) update, float64, clip to
    +-clip_range — the arithmetic of reference src/utils.py:68-98.  O(num_envs * dim) per env
    step on the host; the device version is a 'next' row."""

    def __init__(self, size, clip_range: float = 5.0, eps: float = 1e-8):
        self.mean = np.zeros(size)
        self.var = np.ones(size)
        self.count = eps
        self.clip_range = clip_range

    def update(self, x):
        x = np.asarray(x)
        self._merge(x.mean(axis=0), x.var(axis=0), x.shape[0])

    def _merge(self, b_mean, b_var, b_count):
        n = self.count + b_count
        delta = b_mean - self.mean
        m2 = self.var * self.count + b_var * b_count + np.square(delta) * self.count * b_count / n
        self.mean = self.mean + delta * b_count / n
        self.var = m2 / n
        self.count = n

    # name used by the reference's callers
    _update_from_moments = _merge

    def normalize(self, x):
        z = (x - self.mean) / (np.sqrt(self.var) + 1e-8)
        return np.clip(z, -self.clip_range, self.clip_range)

    def save(self, path: str):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            yaml.dump({"mean": self.mean.tolist(), "var": self.var.tolist(),
                       "count": float(self.count), "clip_range": float(self.clip_range)}, fh)

    def load(self, path: str):
        with open(path, "r") as fh:
            d = yaml.safe_load(fh)
        self.mean = np.array(d["mean"], dtype=np.float32)
        self.var = np.array(d["var"], dtype=np.float32)
        self.count = float(d["count"])
        self.clip_range = float(d["clip_range"])


class DeviceRunningNormalizer:
    """RunningNormalizer living on the GPU (csrc/normalizer.hip): same constructor, `update`, `normalize`, `save`,
    `load`, `mean` / `var` / `count` / `clip_range` as the host class above (reference src/utils.py:68-117), same
    arithmetic bit for bit.  Assign it where the trainer assigns the host one (`buffer.obs_normalizer = ...`,
    src/env.py:93-98): the agents' fused entry points (`observe_act`, `process_step`) then keep a vector-env step's
    observation rows on the device from normalisation to the replay ring.
    `normalize` returns the float32-rounded values (as float64 arrays, like the reference) — the trainer casts the
    result to float32 before anything consumes it (src/env.py:189-190, src/agent.py:1353)."""

    def __init__(self, size, clip_range: float = 5.0, eps: float = 1e-8, device_index: int = 0):
        import ctypes as C
        from .. import _ffi
        self._ffi, self._C = _ffi, C
        self.size = int(size)
        self._clip = float(clip_range)
        self._h = _ffi.check_ptr(_ffi.lib.gcrl_normalizer_create(self.size, self._clip, float(eps), int(device_index)),
                                 "gcrl_normalizer_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._ffi.lib.gcrl_normalizer_destroy(h)

    @property
    def handle(self):
        return self._h

    def _state(self):
        C = self._C
        mean, var, cnt = np.empty(self.size), np.empty(self.size), C.c_double()
        self._ffi.check(self._ffi.lib.gcrl_normalizer_get(self._h, mean.ctypes.data, var.ctypes.data, C.byref(cnt)))
        return mean, var, cnt.value

    mean = property(lambda self: self._state()[0].astype(np.float32) if self.float32 else self._state()[0])
    var = property(lambda self: self._state()[1].astype(np.float32) if self.float32 else self._state()[1])
    count = property(lambda self: self._state()[2])
    clip_range = property(lambda self: self._clip)

    def set_state(self, mean, var, count, clip_range=None, float32=None):
        """float32: None = from the arrays' dtype (float32 arrays, as `load` makes them, select the reference's float32
        regime: after `RunningNormalizer.load` the reference normalises and merges in float32, src/utils.py:108-117)."""
        if float32 is None:
            float32 = np.asarray(mean).dtype == np.float32 and np.asarray(var).dtype == np.float32
        mean = np.ascontiguousarray(mean, np.float64).reshape(-1)
        var = np.ascontiguousarray(var, np.float64).reshape(-1)
        if clip_range is not None:
            self._clip = float(clip_range)
        self._ffi.check(self._ffi.lib.gcrl_normalizer_set(self._h, mean.ctypes.data, var.ctypes.data, float(count), self._clip))
        self._ffi.check(self._ffi.lib.gcrl_normalizer_set_float32(self._h, 1 if float32 else 0))

    @property
    def float32(self):
        return bool(self._ffi.lib.gcrl_normalizer_is_float32(self._h))

    def rows_dtype(self, dtype):
        """Tell the handle which dtype the rows have on the reference's side (numpy's type rules decide its arithmetic):
        float64 for the trainer's observation batches, float32 for goals and for tensors cast by the caller.  `update` and
        `normalize` call this with their argument's own dtype; the agents' fused entries with the dtype of the arrays they
        were given."""
        if dtype is getattr(self, "_rows_dt", None):       # (the same dtype object as last time: nothing to do)
            return
        self._rows_dt = dtype
        on = 1 if np.dtype(dtype) == np.float64 else 0
        if on != getattr(self, "_rows64", 0):
            self._ffi.check(self._ffi.lib.gcrl_normalizer_set_rows_float64(self._h, on))
            self._rows64 = on

    def _check_float32_valued(self, x):
        """The float64-rows regime restates numpy's float64 ARITHMETIC, but the rows travel to the device as float32: exact for
        panda-gym's observations (float32 values in a float64 array, reference src/utils.py:156), silently different for an
        environment that emits genuine float64 values.  Checked on the first float64 batch of every normaliser, warned
        about once."""
        if x.dtype == np.float64 and not getattr(self, "_f64_checked", False):
            self._f64_checked = True
            if not np.array_equal(x.astype(np.float32).astype(np.float64), x):
                import warnings
                warnings.warn("gcrl_amd: float64 observations that are not float32-valued: the device normaliser rounds rows to float32 before its "
                              "float64 arithmetic, so statistics and normalised values differ from the reference's in the last float32 bits "
                              "(INTEGRATION.md, rows_dtype)")

    def update(self, x):
        x = np.asarray(x)
        self.rows_dtype(x.dtype)
        self._check_float32_valued(x)
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim == 1:
            x = x[None, :]
        self._ffi.check(self._ffi.lib.gcrl_normalizer_update(self._h, x.ctypes.data, x.shape[0], x.shape[1], 0, self._ffi.stream_handle()))

    def normalize(self, x):
        x = np.asarray(x)
        self.rows_dtype(x.dtype)
        f32_result = self.float32 and x.dtype != np.float64     # (the reference's result dtype: float32 only for float32 rows on loaded statistics)
        x2 = np.ascontiguousarray(x.reshape(-1, self.size), np.float32)
        out = np.empty_like(x2)
        self._ffi.check(self._ffi.lib.gcrl_normalizer_normalize(self._h, x2.ctypes.data, x2.shape[0], self.size, 0, out.ctypes.data,
                                                                self.size, 0, self._ffi.stream_handle()))
        return (out if f32_result else out.astype(np.float64)).reshape(x.shape)

    def save(self, path: str):
        mean, var, count = self._state()
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            yaml.dump({"mean": mean.tolist(), "var": var.tolist(), "count": float(count), "clip_range": float(self._clip)}, fh)

    def load(self, path: str):
        with open(path, "r") as fh:
            d = yaml.safe_load(fh)
        # the reference keeps float32 arrays after load (src/utils.py:113-114) and computes in float32 from then on: so does
        # the handle (set_state sees the dtype)
        self.set_state(np.array(d["mean"], dtype=np.float32), np.array(d["var"], dtype=np.float32), float(d["count"]), float(d["clip_range"]))


def set_seed(seed: int, env=None):
    """Seeds the three host generators the reference seeds (src/utils.py:197-208)."""
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if env is not None:
        env.action_space.seed(seed)
        env.observation_space.seed(seed)
