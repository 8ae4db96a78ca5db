"""DDPG / TD3Agent / SACAgent / TQCAgent — drop-ins for the reference's classes of the same
names (src/agent.py), backed by the update engine of libgcrl_hip.so (csrc/agent.hip).

Constructor signature `(obs_dim, ac_dim, config, weights, nenvs, gradient_step)` and the
methods the trainer calls (src/env.py: select_action, push_her, update(step), is_buffer_filled,
update_normalizers, normalize_*, save_weights, reset, set_train/set_eval) are the reference's;
`update` returns the same tuple shapes (DDPG 6/4, TD3 8/6, SAC & TQC 9/6 entries) because the
trainer dispatches on the length (src/env.py:448-506).  The entries are lazily materialised
scalars by default (float(x), np.asarray(x), x.item() all work) so that a run of updates is
enqueued without a host sync per step; `sync_metrics=True` returns plain floats with `td_error`
as a 0-d numpy array, exactly like the reference (src/agent.py:1342).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import weakref

import numpy as np
import torch

from .. import _ffi
from .._ffi import lib
from .buffer import HERBuffer, PERBuffer, ReplayBuffer
from .model import Actor, Critic, SACActorModel

KIND = {"DDPG": 0, "TD3": 1, "SAC": 2, "TQC": 3}


class LazyScalar:
    """One entry of an update()'s return tuple, fetched from the device on first use."""
    __slots__ = ("_agent", "_ticket", "_n", "_i", "_as_array", "_val", "__weakref__")

    def __init__(self, agent, ticket, n, i, as_array=False):
        self._agent, self._ticket, self._n, self._i, self._as_array = agent, ticket, n, i, as_array
        self._val = None

    def _value(self) -> float:
        if self._val is None:   # kept: the engine's metric slots are a ring, a trainer may hold this object for long
            self._val = self._agent._metrics(self._ticket, self._n)[self._i]
            self._agent = None
        return self._val

    def __float__(self):
        return float(self._value())

    def item(self):
        return float(self._value())

    def __array__(self, dtype=None, copy=None):
        # td_error is a 0-d float32 array in the reference (src/agent.py:1342), the other entries Python floats
        return np.asarray(self._value(), dtype=dtype or (np.float32 if self._as_array else np.float64))

    def __repr__(self):
        return f"LazyScalar({self._value()!r})"

    def _bin(op):
        def f(self, other):
            return op(float(self), float(other))
        return f

    __add__ = _bin(lambda a, b: a + b)
    __radd__ = _bin(lambda a, b: b + a)
    __sub__ = _bin(lambda a, b: a - b)
    __rsub__ = _bin(lambda a, b: b - a)
    __mul__ = _bin(lambda a, b: a * b)
    __rmul__ = _bin(lambda a, b: b * a)
    __truediv__ = _bin(lambda a, b: a / b)
    __lt__ = _bin(lambda a, b: a < b)
    __gt__ = _bin(lambda a, b: a > b)
    del _bin


class _AlphaView:
    """`agent.alpha.item()` (src/env.py:574,604)."""

    def __init__(self, agent):
        self._agent_ref = weakref.ref(agent)   # (no reference cycle: the agent must die with its last user reference)

    def item(self):
        a = self._agent_ref()
        if a is None:
            raise ReferenceError("the agent of this alpha view is gone")
        if not a._sac:
            raise AttributeError("alpha")
        buf = np.empty(1, np.float32)
        _ffi.check(lib.gcrl_agent_get(a._h, b"alpha", buf.ctypes.data, 1))
        return float(buf[0])


class _EngineAgent:
    KIND_NAME = "DDPG"
    TD_INDEX = {6: 2, 4: 1}  # position of td_error in the tuple, by tuple length

    def __init__(self, obs_dim: int, ac_dim: int, config, weights, nenvs: int, gradient_step: int, *,
                 use_graph: bool = True, pipeline: bool = True, sync_metrics: bool = False, rng: str = "python",
                 seed: int | None = None, device_index: int = 0, num_critics: int = 5,
                 top_quantiles_to_drop: int = 2, n_quantiles: int = 1):
        if not torch.cuda.is_available() or lib.gcrl_device_count() <= 0:
            raise _ffi.GcrlError(f"{type(self).__name__} needs a HIP device; there is no CPU fallback")
        self.device = "cuda"
        self.device_index = device_index
        self.config = config
        self.gradient_step = int(gradient_step)
        self.obs_dim, self.ac_dim = int(obs_dim), int(ac_dim)
        self.sync_metrics = sync_metrics
        kind = KIND[self.KIND_NAME]
        self._sac = kind >= 2

        # buffer factory of the reference (src/agent.py:1214-1228 and its copies)
        if config.buffer_type == "PER":
            self.buffer = PERBuffer(config.max_len, config.alpha, rng=rng, seed=seed, device_index=device_index)
            pipeline = 0      # importance-sampling weights enter the loss in the layer-per-launch schedule
        elif config.buffer_type == "REPLAY":
            self.buffer = ReplayBuffer(config.max_len, rng=rng, seed=seed, device_index=device_index)
        elif config.buffer_type == "HER":
            self.buffer = HERBuffer(config.max_len, config.max_eps_len, nenvs, k_future=config.k_future,
                                    rng=rng, seed=seed, device_index=device_index)
        else:
            raise ValueError(f"[ERROR] Invalid Buffer type. Received {config.buffer_type}.")

        # the YAML keys num_critics / top_quantiles_to_drop are dropped by pydantic in the
        # reference, so getattr(config, ..., 5/2) always yields the defaults (src/agent.py:789-790)
        self.num_critics = int(getattr(config, "num_critics", num_critics)) if kind == 3 else (1 if kind == 0 else 2)
        self.top_quantiles_to_drop = int(getattr(config, "top_quantiles_to_drop", top_quantiles_to_drop))
        # n_quantiles > 1: the DISTRIBUTIONAL TQC variant of BASELINE.json configs[3] (no reference counterpart: the reference's
        # critics are scalar; include/gcrl.h gcrl_agent_config.n_quantiles, oracle/quantile_tqc_oracle.py)
        self.n_quantiles = int(n_quantiles) if kind == 3 else 1
        if self.n_quantiles > 1 and kind == 3:
            self.num_critics = int(num_critics)

        cfg = _ffi.AgentConfig(
            kind=kind, obs_dim=obs_dim, ac_dim=ac_dim, hidden_dim=config.hidden_dim,
            layer_count=config.layer_count, batch_size=config.batch_size,
            num_critics=self.num_critics, top_drop=self.top_quantiles_to_drop if kind == 3 else 0,
            ac_update_freq=config.ac_update_freq, gradient_step=self.gradient_step, polyak_every=40,
            gamma=config.gamma, tau=config.tau,
            grad_clip=-1.0 if config.grad_clip is None else float(config.grad_clip),
            policy_noise=config.policy_noise, noise_clamp=config.noise_clamp,
            actor_lr=config.actor_lr, actor_lr_min=config.actor_lr_min,
            critic_lr=config.critic_lr, critic_lr_min=config.critic_lr_min,
            alpha_lr=float(getattr(config, "alpha_lr", 0.0003)),
            ac_scheduler_steps=config.ac_scheduler_steps, cr_scheduler_steps=config.cr_scheduler_steps,
            alpha_min_steps=float(getattr(config, "alpha_min_steps", 10000)),
            device=device_index, use_graph=int(use_graph), pipeline_steps=(2 if pipeline is True else int(pipeline)),
            n_quantiles=self.n_quantiles,
            seed=0 if seed is None else int(seed))
        self._h = _ffi.check_ptr(lib.gcrl_agent_create(C.byref(cfg)), "gcrl_agent_create")
        self._metric_cache: dict[int, list[float]] = {}
        self._live = collections.deque()      # (ticket, n) of returned tuples, oldest first
        self._lazy: dict[int, list] = {}      # ticket -> weak references to its unresolved LazyScalars

        self.noise_std = config.noise_std
        self.noise_clamp = config.noise_clamp
        self.policy_noise = config.policy_noise
        self.gamma = config.gamma
        self.batch_size = config.batch_size
        self.ac_update_freq = config.ac_update_freq
        self.grad_clip = config.grad_clip
        self.tau = config.tau
        self.beta = config.beta
        self.beta_start = config.beta
        self.beta_max = 1.0
        self.beta_end = config.beta_end
        self.alpha_min = getattr(config, "alpha_min", 0.05)
        self.alpha_min_steps = getattr(config, "alpha_min_steps", 10000)
        self.alpha = _AlphaView(self)

        # The network views reach the native handle through a weak reference: a closure over `self` made every agent part
        # of a reference cycle, so dropping one left its HBM ring, pinned blocks and streams alive until the cyclic
        # collector happened to run — and a process building agents one after another (sweeps) ran the next one 60 %
        # slower beside the undead (tools/rowchain_vs_batch.py found it).
        wself = weakref.ref(self)

        def get():
            a = wself()
            return a._h if a is not None else None
        H, L = config.hidden_dim, config.layer_count
        if self._sac:
            self.actor = SACActorModel(get, "actor", obs_dim, H, ac_dim, L)
        else:
            self.actor = Actor(get, "actor", obs_dim, H, ac_dim, L)
            self.target_actor = Actor(get, "target_actor", obs_dim, H, ac_dim, L)
        self.critics = [Critic(get, f"critic_{i}", obs_dim + ac_dim, H, self.n_quantiles, L) for i in range(self.num_critics)]
        self.target_critics = [Critic(get, f"target_critic_{i}", obs_dim + ac_dim, H, self.n_quantiles, L)
                               for i in range(self.num_critics)]
        self._bind_names()
        if weights:
            self._load_weights(weights)
        self.update_target_network()

    # reference attribute names (critic / critic_1 / critic_2 ...)
    def _bind_names(self):
        pass

    def _load_weights(self, weights: str):
        raise NotImplementedError

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.gcrl_agent_destroy(h)

    def set_meetings(self, on: bool = True) -> int:
        """Switch the launch forms whose workgroups wait for each other inside a kernel (csrc/meet.h) on (where the device
        admits them) or off; returns the bit mask of the forms now active.  Off is the safe setting whenever the GPU is
        shared with other processes or streams (`gcrl_set_shared_device` / GCRL_SHARED_GPU=1 do it process-wide)."""
        return _ffi.check(lib.gcrl_agent_set_meetings(self._h, 1 if on else 0))

    def meetings(self) -> int:
        """Bit mask of the launch forms with in-kernel waits that are active on this handle (1 BatchNorm slab row groups, 2 row-chain
        roles, 4 the opt-in weight-slice DDPG launch, 8 the fused dW + optimiser launch of the row-chain DDPG step); changes nothing."""
        return _ffi.check(lib.gcrl_agent_get_meetings(self._h))

    # ------------------------------------------------------------------ update
    def _metrics(self, ticket: int, n: int):
        vals = self._metric_cache.get(ticket)
        if vals is None:
            buf = (C.c_double * n)()
            _ffi.check(lib.gcrl_agent_metrics(self._h, ticket, buf, n))
            vals = list(buf)
            if len(self._metric_cache) > 2048:
                self._metric_cache.clear()
            self._metric_cache[ticket] = vals
        return vals

    def _tuple(self, ticket: int, n: int):
        td = self.TD_INDEX[n]
        self._live.append((ticket, n))
        # the engine keeps 4096 metric records: resolve what the caller still holds before its slot comes round again
        while self._live and self._live[0][0] <= ticket - 2048:
            t0, n0 = self._live.popleft()
            for ref in self._lazy.pop(t0, ()):
                ls = ref()
                if ls is not None:
                    ls._value()
        if self.sync_metrics:
            vals = self._metrics(ticket, n)
            return tuple(np.asarray(v, dtype=np.float32) if i == td else v for i, v in enumerate(vals))
        out = tuple(LazyScalar(self, ticket, n, i, i == td) for i in range(n))
        self._lazy[ticket] = [weakref.ref(x) for x in out]
        return out

    def beta_scheduler(self, step: int):
        ratio = step / self.beta_end
        self.beta = min(self.beta_max, self.beta_start + ratio * (self.beta_max - self.beta_start))

    @staticmethod
    def _inject(batch=None, noise=None, eps_next=None, eps_cur=None):
        """-> (gcrl_update_inputs or None, tensors to keep alive) for an explicit batch / injected noise."""
        if batch is None and noise is None and eps_next is None and eps_cur is None:
            return None, []
        inputs, keep = _ffi.UpdateInputs(), []

        def dev(t):
            t = torch.as_tensor(t).to(device="cuda", dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        if batch is not None:
            s, a, r, ns, d = (dev(t) for t in batch)
            inputs.s_dev, inputs.ld_s = s.data_ptr(), s.shape[1]
            inputs.a_dev, inputs.ld_a = a.data_ptr(), a.shape[1]
            inputs.r_dev, inputs.d_dev = r.data_ptr(), d.data_ptr()
            inputs.ns_dev, inputs.ld_ns = ns.data_ptr(), ns.shape[1]
        if noise is not None:
            inputs.noise_dev = dev(noise).data_ptr()
        if eps_next is not None:
            inputs.eps_next_dev = dev(eps_next).data_ptr()
        if eps_cur is not None:
            inputs.eps_cur_dev = dev(eps_cur).data_ptr()
        return inputs, keep

    def update(self, step: int, *, batch=None, noise=None, eps_next=None, eps_cur=None):
        """agent.update(step).  Keyword arguments inject an explicit batch / noise (parity tests):
        batch = (states, actions, rewards, next_states, dones) cuda float32 tensors."""
        self.set_train()
        inputs, keep = self._inject(batch, noise, eps_next, eps_cur)
        her = None
        per = isinstance(self.buffer, PERBuffer) and batch is None
        if per:   # src/agent.py:1380-1387: the prioritised draw (host, numpy-exact), weights into the critic losses
            indices, weights = self.buffer.draw(self.batch_size, self.beta)
            if inputs is None:
                inputs = _ffi.UpdateInputs()
            idx32 = np.ascontiguousarray(indices, dtype=np.uint32)
            w32 = np.ascontiguousarray(weights, dtype=np.float32)
            keep += [idx32, w32]
            inputs.idx_host, inputs.weights_host = idx32.ctypes.data, w32.ctypes.data
        if batch is None:
            her = self.buffer.handle
            assert her is not None and len(self.buffer) >= self.batch_size, "[ERROR] Not enough in buffer to sample"
            self.buffer.rng.pull()
        ticket = C.c_int64(-1)
        n = _ffi.check(lib.gcrl_agent_update(self._h, her, int(step), C.byref(inputs) if inputs is not None else None,
                                             C.byref(ticket), _ffi.stream_handle()))
        if batch is None:
            self.buffer.rng.push_back()
        if per:   # update_priorities(indices, td_error) with the per-sample |td| (src/agent.py:1387); td_error stays an array
            td = np.empty(self.batch_size, np.float32)
            _ffi.check(lib.gcrl_agent_get(self._h, b"td_abs", td.ctypes.data, td.size))
            td = td.reshape(-1, 1)
            self.buffer.update_priorities(indices=indices, priorities=td)
        self.beta_scheduler(step)
        if self._sac:
            self.actor.num_batches_tracked += 1 + (1 if n == 9 else 0)
        out = self._tuple(ticket.value, n)
        if per:
            i = self.TD_INDEX[n]
            out = out[:i] + (td,) + out[i + 1:]
        return out

    def update_many(self, step0: int, n: int):
        """The trainer's `for _ in range(gradient_step): update(step)` loop (src/env.py:384-385) as
        one call: all n batches are drawn (same RNG stream order) and gathered by one launch."""
        if isinstance(self.buffer, PERBuffer):   # the priorities change after every step: one draw per step (src/agent.py:1380-1387)
            return [self.update(step0 + i) for i in range(n)]
        self.set_train()
        her = self.buffer.handle
        assert her is not None and len(self.buffer) >= self.batch_size, "[ERROR] Not enough in buffer to sample"
        tickets = (C.c_int64 * n)()
        lens = (C.c_int32 * n)()
        self.buffer.rng.pull()
        _ffi.check(lib.gcrl_agent_update_n(self._h, her, int(step0), int(n), tickets, lens, _ffi.stream_handle()))
        self.buffer.rng.push_back()
        self.beta_scheduler(step0 + n - 1)
        if self._sac:
            self.actor.num_batches_tracked += sum(1 + (1 if l == 9 else 0) for l in lens)
        return [self._tuple(int(t), int(l)) for t, l in zip(tickets, lens)]

    # ------------------------------------------------------------------ acting
    def _actor_forward(self, obs, eps=None) -> torch.Tensor:
        if eps is None and not self._sac and not (isinstance(obs, torch.Tensor) and obs.is_cuda):
            # host rows in, host actions out, one native call (staging, both copies and the sync inside)
            x = np.ascontiguousarray(np.asarray(obs, dtype=np.float32))
            if x.ndim == 1:
                x = x[None, :]
            if x.shape[0] <= int(self.config.batch_size):
                out = np.empty((x.shape[0], self.ac_dim), dtype=np.float32)
                _ffi.check(lib.gcrl_agent_act_host(self._h, x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data,
                                                   self.ac_dim, _ffi.stream_handle()))
                return torch.from_numpy(out)
        obs_t = torch.as_tensor(np.asarray(obs), dtype=torch.float32).to("cuda").contiguous()
        if obs_t.dim() == 1:
            obs_t = obs_t.unsqueeze(0)
        out = torch.empty((obs_t.shape[0], self.ac_dim), dtype=torch.float32, device="cuda")
        _ffi.check(lib.gcrl_agent_act(self._h, obs_t.data_ptr(), obs_t.shape[0], obs_t.shape[1], out.data_ptr(),
                                      self.ac_dim, eps.data_ptr() if eps is not None else None,
                                      _ffi.stream_handle()))
        return out

    # ------------------------------------------------------------------ fused acting side
    def _device_normalizers(self, obs_normalize: bool, g_normalize: bool):
        """(obs handle, goal handle) when the normalisers this step needs live on the device, else None."""
        from .utils import DeviceRunningNormalizer
        on, gn = getattr(self.buffer, "obs_normalizer", None), getattr(self.buffer, "dg_normalizer", None)
        if obs_normalize and not isinstance(on, DeviceRunningNormalizer):
            return None
        if g_normalize and not isinstance(gn, DeviceRunningNormalizer):
            return None
        return (on.handle if obs_normalize else None, gn.handle if g_normalize else None)

    def _rows_dtypes(self, obs_dtype, goal_dtype, obs_normalize: bool, g_normalize: bool):
        """numpy's type rules decide the reference normaliser's arithmetic: tell the device normalisers which dtype the rows
        have on the caller's side (the trainer's observation batches are float64 arrays, its goal batches float32)."""
        if obs_normalize:
            self.buffer.obs_normalizer.rows_dtype(obs_dtype)
        if g_normalize:
            self.buffer.dg_normalizer.rows_dtype(goal_dtype)

    _ACT_MODE_EXPLORE, _ACT_MODE_EVAL = 1, 0      # DDPG; TD3 overrides eval (raw network output)

    def _staging(self, tag: str, key: tuple, shapes):
        """Persistent host staging arrays of the fused acting entries (fixed addresses: their ctypes pointers are built once — a
        fresh numpy array per argument and per call cost more host time than the native call itself, tools/acting_breakdown.py).
        `shapes()` -> {name: (shape, dtype)} is only called when `key` (the dims) changed."""
        cache = self.__dict__.setdefault("_stage_cache", {})
        st = cache.get(tag)
        if st is None or st[0] != key:
            arrs = {k: np.empty(shp, dt) for k, (shp, dt) in shapes().items()}
            st = cache[tag] = (key, arrs, {k: C.c_void_p(v.ctypes.data) for k, v in arrs.items()})
        return st[1], st[2]

    def observe_act(self, observation, desired_goal, eval_action: bool = False, obs_normalize: bool = True,
                    g_normalize: bool = False):
        """`select_action(normalize_state_batch(obs, dg, ...))` for one vector-env step (src/env.py:348-355) as ONE native
        call: raw rows up, normalisation + actor + exploration noise on the device, float64 actions back.  The host
        generators are consumed exactly as in the reference (`random.random()` for DDPG's epsilon branch, `np.random`
        for the Gaussian noise, torch's for SAC / TQC eps).  Falls back to the two separate calls when the normalisers
        are host objects."""
        nz = self._device_normalizers(obs_normalize, g_normalize)
        obs_in = observation if isinstance(observation, np.ndarray) else np.asarray(observation)
        dg_in = desired_goal if isinstance(desired_goal, np.ndarray) else np.asarray(desired_goal)
        n = obs_in.shape[0]
        if nz is None or n > int(self.config.batch_size) or obs_in.ndim != 2 or dg_in.ndim != 2:
            obs, dg = np.ascontiguousarray(obs_in, np.float32), np.ascontiguousarray(dg_in, np.float32)
            return self.select_action(self.normalize_state_batch(obs, dg, obs_normalize, g_normalize), eval_action)
        self.set_eval()
        self._rows_dtypes(obs_in.dtype, dg_in.dtype, obs_normalize, g_normalize)
        noise, mode = self._act_noise(n, eval_action)
        if mode is None:
            return noise                              # DDPG's epsilon-random action: no network involved
        D, G, A = obs_in.shape[1], dg_in.shape[1], self.ac_dim
        buf, ptr = self._staging("act", (n, D, G, A), lambda: dict(obs=((n, D), np.float32), dg=((n, G), np.float32),
                                                                    noise=((n, A), np.float64), out=((n, A), np.float64)))
        np.copyto(buf["obs"], obs_in, casting="unsafe")
        np.copyto(buf["dg"], dg_in, casting="unsafe")
        if noise is not None:
            np.copyto(buf["noise"], noise)
        _ffi.check(lib.gcrl_agent_observe_act(self._h, nz[0], nz[1], ptr["obs"], D, ptr["dg"], G, n,
                                              ptr["noise"] if noise is not None else None, mode, ptr["out"], _ffi.stream_handle()))
        return buf["out"].copy()

    def _act_noise(self, n, eval_action):
        """-> (noise or None, mode); DDPG overrides for the epsilon branch."""
        if eval_action:
            return None, self._ACT_MODE_EVAL
        return np.ascontiguousarray(np.random.normal(0, self.noise_std, size=(n, self.ac_dim))), self._ACT_MODE_EXPLORE

    def process_step(self, state, actions, next_obs_raw, rewards, dones, obs_normalize: bool = True, g_normalize: bool = False):
        """The reference trainer's `_process_step` (src/env.py:163-201) for one vector-env step as ONE native call:
        normaliser update with [obs ; next_obs], both normalised state matrices built on the device from the updated
        statistics, all envs pushed (episode flush + HER relabel on the device when an env finishes).  `state` /
        `next_obs_raw`: the env's dict observations; `dones` = `terminated` (src/env.py:372).  With `g_normalize` the goal
        normaliser is updated from [dg ; next_dg ; ag ; next_ag] and goals are normalised on the device too (src/env.py:167-175,
        :222-223).  Falls back to the separate calls when a normaliser this step needs is a host object (or, with
        g_normalize, when compute_reward runs through the host callback)."""
        nz = self._device_normalizers(obs_normalize, g_normalize)
        buf = self.buffer
        if nz is not None and g_normalize:   # (the ring — and with it the reward kind — exists from here on)
            buf._ensure(np.shape(state["observation"])[1] + np.shape(state["desired_goal"])[1], np.shape(actions)[1],
                        np.shape(next_obs_raw["achieved_goal"])[1])
        if nz is None or (g_normalize and buf._reward_cfg[0] == 2):
            self.update_normalizers([state["observation"], next_obs_raw["observation"]],
                                    [state["desired_goal"], next_obs_raw["desired_goal"], state["achieved_goal"],
                                     next_obs_raw["achieved_goal"]], obs_normalize, g_normalize)
            s = torch.from_numpy(self.normalize_state_batch(state["observation"], state["desired_goal"], obs_normalize, g_normalize)).float().cuda()
            ns = torch.from_numpy(self.normalize_state_batch(next_obs_raw["observation"], next_obs_raw["desired_goal"], obs_normalize, g_normalize)).float().cuda()
            return buf.push_batch(s, actions, ns, rewards, dones, self.normalize_goal(next_obs_raw["achieved_goal"], g_normalize))
        as_arr = lambda x: x if isinstance(x, np.ndarray) else np.asarray(x)
        obs_i, nobs_i = as_arr(state["observation"]), as_arr(next_obs_raw["observation"])
        dg_i, ndg_i, nag_i = as_arr(state["desired_goal"]), as_arr(next_obs_raw["desired_goal"]), as_arr(next_obs_raw["achieved_goal"])
        ag_i = as_arr(state["achieved_goal"]) if g_normalize else None
        act_i = as_arr(actions)
        # np.concatenate's result type, as the reference's update_normalizers forms it (src/agent.py:343-350)
        odt = obs_i.dtype if obs_i.dtype == nobs_i.dtype else np.result_type(obs_i.dtype, nobs_i.dtype)
        gdt = dg_i.dtype
        if g_normalize and not (dg_i.dtype == ndg_i.dtype == ag_i.dtype == nag_i.dtype):
            gdt = np.result_type(dg_i.dtype, ndg_i.dtype, ag_i.dtype, nag_i.dtype)
        self._rows_dtypes(odt, gdt, obs_normalize, g_normalize)
        n, D, G, A = obs_i.shape[0], obs_i.shape[1], dg_i.shape[1], act_i.shape[1]
        buf._ensure(D + G, A, nag_i.shape[1])
        f32 = np.float32
        b, p = self._staging("proc", (n, D, G, A), lambda: dict(obs=((n, D), f32), nobs=((n, D), f32), dg=((n, G), f32), ndg=((n, G), f32),
                                                                ag=((n, G), f32), nag=((n, G), f32), act=((n, A), f32), rew=((n,), f32),
                                                                dn=((n,), np.uint8)))
        cp = np.copyto
        cp(b["obs"], obs_i, casting="unsafe"); cp(b["nobs"], nobs_i, casting="unsafe"); cp(b["dg"], dg_i, casting="unsafe")
        cp(b["ndg"], ndg_i, casting="unsafe"); cp(b["nag"], nag_i, casting="unsafe"); cp(b["act"], act_i, casting="unsafe")
        cp(b["rew"], rewards if (isinstance(rewards, np.ndarray) and rewards.ndim == 1) else np.reshape(rewards, -1), casting="unsafe")
        cp(b["dn"], dones if (isinstance(dones, np.ndarray) and dones.ndim == 1) else np.reshape(dones, -1), casting="unsafe")
        if g_normalize:
            np.copyto(b["ag"], ag_i, casting="unsafe")
        buf.rng.pull()
        rows = lib.gcrl_her_process_step_g(buf.handle, nz[0], 1 if obs_normalize else 0, nz[1], 1 if g_normalize else 0, p["obs"],
                                           p["nobs"], D, p["dg"], p["ndg"], p["ag"] if g_normalize else None, p["nag"], p["act"], p["rew"],
                                           p["dn"], 0, n, _ffi.stream_handle())
        buf._check_rows(rows)
        buf.rng.push_back()
        return int(rows)

    def push(self, state, action, reward, next_state, done):
        # (with a HERBuffer the reference passes 5 args to a push that takes 8 -> TypeError there too)
        self.buffer.push(state, action, reward, next_state, done)

    def push_her(self, idx, state, action, next_state, reward, done, desired_goal, achieved_goal):
        self.buffer.push(idx, state, action, next_state, reward, done, desired_goal, achieved_goal)

    def is_buffer_filled(self):
        return len(self.buffer) >= self.batch_size

    def set_train(self):
        self.actor.train()
        for c in self.critics:
            c.train()

    def set_eval(self):
        self.actor.eval()
        for c in self.critics:
            c.eval()

    # ------------------------------------------------------------------ normalisers (host, src/agent.py:1425-1459)
    def update_normalizers(self, obs_list, dg_list, obs_normalize, g_normalize):
        if hasattr(self.buffer, "obs_normalizer") and obs_list and obs_normalize:
            self.buffer.obs_normalizer.update(np.concatenate(obs_list, axis=0))
        if hasattr(self.buffer, "dg_normalizer") and dg_list and g_normalize:
            self.buffer.dg_normalizer.update(np.concatenate(dg_list, axis=0))

    def normalize_obs(self, obs, normalize: bool):
        if hasattr(self.buffer, "obs_normalizer") and normalize:
            return self.buffer.obs_normalizer.normalize(obs)
        return obs

    def normalize_goal(self, goal, normalize: bool):
        if hasattr(self.buffer, "dg_normalizer") and normalize:
            return self.buffer.dg_normalizer.normalize(goal)
        return goal

    def normalize_state_batch(self, obs_batch, dg_batch, obs_normalize, g_normalize):
        return np.concatenate([self.normalize_obs(obs_batch, obs_normalize),
                               self.normalize_goal(dg_batch, g_normalize)], axis=-1)

    # ------------------------------------------------------------------ targets / reset
    def update_target_network(self, hard_update: bool = True, tau: float = 0.005):
        """src/agent.py:1255-1271: hard copy, or tau*net + (1-tau)*target for every target network."""
        if hard_update:
            _ffi.check(lib.gcrl_agent_hard_update_targets(self._h))
        else:
            _ffi.check(lib.gcrl_agent_soft_update_targets(self._h, float(tau), _ffi.stream_handle()))

    # ------------------------------------------------------------------ full resume state (SURVEY.md §8f-2 extension)
    def save_state(self, path: str):
        """Everything a bitwise-identical continuation needs, which the reference's checkpoints (weights + normalisers,
        src/env.py:430-440) do not hold: parameters and targets, Adam moments, scheduler positions and step counts,
        BatchNorm statistics, alpha, the replay ring's rows + staged partial episodes, the MT19937 stream."""
        import json
        os.makedirs(path, exist_ok=True)
        n = int(lib.gcrl_agent_state_size(self._h))
        blob = np.empty(n, np.uint8)
        _ffi.check(lib.gcrl_agent_save_state(self._h, blob.ctypes.data, n))
        blob.tofile(os.path.join(path, "agent.bin"))
        meta = dict(kind=self.KIND_NAME, beta=self.beta, num_batches_tracked=int(self.actor.num_batches_tracked),
                    ring=self.buffer.save_state(os.path.join(path, "ring.bin")))
        for name in ("obs_normalizer", "dg_normalizer"):
            nz = getattr(self.buffer, name, None)
            if nz is not None:
                meta[name] = dict(mean=np.asarray(nz.mean, np.float64).tolist(), var=np.asarray(nz.var, np.float64).tolist(),
                                  count=float(nz.count), clip_range=float(nz.clip_range),
                                  float32=bool(np.asarray(nz.mean).dtype == np.float32))   # (the regime after RunningNormalizer.load)
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f)

    def load_state(self, path: str):
        import json
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        if meta["kind"] != self.KIND_NAME:      # checked before anything of this agent is overwritten
            raise ValueError(f"state of a {meta['kind']} agent loaded into a {self.KIND_NAME}")
        blob = np.fromfile(os.path.join(path, "agent.bin"), dtype=np.uint8)
        _ffi.check(lib.gcrl_agent_load_state(self._h, blob.ctypes.data, blob.size))
        self.beta = meta["beta"]
        self.actor.num_batches_tracked = meta["num_batches_tracked"]
        self.buffer.load_state(os.path.join(path, "ring.bin"), meta["ring"])
        for name in ("obs_normalizer", "dg_normalizer"):
            nz = getattr(self.buffer, name, None)
            if name in meta and nz is not None:
                d = meta[name]
                if hasattr(nz, "set_state"):     # DeviceRunningNormalizer: its statistics live on the device
                    nz.set_state(np.array(d["mean"]), np.array(d["var"]), d["count"], d["clip_range"], float32=bool(d.get("float32", False)))
                else:
                    dt = np.float32 if d.get("float32", False) else np.float64
                    nz.mean, nz.var = np.array(d["mean"], dtype=dt), np.array(d["var"], dtype=dt)
                    nz.count, nz.clip_range = d["count"], d["clip_range"]
        self._metric_cache.clear()

    def reset(self):
        """Re-initialise Linear layers (src/agent.py:1461-1465, :760-769)."""
        seed = int(torch.randint(0, 2**31 - 1, (1,)).item())
        _ffi.check(lib.gcrl_agent_init_weights(self._h, seed, 1 if self._sac else 0))

    def get_gradient_norm(self, model) -> float:
        g = model.grad_flat().astype(np.float64)
        return float(np.sqrt(np.sum(g * g)))


class DDPG(_EngineAgent):
    KIND_NAME = "DDPG"
    TD_INDEX = {6: 2, 4: 1}

    def _bind_names(self):
        self.critic, self.target_critic = self.critics[0], self.target_critics[0]

    def _load_weights(self, weights):
        self.actor.load(os.path.join(weights, "actor.pth"))
        p = os.path.join(weights, "critic.pth")
        self.critic.load(p if os.path.exists(p) else os.path.join(weights, "critic_1.pth"))

    def select_action(self, obs_tensor, eval_action: bool = False):
        self.set_eval()
        if not eval_action:
            if self.buffer.rng.random() < 0.2:  # src/agent.py:1348 — shares the HER stream
                return np.clip(np.random.randn(np.asarray(obs_tensor).shape[0], self.ac_dim), a_min=-1, a_max=1)
            action = torch.tanh(self._actor_forward(obs_tensor)).cpu().numpy()  # double tanh, as the reference
            return np.clip(action + np.random.normal(0, self.noise_std, size=action.shape), -1, 1)
        return np.clip(torch.tanh(self._actor_forward(obs_tensor)).cpu().numpy(), -1, 1)

    def _act_noise(self, n, eval_action):
        if not eval_action and self.buffer.rng.random() < 0.2:      # src/agent.py:1348
            return np.clip(np.random.randn(n, self.ac_dim), a_min=-1, a_max=1), None
        return super()._act_noise(n, eval_action)

    def save_weights(self, path: str):
        self.actor.save(os.path.join(path, "actor.pth"))
        self.critic.save(os.path.join(path, "critic.pth"))


class TD3Agent(_EngineAgent):
    KIND_NAME = "TD3"
    TD_INDEX = {8: 3, 6: 2}
    _ACT_MODE_EVAL = 2      # eval returns the network output as it is (src/agent.py:265-270)

    def _bind_names(self):
        self.critic_1, self.critic_2 = self.critics
        self.target_critic_1, self.target_critic_2 = self.target_critics

    def _load_weights(self, weights):
        self.actor.load(os.path.join(weights, "actor.pth"))
        self.critic_1.load(os.path.join(weights, "critic_1.pth"))
        self.critic_2.load(os.path.join(weights, "critic_2.pth"))

    def select_action(self, obs_tensor, eval_action: bool = False):
        self.set_eval()
        if not eval_action:
            action = torch.tanh(self._actor_forward(obs_tensor)).cpu().numpy()
            return np.clip(action + np.random.normal(0, self.noise_std, size=action.shape), -1, 1)
        return self._actor_forward(obs_tensor).cpu().numpy()

    def save_weights(self, path: str):
        self.actor.save(os.path.join(path, "actor.pth"))
        self.critic_1.save(os.path.join(path, "critic_1.pth"))
        self.critic_2.save(os.path.join(path, "critic_2.pth"))


class _StochasticAgent(_EngineAgent):
    TD_INDEX = {9: 3, 6: 2}

    def _act_noise(self, n, eval_action):
        if eval_action:
            return None, 2
        # rsample's eps (src/model.py:134); drawn on the host generator here — one fewer device round trip
        return np.ascontiguousarray(torch.randn((n, self.ac_dim), dtype=torch.float32).numpy().astype(np.float64)), 2

    def select_action(self, obs_tensor, eval_action: bool = False):
        self.set_eval()
        eps = None
        if not eval_action:
            n = np.asarray(obs_tensor).reshape(-1, self.obs_dim).shape[0]
            # rsample's eps (src/model.py:134: loc + empty(shape).normal_() * scale on the module's device).  Drawn from
            # torch's HOST generator here — what the reference itself uses on a CPU box, and one device round trip less
            eps = torch.randn((n, self.ac_dim), dtype=torch.float32).cuda()
        return self._actor_forward(obs_tensor, eps).cpu().numpy()

    def _log_alpha_tensor(self):
        buf = np.empty(1, np.float32)
        _ffi.check(lib.gcrl_agent_get(self._h, b"log_alpha", buf.ctypes.data, 1))
        return torch.tensor(buf, requires_grad=True)

    @property
    def log_alpha(self):
        return self._log_alpha_tensor()

    def _save_log_alpha(self, path):
        torch.save(self._log_alpha_tensor(), os.path.join(path, "log_alpha.pth"))


class SACAgent(_StochasticAgent):
    KIND_NAME = "SAC"

    def __init__(self, obs_dim, ac_dim, config, weights, nenvs, gradient_step, **kw):
        super().__init__(obs_dim, ac_dim, config, weights, nenvs, gradient_step, **kw)
        self.target_entropy = -ac_dim * 0.5
        self.train_alpha = False

    def _bind_names(self):
        self.critic_1, self.critic_2 = self.critics
        self.target_critic_1, self.target_critic_2 = self.target_critics

    def _load_weights(self, weights):
        self.actor.load(os.path.join(weights, "actor.pth"))
        self.critic_1.load(os.path.join(weights, "critic_1.pth"))
        self.critic_2.load(os.path.join(weights, "critic_2.pth"))

    def save_weights(self, path: str):
        self.actor.save(os.path.join(path, "actor.pth"))
        self.critic_1.save(os.path.join(path, "critic_1.pth"))
        self.critic_2.save(os.path.join(path, "critic_2.pth"))
        self._save_log_alpha(path)


class TQCAgent(_StochasticAgent):
    KIND_NAME = "TQC"

    def __init__(self, obs_dim, ac_dim, config, weights, nenvs, gradient_step, **kw):
        super().__init__(obs_dim, ac_dim, config, weights, nenvs, gradient_step, **kw)
        self.target_entropy = -ac_dim

    def _load_weights(self, weights):
        self.actor.load(os.path.join(weights, "actor.pth"))
        for i, c in enumerate(self.critics):
            p = os.path.join(weights, f"critic_{i}.pth")
            if os.path.exists(p):
                c.load(p)
        p = os.path.join(weights, "log_alpha.pth")
        if os.path.exists(p):
            la = torch.load(p, map_location="cpu").detach().float().reshape(-1).numpy()
            _ffi.check(lib.gcrl_agent_set(self._h, b"log_alpha", la.ctypes.data, 1))

    def save_weights(self, path: str):
        self.actor.save(os.path.join(path, "actor.pth"))
        for i, c in enumerate(self.critics):
            c.save(os.path.join(path, f"critic_{i}.pth"))
        self._save_log_alpha(path)
