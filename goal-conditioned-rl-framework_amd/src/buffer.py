"""HERBuffer — drop-in for the reference's class of the same name (src/buffer.py:92-179),
backed by the HBM replay ring of libgcrl_hip.so (csrc/her_ring.hip).

Same constructor, `push`, `sample`, `__len__` and externally assigned attributes
(`compute_reward`, `obs_normalizer`, `dg_normalizer`) as the reference, so src/env.py can use
it unchanged.  What differs is where things happen: transitions are staged in HBM, the HER
"future" relabelling + reward recomputation is one kernel at episode flush, and `sample` is a
gather kernel over host-drawn (CPython-exact Mersenne Twister) indices.
"""
from __future__ import annotations

import ctypes as C
import random as _pyrandom

import numpy as np
import torch

from .. import _ffi
from .._ffi import lib

FLUSH_LEN = 50  # literal in the reference (src/buffer.py:117); max_eps_len is not what triggers it


class MTStream:
    """CPython-exact MT19937 living in the library (csrc/mt19937_cpython.cc).

    mode "python": the generator mirrors Python's global `random` state around every use, so
    HER picks, batch draws and the reference's own random.random() calls (src/agent.py:1348)
    interleave exactly as in the reference.  mode "engine": a private stream seeded once —
    bit-identical to the reference as long as nothing else consumes `random` in between,
    without the ~30 us state round trip per call.  mode "device": no MT stream at all — HER future
    indices come from a counter hash evaluated inside the flush kernel (keyed by seed, episode number,
    pick number) and batch indices from a keyed permutation of the ring positions that the gather
    kernels evaluate per row (no host RNG, no index upload);
    NOT the reference's index stream (restated in oracle/her_oracle.py HashRng), for runs that do not
    need index-level parity with a reference run.
    """

    def __init__(self, mode: str = "python", seed: int | None = None):
        if mode not in ("python", "engine", "device"):
            raise ValueError(f"rng mode must be 'python', 'engine' or 'device', got {mode!r}")
        self.mode = mode
        self.seed_value = 0 if seed is None else int(seed)
        self.handle = _ffi.check_ptr(lib.gcrl_mt_create(), "gcrl_mt_create")
        self._buf = (C.c_uint32 * 625)()
        if mode == "engine":
            self.seed(0 if seed is None else seed)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            lib.gcrl_mt_destroy(h)

    def seed(self, s: int):
        _ffi.check(lib.gcrl_mt_seed(self.handle, int(s)))

    def pull(self):
        """python mode: copy random.getstate() into the library's generator."""
        if self.mode != "python":
            return
        _, words, self._gauss = _pyrandom.getstate()
        self._buf[:] = words
        _ffi.check(lib.gcrl_mt_set_state(self.handle, self._buf))

    def push_back(self):
        """python mode: write the advanced state back to the `random` module."""
        if self.mode != "python":
            return
        _ffi.check(lib.gcrl_mt_get_state(self.handle, self._buf))
        _pyrandom.setstate((3, tuple(self._buf), self._gauss))

    def random(self) -> float:
        if self.mode in ("python", "device"):   # device mode: the ring never touches an MT stream
            return _pyrandom.random()
        return float(lib.gcrl_mt_random(self.handle))


def _classify_reward(fn, goal_dim: int, default_threshold: float):
    """Map the injected compute_reward callable (src/env.py:105) to a built-in reward kind by
    probing it: sparse -(d > thr) with thr found by bisection, or dense -d."""
    if fn is None:
        return 0, float(default_threshold)
    try:
        return _classify_reward_probe(fn, goal_dim, default_threshold)
    except Exception:   # noqa: BLE001  a callable that cannot take the probe vectors is not a goal-distance reward: host path,
        return 2, float(default_threshold)   # where its exception surfaces from the push that calls it, as in the reference


def _classify_reward_probe(fn, goal_dim: int, default_threshold: float):
    zero = np.zeros(goal_dim, dtype=np.float32)

    def at(dist):
        g = np.zeros(goal_dim, dtype=np.float32)
        g[0] = dist
        return float(np.asarray(fn(zero, g, {})))

    def agrees(kind, thr):
        """The candidate built-in kind against the callable on random goal pairs in general position (the axis probes below
        cannot tell a distance reward from, say, an L1 or a shaped one)."""
        gen = np.random.default_rng(20240229)
        for _ in range(24):
            ag = gen.uniform(-0.3, 0.3, goal_dim).astype(np.float32)
            g = (ag + gen.standard_normal(goal_dim) * gen.choice([0.01, 0.05, 0.3])).astype(np.float32)
            d = float(np.linalg.norm(ag.astype(np.float64) - g.astype(np.float64)))
            got = float(np.asarray(fn(ag, g, {})))
            if kind == 0:
                if abs(d - thr) > 1e-5 * max(1.0, thr) and got != (-1.0 if d > thr else 0.0):
                    return False
            elif abs(got + d) > 1e-5 * max(1.0, d):
                return False
        return True

    probes = [0.0, 1e-3, 0.3, 5.0]
    vals = [at(p) for p in probes]
    if all(v in (0.0, -1.0) for v in vals) and vals[0] == 0.0 and vals[-1] == -1.0:
        lo, hi = 0.0, 5.0
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if at(mid) == 0.0:
                lo = mid
            else:
                hi = mid
        thr = float(np.float32(0.5 * (lo + hi)))
        # snap to the advertised threshold when the probe agrees with it
        if abs(thr - default_threshold) < 1e-6:
            thr = float(default_threshold)
        if agrees(0, thr):
            return 0, thr
    elif all(abs(v + p) <= 1e-6 * max(1.0, p) for v, p in zip(vals, probes)) and agrees(1, 0.0):
        return 1, float(default_threshold)
    # anything else: the reference calls whatever was injected (src/env.py:105, src/buffer.py:166) — so does the ring,
    # through the host-callback reward kind (include/gcrl.h GCRL_REWARD_HOST): picks and goal swap on the device, the
    # relabel rewards of a flush computed by `fn` on the host
    return 2, float(default_threshold)


class _RingState:
    """save_state / load_state shared by the three buffers (extension of SURVEY.md §8f-2)."""

    def save_state(self, path: str) -> dict:
        """Ring rows (logical order), staged partial episodes and the MT19937 stream -> `path`; returns the metadata
        load_state needs.  An untouched buffer (no transition pushed yet) saves as empty."""
        meta = dict(dims=self._dims, rng_mode=self.rng.mode, py_random=None, mt=None, bytes=0)
        if self.rng.mode == "python":
            st = _pyrandom.getstate()
            meta["py_random"] = [st[0], list(st[1]), st[2]]
        elif self.rng.mode == "engine":
            _ffi.check(lib.gcrl_mt_get_state(self.rng.handle, self.rng._buf))
            meta["mt"] = list(self.rng._buf)
        if self._h is not None:
            n = int(lib.gcrl_her_state_size(self._h))
            blob = np.empty(n, np.uint8)
            _ffi.check(lib.gcrl_her_save_state(self._h, blob.ctypes.data, n))
            blob.tofile(path)
            meta["bytes"] = n
        if hasattr(self, "priorities"):
            meta["priorities"] = [float(p) for p in self.priorities]
        return meta

    def load_state(self, path: str, meta: dict):
        if meta["bytes"]:
            self._ensure(*meta["dims"])
            blob = np.fromfile(path, dtype=np.uint8)
            _ffi.check(lib.gcrl_her_load_state(self._h, blob.ctypes.data, blob.size))
        if meta.get("py_random") is not None and self.rng.mode == "python":
            v, words, gauss = meta["py_random"]
            _pyrandom.setstate((v, tuple(words), gauss))
        if meta.get("mt") is not None and self.rng.mode == "engine":
            self.rng._buf[:] = meta["mt"]
            _ffi.check(lib.gcrl_mt_set_state(self.rng.handle, self.rng._buf))
        if hasattr(self, "priorities") and "priorities" in meta:
            self.priorities.clear()
            self.priorities.extend(np.float32(p) for p in meta["priorities"])


class HERBuffer(_RingState):
    def __init__(self, max_mem_len: int, max_eps_len: int, nenvs: int, threshold: float = 0.05,
                 k_future: int = 4, *, rng: str = "python", seed: int | None = None,
                 device_index: int = 0):
        if not torch.cuda.is_available() or lib.gcrl_device_count() <= 0:
            raise _ffi.GcrlError("HERBuffer needs a HIP device: the replay ring lives in HBM and "
                                 "there is no CPU fallback")
        if int(max_eps_len) < FLUSH_LEN:
            # the reference's staging deque(maxlen=max_eps_len) would silently drop the oldest transitions and never
            # reach the len >= 50 flush (src/buffer.py:102,117); the ring's staging always holds 50: refuse, do not diverge
            raise ValueError(f"max_eps_len={max_eps_len} < {FLUSH_LEN}: the reference flushes at the literal 50 "
                             "(src/buffer.py:117) and would drop staged transitions; not emulated")
        self.max_mem_len = int(max_mem_len)
        self.max_eps_len = int(max_eps_len)
        self.nenvs = int(nenvs)
        self.device = "cuda"
        self.device_index = device_index
        self._h = None
        self._threshold = threshold
        self.k_future = int(k_future)
        self._compute_reward = None
        self.obs_normalizer = None
        self.dg_normalizer = None
        self.rng = MTStream(rng, seed)
        self._dims = None
        self._reward_cfg = None

    # the trainer assigns compute_reward after construction (src/env.py:105); classification into a built-in reward
    # kind happens when the ring is created — a later reassignment that changes the kind / threshold is refused
    # rather than silently ignored
    @property
    def compute_reward(self):
        return self._compute_reward

    @compute_reward.setter
    def compute_reward(self, fn):
        self._compute_reward = fn
        self._recheck_reward()

    @property
    def threshold(self):
        return self._threshold

    @threshold.setter
    def threshold(self, v):
        self._threshold = v
        self._recheck_reward()

    def _recheck_reward(self):
        if self._h is None or self._reward_cfg is None:
            return
        now = _classify_reward(self._compute_reward, self._dims[2], self._threshold)
        if now[0] == 2 and self._reward_cfg[0] == 2:
            return      # host-callback ring: the callable in use is whatever is assigned now, as in the reference
        if now != self._reward_cfg:
            raise ValueError(f"compute_reward / threshold changed after the replay ring was created with reward config "
                             f"{self._reward_cfg} (now {now}): create a new buffer")

    # ------------------------------------------------------------------ handle management
    def _ensure(self, S: int, A: int, G: int):
        if self._h is not None:
            if self._dims != (S, A, G):
                raise ValueError(f"transition dims {(S, A, G)} differ from the ring's {self._dims}")
            return
        kind, thr = _classify_reward(self.compute_reward, G, self.threshold)
        self._reward_cfg = (kind, thr)
        cfg = _ffi.HerConfig(state_dim=S, action_dim=A, goal_dim=G, capacity=self.max_mem_len,
                             nenvs=self.nenvs, k_future=self.k_future, flush_len=FLUSH_LEN,
                             reward_kind=kind, reward_threshold=thr, device=self.device_index,
                             rng_mode=1 if self.rng.mode == "device" else 0, seed=self.rng.seed_value)
        self._h = _ffi.check_ptr(lib.gcrl_her_create(C.byref(cfg), self.rng.handle), "gcrl_her_create")
        self._dims = (S, A, G)
        if kind == 2:
            self._install_reward_callback()

    def _install_reward_callback(self):
        """GCRL_REWARD_HOST: compute_reward(ag_i, ag_f, {}) per relabelled row, one call per pair in the reference's order
        (src/buffer.py:166), float32 like the stored rewards (src/buffer.py:129).  An exception raised by the callable is
        kept and re-raised by the push that triggered the flush."""
        import weakref
        wself = weakref.ref(self)

        def cb(ag_p, goal_p, n, G, out_p, _user):
            me = wself()
            try:
                ag = np.ctypeslib.as_array(ag_p, shape=(n, G))
                goal = np.ctypeslib.as_array(goal_p, shape=(n, G))
                out = np.ctypeslib.as_array(out_p, shape=(n,))
                fn = me._compute_reward
                for i in range(n):
                    out[i] = np.float32(np.asarray(fn(ag[i].copy(), goal[i].copy(), {})))
                return 0
            except BaseException as e:   # noqa: BLE001  (must not propagate through the C frames)
                if me is not None:
                    me._reward_exc = e
                return 1

        self._reward_exc = None
        self._reward_cb = _ffi.REWARD_FN(cb)          # kept alive as long as the ring
        _ffi.check(lib.gcrl_her_set_reward_callback(self._h, C.cast(self._reward_cb, C.c_void_p), None))

    def _check_rows(self, rows: int) -> int:
        """Status of a push-like native call; a failure caused by the compute_reward callable re-raises ITS exception."""
        exc, self._reward_exc = getattr(self, "_reward_exc", None), None
        if exc is not None:
            raise exc
        return _ffi.check(int(rows))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.gcrl_her_destroy(h)

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return 0 if self._h is None else int(lib.gcrl_her_len(self._h))

    # ------------------------------------------------------------------ reference surface
    @staticmethod
    def _state_arg(x):
        """-> (keepalive, pointer, on_device) for a state given as cuda/cpu tensor or ndarray."""
        if isinstance(x, torch.Tensor):
            t = x.detach()
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            return t, t.data_ptr(), 1 if t.is_cuda else 0
        arr = np.ascontiguousarray(x, dtype=np.float32)
        return arr, arr.ctypes.data, 0

    def push(self, idx, state, action, next_state, reward, done, desired_goal, achieved_goal):
        act = np.ascontiguousarray(action, dtype=np.float32).reshape(-1)
        dg = np.ascontiguousarray(desired_goal, dtype=np.float32).reshape(-1)
        ag = np.ascontiguousarray(achieved_goal, dtype=np.float32).reshape(-1)
        ks, ps, ds = self._state_arg(state)
        kn, pn, dn = self._state_arg(next_state)
        S = int(ks.numel() if isinstance(ks, torch.Tensor) else ks.size)
        self._ensure(S, act.size, ag.size)
        flushing = bool(done) or lib.gcrl_her_staged(self._h, int(idx)) + 1 >= FLUSH_LEN
        if flushing:
            self.rng.pull()
        rows = lib.gcrl_her_push(self._h, int(idx), ps, ds, act.ctypes.data, pn, dn,
                                 float(reward), 1 if done else 0, dg.ctypes.data, ag.ctypes.data,
                                 _ffi.stream_handle())
        self._check_rows(rows)
        if flushing:
            self.rng.push_back()

    def push_batch(self, states, actions, next_states, rewards, dones, achieved_goals, env0: int = 0):
        """One vector-env step: the `for i in range(num_envs): push_her(i, ...)` loop of the reference's
        trainer (src/env.py:192-201) as ONE call.  states / next_states: [n, S] cuda tensors (the
        obs_batch / next_obs_batch that _process_step builds); the rest host arrays.  Same rows, same
        order and same RNG draws as n push() calls; envs finishing together are flushed together."""
        s = states.detach().to(device="cuda", dtype=torch.float32).contiguous()
        ns = next_states.detach().to(device="cuda", dtype=torch.float32).contiguous()
        a = np.ascontiguousarray(actions, dtype=np.float32)
        r = np.ascontiguousarray(rewards, dtype=np.float32).reshape(-1)
        d = np.ascontiguousarray(np.asarray(dones).astype(np.uint8)).reshape(-1)
        ag = np.ascontiguousarray(achieved_goals, dtype=np.float32)
        n = s.shape[0]
        assert a.shape[0] == n and ns.shape[0] == n and r.size == n and d.size == n and ag.shape[0] == n
        self._ensure(s.shape[1], a.shape[1], ag.shape[1])
        self.rng.pull()
        rows = lib.gcrl_her_push_batch(self._h, int(env0), n, s.data_ptr(), s.shape[1], a.ctypes.data, ns.data_ptr(),
                                       ns.shape[1], r.ctypes.data, d.ctypes.data, ag.ctypes.data, _ffi.stream_handle())
        self._check_rows(rows)
        self.rng.push_back()
        return int(rows)

    def push_episode(self, idx, states, actions, next_states, rewards, dones, achieved_goals):
        """Whole-episode fast path (one upload + one flush launch); same result as T push calls."""
        s = np.ascontiguousarray(states, dtype=np.float32)
        a = np.ascontiguousarray(actions, dtype=np.float32)
        ns = np.ascontiguousarray(next_states, dtype=np.float32)
        r = np.ascontiguousarray(rewards, dtype=np.float32).reshape(-1)
        d = np.ascontiguousarray(dones, dtype=np.float32).reshape(-1)
        ag = np.ascontiguousarray(achieved_goals, dtype=np.float32)
        T = s.shape[0]
        self._ensure(s.shape[1], a.shape[1], ag.shape[1])
        self.rng.pull()
        rows = lib.gcrl_her_push_episode(self._h, int(idx), T, s.ctypes.data, a.ctypes.data,
                                         ns.ctypes.data, r.ctypes.data, d.ctypes.data,
                                         ag.ctypes.data, None, _ffi.stream_handle())
        self._check_rows(rows)
        self.rng.push_back()
        return int(rows)

    def sample(self, batch_size: int, num_batches: int = 1, indices=None, return_indices: bool = False):
        assert len(self) >= batch_size, "[ERROR] Not enough in buffer to sample"
        S, A, _ = self._dims
        n = batch_size * num_batches
        dev = torch.device("cuda", self.device_index)
        states = torch.empty((n, S), dtype=torch.float32, device=dev)
        actions = torch.empty((n, A), dtype=torch.float32, device=dev)
        rewards = torch.empty((n, 1), dtype=torch.float32, device=dev)
        next_states = torch.empty((n, S), dtype=torch.float32, device=dev)
        dones = torch.empty((n, 1), dtype=torch.float32, device=dev)
        idx_in = None
        if indices is not None:
            idx_in = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
            assert idx_in.size == n
        drawn = np.empty(n, dtype=np.uint32) if return_indices else None
        if idx_in is None:
            self.rng.pull()
        _ffi.check(lib.gcrl_her_sample(
            self._h, batch_size, num_batches, idx_in.ctypes.data if idx_in is not None else None,
            states.data_ptr(), S, actions.data_ptr(), A, rewards.data_ptr(), next_states.data_ptr(), S,
            dones.data_ptr(), drawn.ctypes.data if drawn is not None else None, _ffi.stream_handle()))
        if idx_in is None:
            self.rng.push_back()
        out = (states, actions, rewards, next_states, dones)
        return out + (drawn,) if return_indices else out

    def rows(self, first: int = 0, count: int | None = None):
        """Test helper: ring rows in logical (oldest-first) order as numpy arrays."""
        n = len(self) - first if count is None else count
        S, A, _ = self._dims
        s = np.empty((n, S), np.float32); a = np.empty((n, A), np.float32)
        ns = np.empty((n, S), np.float32); r = np.empty(n, np.float32); d = np.empty(n, np.float32)
        _ffi.check(lib.gcrl_her_read_rows(self._h, first, n, s.ctypes.data, a.ctypes.data,
                                          ns.ctypes.data, r.ctypes.data, d.ctypes.data))
        return s, a, ns, r, d

    def compute_termination(self, dg, ag):
        return np.linalg.norm(dg - ag, axis=-1) < self.threshold


class ReplayBuffer(_RingState):
    """Drop-in for the reference's ReplayBuffer (src/buffer.py:8-35): FIFO rows in the HBM ring (same packed record as
    HERBuffer, no staging, no relabelling), `sample` = `random.sample` over the stored rows (CPython-exact index stream)
    + the gather kernel.  `push(state, action, reward, next_state, done)` appends ONE row at once."""

    def __init__(self, max_len: int, *, rng: str = "python", seed: int | None = None, device_index: int = 0):
        if not torch.cuda.is_available() or lib.gcrl_device_count() <= 0:
            raise _ffi.GcrlError(f"{type(self).__name__} needs a HIP device: the replay ring lives in HBM and there is no CPU fallback")
        self.max_len = int(max_len)
        self.device = "cuda"
        self.device_index = device_index
        self.rng = MTStream(rng, seed)
        self._h = None
        self._dims = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.gcrl_her_destroy(h)

    @property
    def handle(self):
        return self._h

    def __len__(self):
        return 0 if self._h is None else int(lib.gcrl_her_len(self._h))

    def _ensure(self, S: int, A: int):
        if self._h is not None:
            if self._dims != (S, A):
                raise ValueError(f"transition dims {(S, A)} differ from the ring's {self._dims}")
            return
        cfg = _ffi.HerConfig(state_dim=S, action_dim=A, goal_dim=1, capacity=self.max_len, nenvs=1, k_future=0,
                             flush_len=FLUSH_LEN, reward_kind=0, reward_threshold=0.0, device=self.device_index,
                             rng_mode=1 if self.rng.mode == "device" else 0, seed=self.rng.seed_value)
        self._h = _ffi.check_ptr(lib.gcrl_her_create(C.byref(cfg), self.rng.handle), "gcrl_her_create")
        self._dims = (S, A)

    def push(self, state, action, reward, next_state, done):
        act = np.ascontiguousarray(action, dtype=np.float32).reshape(-1)
        ks, ps, ds = HERBuffer._state_arg(state)
        kn, pn, dn = HERBuffer._state_arg(next_state)
        S = int(ks.numel() if isinstance(ks, torch.Tensor) else ks.size)
        self._ensure(S, act.size)
        _ffi.check(int(lib.gcrl_her_append(self._h, ps, ds, act.ctypes.data, float(reward), pn, dn, 1 if done else 0,
                                           _ffi.stream_handle())))

    def _gather(self, batch_size: int, indices=None):
        S, A = self._dims
        dev = torch.device("cuda", self.device_index)
        out = (torch.empty((batch_size, S), dtype=torch.float32, device=dev), torch.empty((batch_size, A), dtype=torch.float32, device=dev),
               torch.empty((batch_size, 1), dtype=torch.float32, device=dev), torch.empty((batch_size, S), dtype=torch.float32, device=dev),
               torch.empty((batch_size, 1), dtype=torch.float32, device=dev))
        idx = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        if idx is None:
            self.rng.pull()
        _ffi.check(lib.gcrl_her_sample(self._h, batch_size, 1, idx.ctypes.data if idx is not None else None, out[0].data_ptr(), S,
                                       out[1].data_ptr(), A, out[2].data_ptr(), out[3].data_ptr(), S, out[4].data_ptr(), None,
                                       _ffi.stream_handle()))
        if idx is None:
            self.rng.push_back()
        return out

    def sample(self, batch_size: int):
        assert len(self) >= batch_size, "Not enough in buffer to sample"
        return self._gather(batch_size)

    def rows(self, first: int = 0, count: int | None = None):
        n = len(self) - first if count is None else count
        S, A = self._dims
        s = np.empty((n, S), np.float32); a = np.empty((n, A), np.float32)
        ns = np.empty((n, S), np.float32); r = np.empty(n, np.float32); d = np.empty(n, np.float32)
        _ffi.check(lib.gcrl_her_read_rows(self._h, first, n, s.ctypes.data, a.ctypes.data, ns.ctypes.data, r.ctypes.data, d.ctypes.data))
        return s, a, ns, r, d


class PERBuffer(ReplayBuffer):
    """Drop-in for the reference's proportional PERBuffer (src/buffer.py:38-89).  Rows live in the HBM ring; the
    priorities and the draw stay on the host in numpy, operation for operation as the reference writes them
    (`np.random.choice(N, B, p=P)` consumes numpy's global stream; float32 priorities / weights), because the
    priority update needs the per-sample |td| on the host anyway (one read-back per step, as in the reference)."""

    def __init__(self, max_len: int, alpha: float, **kw):
        super().__init__(max_len, **kw)
        from collections import deque
        self.priorities = deque(maxlen=int(max_len))
        self.alpha = alpha
        self.epsilon = 1e-6

    def push(self, state, action, reward, next_state, done):
        super().push(state, action, reward, next_state, done)
        self.priorities.append(1.0)

    def draw(self, batch_size: int, beta: float):
        """-> (indices, weights float32 [B]) exactly as src/buffer.py:50-65."""
        assert len(self) >= batch_size, "Not enough in buffer to sample"
        N = len(self)
        P = np.array(self.priorities, dtype=np.float32)
        P_sum = P.sum()
        if P_sum > 0:
            P /= P_sum
        else:
            P[:] = 1.0 / N
        indices = np.random.choice(N, batch_size, p=P)
        weights = (N * P[indices]) ** (-beta)
        weights /= weights.max()
        return indices, weights

    def sample(self, batch_size: int, beta: float):
        indices, weights = self.draw(batch_size, beta)
        batch = self._gather(batch_size, indices)
        w = torch.as_tensor(weights, dtype=torch.float32).unsqueeze(-1).to(batch[0].device)
        return batch + (w, indices)

    def update_priorities(self, indices, priorities):
        priorities = np.asarray(priorities).squeeze(-1)
        for index, priority in zip(indices, priorities):
            self.priorities[index] = (abs(priority) + self.epsilon) ** self.alpha
