"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (gcrl_amd / libgcrl_hip.so).

CPU restatement of the reference's HER replay path, used as the checker in tests/, in
__graft_entry__.smoke() and as the timed `cpu_baseline` ("port") in bench.py.

Pinned against the real reference: tests/golden/*.npz hold outputs of the reference's own
`HERBuffer` (imported from /root/reference by tests/golden/make_golden.py in the build
container); tests/test_oracle_golden.py replays them through this file bit-for-bit.
The reward function is the one third-party piece (panda-gym `Task.compute_reward`, version
unpinned in the reference's requirements.txt:11, not present anywhere): restated from its
published definition `-(||ag - g||_2 > 0.05)` as float32 -> that boundary is "parity unpinned".

Each function cites the reference lines it follows (paths under /root/reference).
"""
from __future__ import annotations

import random
from collections import deque

import numpy as np

FLUSH_LEN = 50  # src/buffer.py:117 (literal)


def sparse_reward(achieved_goal, desired_goal, info=None, threshold: float = 0.05):
    """panda-gym sparse task reward as the reference receives it (src/env.py:105): 0-d/ND
    float32, -1.0 when farther than the threshold, -0.0 on success."""
    d = np.linalg.norm(np.asarray(achieved_goal) - np.asarray(desired_goal), axis=-1)
    return -np.array(d > threshold, dtype=np.float32)


def dense_reward(achieved_goal, desired_goal, info=None):
    d = np.linalg.norm(np.asarray(achieved_goal) - np.asarray(desired_goal), axis=-1)
    return -d.astype(np.float32)


class HERBufferOracle:
    """src/buffer.py:92-179.  Stored row = (s, a, ns, r, d, dg, ag) exactly like the reference."""

    def __init__(self, max_mem_len, max_eps_len, nenvs, threshold=0.05, k_future=4, rng=random):
        self.rows = deque(maxlen=max_mem_len)                              # :101
        self.staged = [deque(maxlen=max_eps_len) for _ in range(nenvs)]    # :102
        self.threshold = threshold
        self.k_future = k_future
        self.compute_reward = sparse_reward
        self.rng = rng  # module `random` or a random.Random instance
        self.future_log: list[int] = []

    def __len__(self):
        return len(self.rows)

    def push(self, idx, state, action, next_state, reward, done, desired_goal, achieved_goal):
        """:110-119 — stage, flush on done or at 50 staged transitions."""
        self.staged[idx].append((np.asarray(state, dtype=np.float32), action, np.asarray(next_state, dtype=np.float32),
                                 reward, done, desired_goal, achieved_goal))
        if done or len(self.staged[idx]) >= FLUSH_LEN:
            self._relabel_and_store(idx)
            self.staged[idx].clear()

    def _relabel_and_store(self, idx):
        """:143-179 — original row, then k_future 'future' relabels for every step but the last."""
        episode = self.staged[idx]
        T = len(episode)
        if hasattr(self.rng, "begin_episode"):      # HashRng: picks are keyed by episode number
            self.rng.begin_episode()
        for i, (s, a, ns, r, d, dg, ag) in enumerate(episode):
            self.rows.append((s, a, ns, r, d, dg, ag))
            if i >= T - 1:
                # the reference still loops k times here but its `if i < eps_len - 1` draws nothing
                continue
            for _ in range(self.k_future):
                f = self.rng.randint(i + 1, T - 1)                      # :153
                self.future_log.append(f)
                goal = np.array(episode[f][6], dtype=np.float32)        # :154-156 future achieved goal
                g = goal.shape[0]
                s2 = np.concatenate([s[:-g], goal], axis=-1)            # :159-164 goal = last g entries
                ns2 = np.concatenate([ns[:-g], goal], axis=-1)
                r2 = self.compute_reward(ag, episode[f][6], {})         # :166
                self.rows.append((s2, a, ns2, r2, False, goal, ag))     # :167-179

    def draw(self, batch_size):
        """The index-free draw of :124 (random.sample over the deque)."""
        assert len(self.rows) >= batch_size, "[ERROR] Not enough in buffer to sample"
        return self.rng.sample(self.rows, batch_size)

    def sample(self, batch_size):
        """:121-135 — numpy collation; returns float32 arrays shaped like the reference's tensors."""
        picked = self.draw(batch_size)
        s, a, ns, r, d, _, _ = zip(*picked)
        return (np.array(s, dtype=np.float32), np.array(a, dtype=np.float32),
                np.array(r, dtype=np.float32).reshape(-1, 1), np.array(ns, dtype=np.float32),
                np.array(d, dtype=np.float32).reshape(-1, 1))

    def as_arrays(self):
        """All stored rows, oldest first: (s, a, ns, r, d)."""
        if not self.rows:
            return tuple(np.zeros((0,), np.float32) for _ in range(5))
        s, a, ns, r, d, _, _ = zip(*self.rows)
        return (np.array(s, np.float32), np.array(a, np.float32), np.array(ns, np.float32),
                np.array(r, np.float32), np.array(d, np.float32))


# ---- index streams (what bit-exactness is asserted on) -------------------------------------

def future_indices(rng, T: int, k_future: int):
    """Draw order of :146-153."""
    return [rng.randint(i + 1, T - 1) for i in range(T - 1) for _ in range(k_future)]


def sample_indices(rng, n: int, k: int):
    """Indices random.sample(population of length n, k) would pick (CPython Lib/random.py)."""
    return rng.sample(range(n), k)


# ---- device-RNG ("fast") mode restatement: csrc/her_ring.h mix64/hash_below -------------------
_M = (1 << 64) - 1


def mix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & _M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def hash_below(seed: int, stream: int, ctr: int, n: int) -> int:
    h = mix64((mix64(seed ^ ((stream * 0xD1342543DE82EF95) & _M)) + ctr) & _M)
    return ((h >> 32) * n) >> 32


class HashRng:
    """The engine's device-RNG mode (GCRL_RNG_DEVICE) as a `random`-like object for HERBufferOracle:
    randint() = the flush kernel's future pick (csrc/her_ring.hip her_flush_kernel: stream = episode
    number, counter = pick number inside the episode), sample() = the batch draw the gather kernels compute
    per row (csrc/her_ring.h idxgen_at: a keyed Feistel permutation of the ring positions)."""

    def __init__(self, seed: int):
        self.seed = int(seed)
        self.episode = -1
        self.pick = 0
        self.draws = 0

    def begin_episode(self):
        self.episode += 1
        self.pick = 0

    def randint(self, a: int, b: int) -> int:
        v = a + hash_below(self.seed, self.episode, self.pick, b - a + 1)
        self.pick += 1
        return v

    def sample(self, population, k: int):
        """Batch element t of draw d = feistel_index(seed, d, n, t) (csrc/her_ring.h): a keyed permutation of
        [0, n), so the k picks are distinct by construction."""
        n = len(population)
        out = [feistel_index(self.seed, self.draws, n, t) for t in range(k)]
        self.draws += 1
        return [population[j] for j in out]


def feistel_index(seed: int, draw: int, n: int, t: int) -> int:
    bits = 2
    while bits < 32 and (1 << bits) < n:
        bits += 1
    hb = (bits + 1) >> 1
    mask = (1 << hb) - 1
    key = mix64((seed ^ ((draw * 0xD1342543DE82EF95) & _M) ^ 0x5BD1E995) & _M)
    x = t
    while True:
        L, R = (x >> hb) & mask, x & mask
        for r in range(4):
            f = (mix64((key + r * 0x9E3779B97F4A7C15 + R) & _M) >> 32) & mask
            L, R = R, L ^ f
        x = (L << hb) | R
        if x < n:
            return x


def synthetic_episode(rng: np.random.Generator, T: int, S: int, A: int, G: int = 3):
    """SURVEY.md §8d synthetic transitions: obs ~ N(0,1) with a t/50 time feature, goal fixed per
    episode, achieved goal = random walk (so relabel rewards mix -1 / -0)."""
    dg = rng.uniform(-0.15, 0.15, size=G).astype(np.float32)
    ag = rng.uniform(-0.15, 0.15, size=G).astype(np.float32)
    obs_dim = S - G
    steps = []
    obs = rng.standard_normal(obs_dim).astype(np.float32)
    for t in range(T):
        obs[-1] = t / 50.0
        s = np.concatenate([obs, dg]).astype(np.float32)
        a = rng.uniform(-1, 1, size=A).astype(np.float32)
        ag = (ag + rng.normal(0, 0.02, size=G)).astype(np.float32)
        nobs = rng.standard_normal(obs_dim).astype(np.float32)
        nobs[-1] = (t + 1) / 50.0
        ns = np.concatenate([nobs, dg]).astype(np.float32)
        r = float(sparse_reward(ag, dg))
        steps.append((s, a, ns, np.float64(r), False, dg.copy(), ag.copy()))
        obs = nobs
    return steps
