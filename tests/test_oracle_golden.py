"""CPU tests: the oracle (oracle/*.py) reproduces the golden vectors captured from the REAL
reference (tests/golden/make_golden.py).  Index streams and stored rows bit-exact; update-step
losses / gradients within 1e-5 relative (abs floor 1e-6), the north-star tolerance."""
import random

import numpy as np
import pytest
import torch

from conftest import hparams_from_golden, load_golden
from oracle import her_oracle
from oracle.agent_oracle import OracleAgent

RTOL, ATOL = 1e-5, 1e-6


def close(a, b, rtol=RTOL, atol=ATOL, scale=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    floor = atol + rtol * (np.max(np.abs(b)) if scale is None else scale)
    return np.max(np.abs(a - b)) <= floor if a.size else True


# ---------------------------------------------------------------- G1/G2: index streams
def test_future_index_streams_match_reference():
    g = load_golden("her_index_streams.npz")
    for key in [k for k in g.files if k.startswith("future_") and not k.endswith("_state")]:
        _, s, T, k = key.split("_")
        rng = random.Random(int(s[1:]))
        got = her_oracle.future_indices(rng, int(T[1:]), int(k[1:]))
        assert got == g[key].tolist(), key


def test_sample_index_streams_match_reference():
    g = load_golden("her_index_streams.npz")
    for key in [k for k in g.files if k.startswith("sample_") and not k.endswith("_state")]:
        _, s, n, b = key.split("_")
        rng = random.Random(int(s[1:]))
        got = her_oracle.sample_indices(rng, int(n[1:]), int(b[1:]))
        assert got == g[key].tolist(), key
        assert np.array_equal(np.array(rng.getstate()[1], dtype=np.uint32), g[key + "_state"])


# ---------------------------------------------------------------- G3/G4: stored rows, batch
CASES = ["full50", "done12", "single", "wrap300", "k8_two_envs", "tiny_cap100"]


def replay_case(g, name, buf):
    e = 0
    while f"{name}_ep{e}_s" in g.files:
        env = int(g[f"{name}_ep{e}_env"][0])
        done_last = bool(g[f"{name}_ep{e}_done_last"][0])
        s, a, ns = g[f"{name}_ep{e}_s"], g[f"{name}_ep{e}_a"], g[f"{name}_ep{e}_ns"]
        r, dg, ag = g[f"{name}_ep{e}_r"], g[f"{name}_ep{e}_dg"], g[f"{name}_ep{e}_ag"]
        T = s.shape[0]
        for t in range(T):
            buf.push(env, s[t], a[t], ns[t], r[t], bool(done_last and t == T - 1), dg[t], ag[t])
        e += 1


@pytest.mark.parametrize("name", CASES)
def test_her_rows_match_reference(name):
    g = load_golden("her_rows.npz")
    cap, k = (int(x) for x in g[f"{name}_cap_k"])
    rng = random.Random(1898)
    buf = her_oracle.HERBufferOracle(cap, 50, 2, k_future=k, rng=rng)
    replay_case(g, name, buf)
    s, a, ns, r, d = buf.as_arrays()
    for key, got in zip("s a ns r d".split(), (s, a, ns, r, d)):
        want = g[f"{name}_rows_{key}"]
        assert got.shape == want.shape, (name, key)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, key)  # bitwise (-0.0 too)
    assert np.array_equal(np.array(rng.getstate()[1], dtype=np.uint32), g[f"{name}_state_after_push"])
    if f"{name}_batch_s" in g.files:
        batch = buf.sample(32)
        for key, got in zip("s a r ns d".split(), batch):
            assert np.array_equal(got.view(np.uint32), g[f"{name}_batch_{key}"].view(np.uint32)), (name, key)
        assert np.array_equal(np.array(rng.getstate()[1], dtype=np.uint32), g[f"{name}_state_after_sample"])


def test_row_count_and_order_properties():
    """T + k(T-1) rows per episode, originals at stride 1+k, relabels keep s[:-G] and done=0."""
    rng = random.Random(5)
    gen = np.random.default_rng(0)
    buf = her_oracle.HERBufferOracle(10000, 50, 1, k_future=4, rng=rng)
    steps = her_oracle.synthetic_episode(gen, 50, 10, 3)
    for st in steps:
        buf.push(0, *st)
    assert len(buf) == 50 + 4 * 49
    s, a, ns, r, d = buf.as_arrays()
    for i in range(49):
        base = i * 5
        assert np.array_equal(s[base], steps[i][0])
        for j in range(1, 5):
            assert np.array_equal(s[base + j][:-3], steps[i][0][:-3])
            assert np.array_equal(a[base + j], steps[i][1])
            assert d[base + j] == 0.0
            assert r[base + j] in (-1.0, 0.0)
            f = buf.future_log[i * 4 + j - 1]
            assert i < f <= 49
            assert np.array_equal(s[base + j][-3:], steps[f][6])


# ---------------------------------------------------------------- G5-G8: update steps
def run_update_fixture(tag):
    g = load_golden(f"update_{tag}.npz")
    kind = str(g["kind"][0])
    S, A, B, gstep = (int(x) for x in g["dims"])
    cfg = hparams_from_golden(g)
    torch.set_num_threads(1)
    ag = OracleAgent(kind, S, A, cfg, nenvs=1, gradient_step=gstep)
    nets = {"actor": ag.actor}
    if ag.target_actor is not None:
        nets["target_actor"] = ag.target_actor
    for i, (c, t) in enumerate(zip(ag.critics, ag.target_critics)):
        nets[f"critic_{i}"] = c
        nets[f"target_critic_{i}"] = t
    for name, net in nets.items():
        key = f"init_{name}"
        src = g[key] if key in g.files else g[f"init_{name.replace('target_', '')}"]
        ag.set_flat_params(net, src)
    checks = []
    for i, step in enumerate(g["steps"]):
        batch = tuple(torch.from_numpy(g[f"step{i}_{k}"]) for k in ("s", "a", "r", "ns", "d"))
        kw = {}
        if f"step{i}_noise" in g.files:
            kw["noise"] = torch.from_numpy(g[f"step{i}_noise"])
        if f"step{i}_eps_next" in g.files:
            kw["eps_next"] = torch.from_numpy(g[f"step{i}_eps_next"])
            kw["eps_cur"] = torch.from_numpy(g[f"step{i}_eps_cur"])
        info = ag.update(int(step), batch=batch, **kw)
        want = g[f"step{i}_tuple"]
        assert len(info) == len(want), (tag, i)
        got = np.array([float(np.asarray(x)) for x in info])
        checks.append((f"tuple step{i}", got, want, None))
        for c in range(len(ag.critics)):
            k = f"step{i}_gradpre_critic_{c}"
            if k in g.files:
                checks.append((k, ag.last["critic_grads_pre"][c], g[k], None))
            k = f"step{i}_gradpost_critic_{c}"
            if k in g.files:
                checks.append((k, ag.last["critic_grads_post"][c], g[k], None))
        k = f"step{i}_gradpre_actor"
        if k in g.files:
            checks.append((k, ag.last["actor_grads_pre"], g[k], None))
            if f"step{i}_gradpost_actor" in g.files:
                checks.append((f"step{i}_gradpost_actor", ag.last["actor_grads_post"], g[f"step{i}_gradpost_actor"], None))
        for name, net in nets.items():
            k = f"step{i}_param_{name}"
            if k in g.files:
                checks.append((k, ag.flat_params(net), g[k], None))
        if f"step{i}_log_alpha" in g.files:
            checks.append((f"step{i}_log_alpha", ag.log_alpha.detach().numpy().copy(), g[f"step{i}_log_alpha"], 1.0))
            bns = [m for m in ag.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
            checks.append((f"step{i}_bn_mean", np.concatenate([m.running_mean.numpy() for m in bns]), g[f"step{i}_bn_mean"], None))
            checks.append((f"step{i}_bn_var", np.concatenate([m.running_var.numpy() for m in bns]), g[f"step{i}_bn_var"], None))
    return checks


@pytest.mark.parametrize("tag", ["ddpg_reach", "ddpg_cosine", "ddpg_pickplace_h256", "td3", "sac", "tqc"])
def test_oracle_update_matches_reference(tag):
    bad = []
    for name, got, want, scale in run_update_fixture(tag):
        if name.startswith("tuple"):
            ok = all(abs(a - b) <= ATOL + RTOL * abs(b) for a, b in zip(got, want))
        else:
            ok = close(got, want, scale=scale)
        if not ok:
            bad.append((name, float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))))))
    assert not bad, bad


def test_device_rng_batch_draw_is_a_permutation_prefix():
    """Device-RNG batch draws (oracle restatement of csrc/her_ring.h feistel_index): within a draw the picks
    are distinct and in range for any population size (power of two or not, barely above the batch), different
    draws differ, and over many draws the picks are spread evenly over the population."""
    from oracle.her_oracle import feistel_index
    for n, B in ((256, 256), (257, 256), (1000, 64), (5000, 256), (1_000_000, 256), (3, 2)):
        for draw in (0, 1, 12345):
            picks = [feistel_index(77, draw, n, t) for t in range(B)]
            assert len(set(picks)) == B and min(picks) >= 0 and max(picks) < n
        if n > 8:
            assert [feistel_index(77, 0, n, t) for t in range(B)] != [feistel_index(77, 1, n, t) for t in range(B)]
    n, B, draws = 1000, 50, 400
    counts = np.zeros(n)
    for d in range(draws):
        for t in range(B):
            counts[feistel_index(5, d, n, t)] += 1
    expect = draws * B / n                      # 20 per position
    assert abs(counts.mean() - expect) < 1e-9
    assert counts.min() >= 3 and counts.max() <= 45 and abs(counts.std() - np.sqrt(expect)) < 1.5   # ~Poisson(20)


def test_synthetic_workload_helpers_agree_with_the_oracle_copies(gcrl):
    """bench.py / examples build their synthetic inputs and configs with the package's own helpers (nothing
    outside tests, smoke() and the CPU-baseline leg touches oracle/); the oracle keeps separate copies for the
    tests — same streams, same defaults."""
    from gcrl_amd.src import synthetic
    from oracle import agent_oracle, her_oracle
    a = synthetic.synthetic_episode(np.random.default_rng(3), 50, 13, 4)
    b = her_oracle.synthetic_episode(np.random.default_rng(3), 50, 13, 4)
    for x, y in zip(a, b):
        for u, v in zip(x, y):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    for kind in ("DDPG", "TD3", "SAC", "TQC"):
        assert vars(synthetic.agent_config(kind, batch_size=7)) == vars(agent_oracle.make_config(kind, batch_size=7))


# ---------------------------------------------------------------- G9: RunningNormalizer
def test_normalizer_oracle_matches_reference_bitwise():
    from oracle.normalizer_oracle import RunningNormalizerOracle
    g = load_golden("normalizer.npz")
    nz = RunningNormalizerOracle(int(g["D"][0]))
    for i in range(len(g["sizes"])):
        nz.update(g[f"x{i}"])
        assert np.array_equal(nz.mean, g[f"mean{i}"]) and np.array_equal(nz.var, g[f"var{i}"]) and nz.count == g[f"count{i}"][0], i
        z = nz.normalize(g["probe"])
        assert np.array_equal(z, g[f"norm64_{i}"]) and np.array_equal(z.astype(np.float32), g[f"norm32_{i}"]), i


def test_normalizer_oracle_after_load_matches_reference_bitwise():
    """The float32 regime the reference enters with RunningNormalizer.load (src/utils.py:108-117): statistics, float32
    normalised probes and float32 merges, against tests/golden/normalizer_loaded.npz (make_golden_norm_load.py)."""
    import yaml
    from oracle.normalizer_oracle import RunningNormalizerOracle
    g = load_golden("normalizer_loaded.npz")
    D = int(g["D"][0])
    pre = RunningNormalizerOracle(D)
    for i in range(int(g["n_pre"][0])):
        pre.update(g[f"pre_x{i}"])
    d = yaml.safe_load(str(g["yaml_text"]))
    assert d["mean"] == pre.mean.tolist() and d["var"] == pre.var.tolist() and d["count"] == float(pre.count)   # what save() wrote
    nz = RunningNormalizerOracle(D)
    nz.load_state(d["mean"], d["var"], d["count"], d["clip_range"])
    assert np.array_equal(nz.mean, g["load_mean"]) and np.array_equal(nz.var, g["load_var"]) and nz.count == g["load_count"][0]
    z = nz.normalize(g["probe"])
    assert z.dtype == np.float32 and np.array_equal(z, g["load_norm"])
    for i in range(len(g["sizes"])):
        nz.update(g[f"x{i}"])
        assert nz.mean.dtype == np.float32 and nz.var.dtype == np.float32
        assert np.array_equal(nz.mean, g[f"mean{i}"]) and np.array_equal(nz.var, g[f"var{i}"]) and nz.count == g[f"count{i}"][0], i
        assert np.array_equal(nz.normalize(g["probe"]), g[f"norm{i}"]), i


def test_normalizer_oracle_with_float64_rows_matches_reference_bitwise():
    """float64 rows — what the reference's trainer feeds the observation normaliser (the vector env allocates observation
    batches with the float64 dtype TimeFeatureWrapper declares, src/utils.py:156) — against tests/golden/normalizer_f64.npz
    (make_golden_norm_f64.py): created normaliser; loaded normaliser (float32 statistics turn float64 with the first update);
    float32 rows then float64 rows on a loaded one."""
    import yaml
    from oracle.normalizer_oracle import RunningNormalizerOracle
    g = load_golden("normalizer_f64.npz")
    D = int(g["D"][0])
    probe = g["probe"]
    assert probe.dtype == np.float64
    nz = RunningNormalizerOracle(D)
    for i in range(len(g["a_sizes"])):
        nz.update(g[f"a_x{i}"])
        assert nz.mean.dtype == np.float64
        assert np.array_equal(nz.mean, g[f"a_mean{i}"]) and np.array_equal(nz.var, g[f"a_var{i}"]) and nz.count == g[f"a_count{i}"][0], i
        assert np.array_equal(nz.normalize(probe), g[f"a_norm{i}"]), i
    d = yaml.safe_load(str(g["yaml_text"]))
    ld = RunningNormalizerOracle(D)
    ld.load_state(d["mean"], d["var"], d["count"], d["clip_range"])
    assert np.array_equal(ld.mean, g["load_mean"]) and np.array_equal(ld.var, g["load_var"])
    z = ld.normalize(probe)
    assert z.dtype == np.float64 and np.array_equal(z, g["b_load_norm"])
    for i in range(len(g["b_sizes"])):
        ld.update(g[f"b_x{i}"])
        assert str(ld.mean.dtype) == str(g[f"b_mean{i}_dtype"]) == "float64"
        assert np.array_equal(ld.mean, g[f"b_mean{i}"]) and np.array_equal(ld.var, g[f"b_var{i}"]) and ld.count == g[f"b_count{i}"][0], i
        assert np.array_equal(ld.normalize(probe), g[f"b_norm{i}"]), i
    ld2 = RunningNormalizerOracle(D)
    ld2.load_state(d["mean"], d["var"], d["count"], d["clip_range"])
    ld2.update(g["c_x32"])
    assert ld2.mean.dtype == np.float32 and np.array_equal(ld2.mean, g["c_mean0"]) and np.array_equal(ld2.var, g["c_var0"])
    assert np.array_equal(ld2.normalize(probe.astype(np.float32)), g["c_norm0_f32rows"]) and np.array_equal(ld2.normalize(probe), g["c_norm0_f64rows"])
    ld2.update(g["c_x64"])
    assert ld2.mean.dtype == np.float64 and np.array_equal(ld2.mean, g["c_mean1"]) and np.array_equal(ld2.var, g["c_var1"]) and ld2.count == g["c_count1"][0]
    assert np.array_equal(ld2.normalize(probe), g["c_norm1"])


# ---------------------------------------------------------------- G10: PER buffer + weighted critic losses
@pytest.mark.parametrize("tag", ["ddpg", "td3", "sac", "tqc"])
def test_per_oracle_matches_reference(tag):
    from oracle.replay_oracle import PERBufferOracle, per_update
    g = load_golden(f"per_{tag}.npz")
    kind = str(g["kind"][0])
    S, A, B, N = (int(x) for x in g["dims"])
    cfg = hparams_from_golden(g)
    torch.set_num_threads(1)
    orc = OracleAgent(kind, S, A, cfg, nenvs=1, gradient_step=2)
    nets = {"actor": orc.actor, **{f"critic_{i}": c for i, c in enumerate(orc.critics)}}
    for n, net in nets.items():
        orc.set_flat_params(net, g[f"init_{n}"])
    orc.hard_update()
    buf = PERBufferOracle(cfg.max_len, cfg.alpha)
    for i in range(N):
        buf.push(g["rows_s"][i], g["rows_a"][i], float(g["rows_r"][i, 0]), g["rows_ns"][i], bool(g["rows_d"][i, 0]))
    np.random.seed(4242)
    beta = dict(beta=cfg.beta, beta0=cfg.beta)
    stoch = kind in ("SAC", "TQC")      # their goldens hold 4 steps, the recorded rsample eps, BN statistics and log_alpha
    for i, step in enumerate((1, 2, 3, 4) if stoch else (1, 2, 3)):
        eps = dict(eps_next=torch.from_numpy(g[f"step{i}_eps_next"]), eps_cur=torch.from_numpy(g[f"step{i}_eps_cur"])) if stoch else {}
        info, td, idx, w = per_update(orc, buf, step, beta, **eps)
        assert np.array_equal(idx, g[f"step{i}_indices"])
        assert np.array_equal(td, g[f"step{i}_td"])
        assert np.array_equal(np.array(buf.priorities, np.float64), g[f"step{i}_priorities"])
        want = g[f"step{i}_tuple"]
        got = np.array([float(np.asarray(x)) for x in info])
        td_pos = {6: 2, 8: 3, 9: 3}[len(want)]
        if stoch:
            assert close(orc.log_alpha.detach().numpy(), g[f"step{i}_log_alpha"])
            bns = [m for m in orc.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
            assert close(np.concatenate([m.running_mean.numpy() for m in bns]), g[f"step{i}_bn_mean"])
        got[td_pos] = float(np.mean(td))        # on this path the reference returns the per-sample array there
        assert close(got, want), (i, got, want)
        for n in nets:
            k = f"step{i}_gradpre_{n}"
            if k in g.files:
                pre = orc.last["actor_grads_pre"] if n == "actor" else orc.last["critic_grads_pre"][int(n[-1])]
                assert close(pre, g[k]), (i, n)
