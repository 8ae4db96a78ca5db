"""GPU (-m gpu): the data-parallel engine path end to end — two ranks sharing ONE GPU over gloo
(RCCL needs one GPU per rank; the exchange code is backend-agnostic) vs a single agent fed the
concatenated batch.  SURVEY.md §8e invariant: G ranks x B rows with summed gradients scaled by 1/G
== 1 rank x G*B rows (DDPG / TD3, up to fp32 summation order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu
S, A, B, H, L = 10, 3, 32, 32, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg(kind, batch):
    from oracle.agent_oracle import make_config
    return make_config(kind, hidden_dim=H, layer_count=L, batch_size=batch, max_len=4000, grad_clip=0.5,
                       ac_update_freq=1, policy_noise=0.0)


def _global_batches(world, steps):
    gen = np.random.default_rng(21)
    out = []
    for _ in range(steps):
        n = world * B
        out.append((gen.standard_normal((n, S)).astype(np.float32), gen.uniform(-1, 1, (n, A)).astype(np.float32),
                    -(gen.uniform(size=(n, 1)) > 0.3).astype(np.float32), gen.standard_normal((n, S)).astype(np.float32),
                    (gen.uniform(size=(n, 1)) > 0.9).astype(np.float32)))
    return out


def _global_eps(world, steps):
    gen = np.random.default_rng(22)
    return [(gen.standard_normal((world * B, A)).astype(np.float32), gen.standard_normal((world * B, A)).astype(np.float32))
            for _ in range(steps)]


def _init_vectors(n_actor, n_critic, n_critics):
    gen = np.random.default_rng(5)
    act = (0.2 * gen.standard_normal(n_actor)).astype(np.float32)
    return act, [(0.2 * gen.standard_normal(n_critic)).astype(np.float32) for _ in range(n_critics)]


def _init_params(agent):
    act, crit = _init_vectors(agent.actor.numel(), agent.critics[0].numel(), len(agent.critics))
    if agent._sac:   # BatchNorm affine parameters near (1, 0), a log_std head that keeps std moderate
        lay, off = agent.actor._param_layout(), 0
        for key, shape in lay:
            n = int(np.prod(shape))
            if key.startswith("base_net") and len(shape) == 1 and int(key.split(".")[1]) % 3 == 1:
                act[off:off + n] = (1.0 if key.endswith("weight") else 0.0) + 0.1 * act[off:off + n]
            off += n
    agent.actor.set_flat(act)
    for c, v in zip(agent.critics, crit):
        c.set_flat(v)
    agent.update_target_network()


def _worker(rank, world, port, kind, out_dir, sync_bn=False, exchange="auto", tag="rank", sep_norm=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if sep_norm:
        os.environ["GCRL_XCHG_SEPARATE_NORM"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import gcrl_amd
    from gcrl_amd.src.dp import DataParallelUpdater
    cls = dict(DDPG=gcrl_amd.DDPG, TD3=gcrl_amd.TD3Agent, SAC=gcrl_amd.SACAgent, TQC=gcrl_amd.TQCAgent)[kind]
    ag = cls(S, A, _cfg(kind, B), None, nenvs=1, gradient_step=2, rng="engine", seed=100 + rank)
    if rank == 0:
        _init_params(ag)            # rank 1 keeps its own random init until the broadcast
    dp = DataParallelUpdater(ag, sync_bn=sync_bn, exchange=exchange)    # broadcasts rank 0's parameters
    assert dp.exchange == {"auto": "engine-ipc", "ipc": "engine-ipc", "python": "python"}[exchange], dp.exchange
    assert dp.sync_bn == (sync_bn and kind in ("SAC", "TQC"))
    if dp.sync_bn:
        assert dp.sync_bn_exchange == ("engine-ipc" if dp.exchange == "engine-ipc" else "python"), dp.sync_bn_exchange
    tuples = []
    for step, (full, eps) in enumerate(zip(_global_batches(world, 3), _global_eps(world, 3)), start=1):
        mine = tuple(torch.from_numpy(x[rank * B:(rank + 1) * B]).cuda() for x in full)
        kw = {}
        if kind in ("SAC", "TQC"):
            kw = dict(eps_next=torch.from_numpy(eps[0][rank * B:(rank + 1) * B]), eps_cur=torch.from_numpy(eps[1][rank * B:(rank + 1) * B]))
        t = [float(x) for x in dp.update(step, batch=mine, **kw)]
        tuples.append(t + [0.0] * (9 - len(t)))
    torch.cuda.synchronize()
    extra = {}
    if kind in ("SAC", "TQC"):
        extra = dict(bn_mean=ag.actor._get("bn_running_mean"), bn_var=ag.actor._get("bn_running_var"),
                     log_alpha=ag.log_alpha.detach().numpy())
    np.savez(os.path.join(out_dir, f"{tag}{rank}.npz"), actor=ag.actor.flat(), critic=ag.critics[0].flat(),
             critic_last=ag.critics[-1].flat(), target=ag.target_critics[0].flat(), tuples=np.array(tuples), **extra)
    dist.barrier()
    dist.destroy_process_group()


def _track(got, ref, rtol_rest, atol_rest, what=""):
    """A DP run against its single-process / oracle twin: the FIRST step — same parameters on both sides, only the summation
    order of the exchanged gradients and of the clip norm differs — is held to the north star (2e-5 covers the two sums);
    later steps compare trajectories in which Adam has turned ~0 gradients' rounding noise into +-lr parameter moves
    (SURVEY.md hard part 3; the reference itself moves 9.8e-4 between thread counts), hence the looser tracking bound."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert np.allclose(got[0], ref[0], rtol=2e-5, atol=2e-6), (what, "step 1", np.abs(got[0] - ref[0]).max())
    assert np.allclose(got, ref, rtol=rtol_rest, atol=atol_rest), (what, np.abs(got - ref).max())


@pytest.mark.parametrize("exchange", ["ipc", "python"])
@pytest.mark.parametrize("kind", ["DDPG", "TD3"])
def test_two_ranks_equal_one_big_batch(gcrl, tmp_path, kind, exchange):
    """exchange: the engine's own peer-to-peer kernel over IPC-mapped gradient arenas (csrc/xchg_ipc.hip; two processes sharing
    this box's one GPU) | torch.distributed (gloo) calls between the engine's segments."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), False, exchange), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("actor", "critic", "target"):
        assert np.array_equal(r0[k], r1[k]), k                 # replicas stay bitwise identical
    cls = dict(DDPG=gcrl.DDPG, TD3=gcrl.TD3Agent)[kind]
    big = cls(S, A, _cfg(kind, world * B), None, nenvs=1, gradient_step=4, rng="engine", seed=1)
    _init_params(big)
    for step, full in enumerate(_global_batches(world, 3), start=1):
        big.update(step, batch=tuple(torch.from_numpy(x).cuda() for x in full))
    # same math up to fp32 summation order; Adam can amplify ~0 gradients to +-lr
    for k, v in (("actor", big.actor), ("critic", big.critics[0]), ("target", big.target_critics[0])):
        err = np.abs(r0[k].astype(np.float64) - v.flat())
        assert float(np.mean(err > 2e-5)) < 0.02 and float(err.max()) < 3 * 2.2e-3, (k, float(err.max()))


def test_three_ranks_through_the_ipc_exchange_equal_one_big_batch(gcrl, tmp_path):
    """A world that is not a power of two: chunk c of the gradient block belongs to rank c mod 3, the owner adds (g0 + g1) + g2.
    Three processes on this box's one GPU; DDPG, three injected-batch steps; replicas bitwise identical, parameters those of ONE
    agent fed the 96-row batch (fp32 summation order apart)."""
    world, kind = 3, "DDPG"
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), False, "ipc"), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    for k in ("actor", "critic", "target"):
        assert np.array_equal(r[0][k], r[1][k]) and np.array_equal(r[0][k], r[2][k]), k
    big = gcrl.DDPG(S, A, _cfg(kind, world * B), None, nenvs=1, gradient_step=4, rng="engine", seed=1)
    _init_params(big)
    for step, full in enumerate(_global_batches(world, 3), start=1):
        big.update(step, batch=tuple(torch.from_numpy(x).cuda() for x in full))
    for k, v in (("actor", big.actor), ("critic", big.critics[0]), ("target", big.target_critics[0])):
        err = np.abs(r[0][k].astype(np.float64) - v.flat())
        assert float(np.mean(err > 2e-5)) < 0.02 and float(err.max()) < 3 * 2.2e-3, (k, float(err.max()))


def _worker_cycle(rank, world, port, out_dir, kind="DDPG", exchange="auto", tag="cycle", sep_norm=False, sync_bn=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if sep_norm:
        os.environ["GCRL_XCHG_SEPARATE_NORM"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import gcrl_amd
    from gcrl_amd.src.dp import DataParallelUpdater
    from oracle import her_oracle
    cls = dict(DDPG=gcrl_amd.DDPG, TD3=gcrl_amd.TD3Agent, SAC=gcrl_amd.SACAgent)[kind]
    ag = cls(S, A, _cfg(kind, B), None, nenvs=1, gradient_step=45, rng="engine", seed=7)   # same seed on both ranks
    gen = np.random.default_rng(3)
    for _ in range(3):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(0, *st)
    if rank == 0:
        _init_params(ag)
    dp = DataParallelUpdater(ag, exchange=exchange, sync_bn=sync_bn)
    if sync_bn:      # the partials travel with the gradients' exchange: in the engine (graphs stay on) or through the callback
        assert dp.sync_bn and dp.sync_bn_exchange == ("engine-ipc" if exchange == "ipc" else "python"), dp.sync_bn_exchange
    out = [[float(x) for x in t] for t in dp.update_many(1, 45)]      # crosses the Polyak step 40
    out += [[float(x) for x in t] for t in dp.update_many(46, 5)]
    torch.cuda.synchronize()
    tgt = ag.target_actor.flat() if hasattr(ag, "target_actor") else ag.target_critics[0].flat()
    np.savez(os.path.join(out_dir, f"{tag}{rank}.npz"), actor=ag.actor.flat(), critic=ag.critics[0].flat(),
             tactor=tgt, tuples=np.array([t + [0.0] * (9 - len(t)) for t in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["DDPG", "TD3", "SAC"])
def test_ipc_exchange_is_bitwise_the_gloo_exchange(gcrl, tmp_path, kind):
    """the engine's two-shot peer-to-peer exchange over IPC-mapped gradient arenas (csrc/xchg_ipc.hip),
    two processes on this box's one GPU, against the torch.distributed (gloo) all-reduce between the engine's segments —
    BITWISE: tuples, parameters and targets of a 50-step trainer cycle (update_many: pipelined DDPG segments with one exchange
    per overlapped step, Polyak step, multi-step graphs on the IPC side) and of three injected-batch update() steps.  With two
    ranks the rank-order sum a + b is the all-reduce's; the clip norm is taken from a sum-of-squares launch on both sides here
    (GCRL_XCHG_SEPARATE_NORM=1) because the exchange kernel's own partials sum the squares in another order (1 ulp on the
    norm: covered by the tracking tests below and above, which run with the partials)."""
    world = 2
    for exchange, tag, sep in (("ipc", "i_", True), ("python", "p_", False)):
        mp.spawn(_worker_cycle, args=(world, _free_port(), str(tmp_path), kind, exchange, tag + "cycle", sep), nprocs=world, join=True)
        mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), False, exchange, tag + "rank", sep), nprocs=world, join=True)
    for base, keys in (("cycle", ("actor", "critic", "tactor", "tuples")), ("rank", ("actor", "critic", "critic_last", "target", "tuples"))):
        for r in range(world):
            i, p = np.load(tmp_path / f"i_{base}{r}.npz"), np.load(tmp_path / f"p_{base}{r}.npz")
            for k in keys:
                assert np.array_equal(i[k], p[k]), (kind, base, r, k, float(np.abs(i[k].astype(np.float64) - p[k]).max()))


def test_sync_batchnorm_through_the_engine_exchange_is_bitwise_the_callback_path(gcrl, tmp_path):
    """with `sync_bn=True` the BatchNorm row-block partials ([world x nrb x 2H] floats per layer and pass:
    reference semantics src/model.py:107 on the concatenated batch) used to travel through a host callback between the launches,
    with hipGraphs off.  Now they go through the same peer-to-peer kernel as the gradients (an exchange handle over the partials'
    own arena, gcrl_agent_bn_xchg_create / gcrl_agent_dp_sync_bn_xchg): an exchange of zero-padded slots in rank order is an
    all-gather, so the merged statistics — and with them a 50-step SAC trainer cycle replayed from multi-step graphs and three
    injected-batch steps — must be BITWISE what the callback path (torch.distributed all-reduce, plain launches) gives."""
    world, kind = 2, "SAC"
    for exchange, tag, sep in (("ipc", "i_", True), ("python", "p_", False)):
        mp.spawn(_worker_cycle, args=(world, _free_port(), str(tmp_path), kind, exchange, tag + "cycle", sep, True), nprocs=world, join=True)
        mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), True, exchange, tag + "rank", sep), nprocs=world, join=True)
    for base, keys in (("cycle", ("actor", "critic", "tactor", "tuples")), ("rank", ("actor", "critic", "critic_last", "target", "tuples", "bn_mean", "bn_var"))):
        for r in range(world):
            i, p = np.load(tmp_path / f"i_{base}{r}.npz"), np.load(tmp_path / f"p_{base}{r}.npz")
            for k in keys:
                assert np.array_equal(i[k], p[k]), (base, r, k, float(np.abs(i[k].astype(np.float64) - p[k]).max()))


@pytest.mark.parametrize("exchange", ["ipc", "python"])
@pytest.mark.parametrize("kind", ["DDPG", "TD3"])
def test_dp_cycle_schedule_tracks_single_process(gcrl, tmp_path, kind, exchange):
    """Engine-scheduled DP cycle (pipelined DDPG segments with ONE exchange per step, ordinary
    phases around the Polyak step): two ranks holding IDENTICAL rings and RNG seeds draw identical
    batches, so the averaged gradients equal each rank's own and the run must track a single-process
    update_many (same math; the clip norm is summed by a different kernel, hence a tolerance)."""
    from oracle import her_oracle
    world = 2
    mp.spawn(_worker_cycle, args=(world, _free_port(), str(tmp_path), kind, exchange), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "cycle0.npz"), np.load(tmp_path / "cycle1.npz")
    for k in ("actor", "critic", "tactor", "tuples"):
        assert np.array_equal(r0[k], r1[k]), k
    cls = dict(DDPG=gcrl.DDPG, TD3=gcrl.TD3Agent)[kind]
    ag = cls(S, A, _cfg(kind, B), None, nenvs=1, gradient_step=45, rng="engine", seed=7)
    gen = np.random.default_rng(3)
    for _ in range(3):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(0, *st)
    _init_params(ag)
    ref = [[float(x) for x in t] for t in ag.update_many(1, 45)] + [[float(x) for x in t] for t in ag.update_many(46, 5)]
    ref = np.array([t + [0.0] * (9 - len(t)) for t in ref])
    _track(r0["tuples"], ref, 2e-4, 2e-5, kind)
    for k, v in (("actor", ag.actor), ("critic", ag.critics[0]), ("tactor", ag.target_actor)):
        assert float(np.max(np.abs(r0[k] - v.flat()))) < 5e-4, k


@pytest.mark.parametrize("kind", ["SAC", "TQC"])
def test_two_ranks_local_batchnorm_matches_the_dp_oracle(gcrl, tmp_path, kind):
    """cfg 5's agent under data parallelism.  The BatchNorm actor keeps LOCAL batch statistics per rank
    (DESIGN.md §6), so G x B is not 1 x G*B; the specification is oracle/dp_oracle.py: G replicas of the
    reference's step, each on its rank's rows with its own BatchNorm statistics, gradients averaged after
    every backward pass.  Checked: replicas' parameters stay bitwise identical across ranks, every rank's
    tuple, the parameters after three steps, log_alpha and each rank's own running statistics."""
    from oracle.agent_oracle import OracleAgent
    from oracle.dp_oracle import DPOracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    for k in ("actor", "critic", "critic_last", "target", "log_alpha"):
        assert np.array_equal(r[0][k], r[1][k]), k
    torch.set_num_threads(1)
    cfg = _cfg(kind, B)
    reps = [OracleAgent(kind, S, A, cfg, nenvs=1, gradient_step=2) for _ in range(world)]
    probe = dict(SAC=gcrl.SACAgent, TQC=gcrl.TQCAgent)[kind](S, A, cfg, None, nenvs=1, gradient_step=2, rng="engine", seed=1)
    _init_params(probe)
    for o in reps:
        o.set_flat_params(o.actor, probe.actor.flat())
        for oc, pc in zip(o.critics, probe.critics):
            o.set_flat_params(oc, pc.flat())
        o.hard_update()
    dpo = DPOracle(reps)
    want = []
    for step, (full, eps) in enumerate(zip(_global_batches(world, 3), _global_eps(world, 3)), start=1):
        bs = [tuple(torch.from_numpy(x[i * B:(i + 1) * B]) for x in full) for i in range(world)]
        kw = [dict(eps_next=torch.from_numpy(eps[0][i * B:(i + 1) * B]), eps_cur=torch.from_numpy(eps[1][i * B:(i + 1) * B]))
              for i in range(world)]
        outs = dpo.update(step, bs, kw)
        want.append([[float(np.asarray(x)) for x in o] for o in outs])
    for i in range(world):
        w = np.array([t + [0.0] * (9 - len(t)) for t in (want[s][i] for s in range(3))])
        _track(r[i]["tuples"], w, 2e-4, 2e-5, (kind, i))
        bns = [m for m in reps[i].actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
        # a Linear bias in front of a BatchNorm has an analytically zero gradient: Adam moves it by +-lr per step on
        # rounding noise, and the batch mean moves with it (0.1 * lr per step into the running mean)
        assert np.allclose(r[i]["bn_mean"], np.concatenate([m.running_mean.numpy() for m in bns]), rtol=1e-3, atol=1e-3)
        assert np.allclose(r[i]["bn_var"], np.concatenate([m.running_var.numpy() for m in bns]), rtol=1e-3, atol=1e-4)
    assert not np.array_equal(r[0]["bn_mean"], r[1]["bn_mean"])      # local statistics: the ranks saw different rows
    o = reps[0]
    # Linear biases in front of a BatchNorm: zero gradient, moved by +-lr per step on rounding noise -> left out
    keep, off = np.ones(probe.actor.numel(), bool), 0
    for key, shape in probe.actor._param_layout():
        n = int(np.prod(shape))
        if key.startswith("base_net") and key.endswith("bias") and int(key.split(".")[1]) % 3 == 0:
            keep[off:off + n] = False
        off += n
    for k, v in (("actor", o.flat_params(o.actor)), ("critic", o.flat_params(o.critics[0])),
                 ("critic_last", o.flat_params(o.critics[-1])), ("target", o.flat_params(o.target_critics[0]))):
        err = np.abs(r[0][k].astype(np.float64) - v)
        if k == "actor":
            err = err[keep]
        assert float(np.mean(err > 2e-5)) < 0.02 and float(err.max()) < 3 * 2.2e-3, (k, float(err.max()))
    assert abs(float(r[0]["log_alpha"][0]) - float(o.log_alpha.detach())) < 1e-5


@pytest.mark.parametrize("kind", ["SAC", "TQC"])
def test_sync_batchnorm_two_ranks_equal_one_big_batch(gcrl, tmp_path, kind):
    """cfg 5's agent with SyncBN (gcrl_agent_dp_sync_bn): BatchNorm statistics of the concatenated batch, forward and
    backward, so the SURVEY §8e invariant holds for the BatchNorm actors as well — 2 ranks x B rows == 1 rank x 2B rows:
    replicas bitwise identical, parameters / running statistics / log_alpha those of ONE agent fed the concatenated batch,
    gradient-norm entries of the tuples equal to its, mean-type entries equal to it on average over the ranks."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), True), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    for k in ("actor", "critic", "critic_last", "target", "log_alpha", "bn_mean", "bn_var"):
        assert np.array_equal(r[0][k], r[1][k]), k          # global statistics: even the running ones are identical now
    cls = dict(SAC=gcrl.SACAgent, TQC=gcrl.TQCAgent)[kind]
    big = cls(S, A, _cfg(kind, world * B), None, nenvs=1, gradient_step=2, rng="engine", seed=1)
    _init_params(big)
    ref = []
    for step, (full, eps) in enumerate(zip(_global_batches(world, 3), _global_eps(world, 3)), start=1):
        t = big.update(step, batch=tuple(torch.from_numpy(x).cuda() for x in full), eps_next=torch.from_numpy(eps[0]), eps_cur=torch.from_numpy(eps[1]))
        ref.append([float(x) for x in t])
    ref = np.array(ref)
    got = (r[0]["tuples"] + r[1]["tuples"]) / 2.0           # losses, td, q, alpha loss: means over a rank's rows
    norms = [5, 6, 7]                                        # gradient norms: of the summed-and-scaled gradients, the same on every rank
    assert np.array_equal(r[0]["tuples"][:, norms], r[1]["tuples"][:, norms])
    got[:, norms] = r[0]["tuples"][:, norms]
    assert np.allclose(got[0], ref[0], rtol=2e-5, atol=2e-6), ("step 1", np.abs(got[0] - ref[0]).max(), got[0], ref[0])
    assert np.allclose(got, ref, rtol=2e-4, atol=2e-5), np.abs(got - ref).max()
    # (a Linear bias in front of a BatchNorm has an analytically zero gradient: Adam moves it by +-lr per step on rounding
    # noise, and the batch mean moves with it — 0.1 * lr per step into the running mean)
    assert np.allclose(r[0]["bn_mean"], big.actor._get("bn_running_mean"), rtol=1e-3, atol=1e-3)
    assert np.allclose(r[0]["bn_var"], big.actor._get("bn_running_var"), rtol=1e-3, atol=1e-4)
    assert abs(float(r[0]["log_alpha"][0]) - float(big.log_alpha.detach())) < 1e-5
    keep, off = np.ones(big.actor.numel(), bool), 0       # Linear biases in front of a BatchNorm: zero gradient, +-lr on rounding noise
    for key, shape in big.actor._param_layout():
        n = int(np.prod(shape))
        if key.startswith("base_net") and key.endswith("bias") and int(key.split(".")[1]) % 3 == 0:
            keep[off:off + n] = False
        off += n
    for k, v in (("actor", big.actor.flat()), ("critic", big.critics[0].flat()), ("critic_last", big.critics[-1].flat()),
                 ("target", big.target_critics[0].flat())):
        err = np.abs(r[0][k].astype(np.float64) - v)
        if k == "actor":
            err = err[keep]
        assert float(np.mean(err > 2e-5)) < 0.02 and float(err.max()) < 3 * 2.2e-3, (k, float(err.max()))


def _worker_rccl(rank, world, port, out_dir, kind, exchange="rccl"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import gcrl_amd
    from gcrl_amd.src.dp import DataParallelUpdater
    from oracle import her_oracle
    cls = dict(DDPG=gcrl_amd.DDPG, SAC=gcrl_amd.SACAgent)[kind]
    ag = cls(S, A, _cfg(kind, B), None, nenvs=1, gradient_step=45, rng="engine", seed=7)
    gen = np.random.default_rng(3)
    for _ in range(3):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(0, *st)
    _init_params(ag)
    dp = DataParallelUpdater(ag, exchange=exchange, require_native=True)
    assert dp.exchange == {"rccl": "engine-rccl", "ipc": "engine-ipc"}[exchange], dp.exchange
    out = [[float(x) for x in t] for t in dp.update_many(1, 45)]      # one native call: segments + all-reduces
    out += [[float(x) for x in dp.update(46)]]                        # the per-phase entry, collectives native too
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl.npz"), actor=ag.actor.flat(), critic=ag.critics[0].flat(),
             tuples=np.array([t + [0.0] * (9 - len(t)) for t in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "ipc"])
@pytest.mark.parametrize("kind", ["DDPG", "SAC"])
def test_in_engine_exchange_world_size_one(gcrl, tmp_path, kind, exchange):
    """The library-owned RCCL communicator (csrc/dp_rccl.cc) with the one-call trainer cycle (gcrl_agent_dp_run_all), and the
    engine's own exchange kernel (csrc/xchg_ipc.hip, the default) under an RCCL process group, on the one GPU this box has: a
    single-rank all-reduce is the identity, so the run must track a single-process update_many (the clip norm is summed by a
    different kernel: tolerance)."""
    from oracle import her_oracle
    mp.spawn(_worker_rccl, args=(1, _free_port(), str(tmp_path), kind, exchange), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    cls = dict(DDPG=gcrl.DDPG, SAC=gcrl.SACAgent)[kind]
    ag = cls(S, A, _cfg(kind, B), None, nenvs=1, gradient_step=45, rng="engine", seed=7)
    gen = np.random.default_rng(3)
    for _ in range(3):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(0, *st)
    _init_params(ag)
    ref = [[float(x) for x in t] for t in ag.update_many(1, 45)] + [[float(x) for x in ag.update(46)]]
    ref = np.array([t + [0.0] * (9 - len(t)) for t in ref])
    if kind == "DDPG":
        _track(r["tuples"], ref, 2e-4, 2e-5, kind)
        assert float(np.max(np.abs(r["actor"] - ag.actor.flat()))) < 5e-4
    else:   # SAC draws its exploration noise from the device counter hash: same counters, same draws
        _track(r["tuples"][:5], ref[:5], 5e-4, 5e-5, kind)


# ------------------------------------------------------------------ the exchange kernel by itself (csrc/xchg_ipc.hip)
_XSEGS = [(0, 5000), (5056, 1025), (6144, 3), (6208, 1), (6272, 4096)]       # (offset, floats): ragged tails, one-float segment, exact chunks
_XARENA = 10368


def _xchg_expected(world):
    gens = [np.random.default_rng(900 + r) for r in range(world)]
    arenas = [g.standard_normal(_XARENA).astype(np.float32) for g in gens]
    want = arenas[0].copy()
    for off, n in _XSEGS:
        acc = arenas[0][off:off + n].copy()
        for r in range(1, world):
            acc = (acc + arenas[r][off:off + n]).astype(np.float32)      # rank order
        want[off:off + n] = acc
    return arenas, want


def _worker_xchg_raw(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import ctypes as C
    import gcrl_amd
    lib, ffi = gcrl_amd._ffi.lib, gcrl_amd._ffi
    arenas, _ = _xchg_expected(world)
    buf = torch.from_numpy(arenas[rank]).cuda()
    off = (C.c_int64 * len(_XSEGS))(*[o for o, _ in _XSEGS])
    num = (C.c_int64 * len(_XSEGS))(*[n for _, n in _XSEGS])
    x = ffi.check_ptr(lib.gcrl_xchg_create(buf.data_ptr(), buf.numel(), off, num, len(_XSEGS), rank, world, 0), "gcrl_xchg_create")
    HB = ffi.XCHG_HANDLE_BYTES
    rec = (C.c_uint8 * HB)()
    ffi.check(lib.gcrl_xchg_handles(x, rec, HB))
    recs = [None] * world
    dist.all_gather_object(recs, bytes(rec))
    ffi.check(lib.gcrl_xchg_connect(x, b"".join(recs), HB * world))
    dist.barrier()
    st = ffi.stream_handle()
    # exchange 1: segments 0..2; exchange 2: segments 3..4 (a second launch on the same counters); then both again inside a hipGraph
    ffi.check(lib.gcrl_xchg_allreduce(x, 0, 3, st))
    ffi.check(lib.gcrl_xchg_allreduce(x, 3, 2, st))
    torch.cuda.synchronize()

    def result():       # the reduced values (the fine-grained receive buffer; in a world of one the arena itself)
        out = np.empty(_XARENA, np.float32)
        ffi.check(lib.gcrl_xchg_read(x, 0, _XARENA, out.ctypes.data))
        return out

    first = result()
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), arenas[rank].view(np.uint32))      # the arena is only ever READ by the exchange
    parts = []
    for sgi in range(len(_XSEGS)):
        pb = np.zeros(8, np.float32)
        k = ffi.check(lib.gcrl_xchg_get_partials(x, sgi, pb.ctypes.data, 8))
        parts.append(pb[:k].copy())
    g = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        buf.copy_(torch.from_numpy(arenas[rank]))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s2):
            ffi.check(lib.gcrl_xchg_allreduce(x, 0, 5, int(s2.cuda_stream)))
        dist.barrier()
        for _ in range(3):                 # replays: the exchange number lives on the device
            buf.copy_(torch.from_numpy(arenas[rank]))
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            dist.barrier()
    np.savez(os.path.join(out_dir, f"x{rank}.npz"), first=first, replay=result(), parts=np.concatenate(parts))
    dist.barrier()
    lib.gcrl_xchg_destroy(x)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_exchange_kernel_sums_in_rank_order_on_every_rank(gcrl, tmp_path, world):
    """gcrl_xchg_* over a raw arena: every rank ends with the rank-order sum of the segments' floats (ragged segment tails, a
    one-float segment) in its receive buffer, its arena untouched, two exchanges back to back on the same counters, and three
    replays of the exchange captured in a hipGraph (the exchange number is device state)."""
    mp.spawn(_worker_xchg_raw, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    arenas, want = _xchg_expected(world)
    for r in range(world):
        got = np.load(tmp_path / f"x{r}.npz")
        mask = np.zeros(_XARENA, bool)
        for off, n in _XSEGS:
            mask[off:off + n] = True
        for key in ("first", "replay"):                    # inside the segments: the rank-order sum, on every rank
            assert np.array_equal(got[key][mask].view(np.uint32), want[mask].view(np.uint32)), (r, key)
        # the clip norm's partials: one per 1024-float chunk of every segment, of the REDUCED values, the same on every rank
        wantp = []
        for off, n in _XSEGS:
            for c0 in range(0, n, 1024):
                v = want[off + c0:off + min(n, c0 + 1024)].astype(np.float64)
                wantp.append(float(np.dot(v, v)))
        assert got["parts"].shape == (len(wantp),) and np.allclose(got["parts"], wantp, rtol=2e-6, atol=0), r
        assert np.array_equal(got["parts"], np.load(tmp_path / "x0.npz")["parts"])


def test_sync_bn_reconfiguration_grows_its_buffer(gcrl):
    """gcrl_agent_dp_sync_bn called again with a LARGER world must not keep the buffer sized for the first call (the
    statistics kernels then wrote 2 * world * blocks * H floats past its end).  World 2 -> 4 -> 1 on one agent, a step after
    each, with an exchange function that is the identity (the other ranks' slots stay zero: the statistics are those of a batch
    padded with zero-weight blocks — only finiteness and the absence of a fault are asserted here)."""
    import ctypes as C
    from gcrl_amd import _ffi
    ag = gcrl.SACAgent(S, A, _cfg("SAC", B), None, nenvs=1, gradient_step=2, rng="engine", seed=3)
    _init_params(ag)
    calls = []

    def ident(ptr, n, stream, user):
        calls.append(int(n))
        return 0
    cb = _ffi.EXCHANGE_FN(ident)
    full, eps = _global_batches(1, 3), _global_eps(1, 3)
    for step, world in enumerate((2, 4, 1), start=1):
        _ffi.check(_ffi.lib.gcrl_agent_dp_sync_bn(ag._h, world, 0, None, C.cast(cb, C.c_void_p) if world > 1 else None, None))
        before = len(calls)
        t = ag.update(step, batch=tuple(torch.from_numpy(x).cuda() for x in full[step - 1]),
                      eps_next=torch.from_numpy(eps[step - 1][0]), eps_cur=torch.from_numpy(eps[step - 1][1]))
        assert all(np.isfinite(float(x)) for x in t), (world, [float(x) for x in t])
        assert (len(calls) > before) == (world > 1)
        if world > 1:
            assert max(calls[before:]) >= 2 * world * ((B + 63) // 64) * H      # every rank's slots travel
    torch.cuda.synchronize()
    assert np.all(np.isfinite(ag.actor.flat()))
