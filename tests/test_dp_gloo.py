"""CPU, world_size=2, gloo: the data-parallel host logic of src/dp.py (the RCCL path's twin).

Invariant: G ranks x B rows with all-reduce-averaged gradients == 1 rank x G*B rows,
for the critic loss and for the actor loss through the (identically) stepped critic — computed here
with the oracle's networks, exchanged with dp.allreduce_mean_ / dp.broadcast_ exactly as the GPU
path exchanges its flat gradient blocks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401  (sys.path)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat_grads(net):
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gcrl_amd  # noqa: F401
    from gcrl_amd.src import dp
    from oracle.agent_oracle import OracleAgent, make_config

    torch.set_num_threads(1)
    S, A, B = 10, 3, 32
    cfg = make_config("DDPG", hidden_dim=32, layer_count=2, batch_size=B)
    torch.manual_seed(100 + rank)                      # different initial weights per rank ...
    ag = OracleAgent("DDPG", S, A, cfg)
    for net in [ag.actor] + ag.critics:                # ... until rank 0's are broadcast
        flat = torch.from_numpy(ag.flat_params(net))
        dp.broadcast_(flat, 0)
        ag.set_flat_params(net, flat.numpy())
    ag.hard_update()

    gen = np.random.default_rng(7)                     # same global batch on every rank
    full = [gen.standard_normal((world * B, S)).astype(np.float32), gen.uniform(-1, 1, (world * B, A)).astype(np.float32),
            -(gen.uniform(size=(world * B, 1)) > 0.3).astype(np.float32),
            gen.standard_normal((world * B, S)).astype(np.float32), (gen.uniform(size=(world * B, 1)) > 0.9).astype(np.float32)]
    mine = [torch.from_numpy(x[rank * B:(rank + 1) * B]) for x in full]
    s, a, r, ns, d = mine

    # phase 0: local critic gradient -> averaged over ranks
    with torch.no_grad():
        y = r + cfg.gamma * (1 - d) * ag.target_critics[0](torch.cat([ns, ag.target_actor(ns)], -1))
        y = torch.clamp(y, -1 / (1 - cfg.gamma), 0.0)
    ag.critics[0].zero_grad()
    torch.nn.functional.mse_loss(ag.critics[0](torch.cat([s, a], -1)), y).backward()
    g_c = _flat_grads(ag.critics[0]).clone()
    dp.allreduce_mean_(g_c)
    # phase 1: identical optimiser step everywhere, then the actor gradient through the stepped critic
    off = 0
    for p in ag.critics[0].parameters():
        p.grad.copy_(g_c[off:off + p.numel()].view_as(p)); off += p.numel()
    ag.critic_opts[0].step()
    ag.actor.zero_grad()
    (-ag.critics[0](torch.cat([s, ag.actor(s)], -1)).mean()).backward()
    g_a = _flat_grads(ag.actor).clone()
    dp.allreduce_mean_(g_a)

    shards = [list(dp.shard_env_streams(64, r_, 8)) for r_ in range(8)]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), g_c=g_c.numpy(), g_a=g_a.numpy(),
             critic=ag.flat_params(ag.critics[0]), shard_ok=np.array([sorted(sum(shards, [])) == list(range(64))]),
             seed=np.array([dp.rank_seed(1898, rank)]))
    if rank == 0:   # single-process reference on the concatenated batch
        torch.manual_seed(100)
        ref = OracleAgent("DDPG", S, A, cfg)
        S_, A_, R_, NS_, D_ = (torch.from_numpy(x) for x in full)
        with torch.no_grad():
            y = R_ + cfg.gamma * (1 - D_) * ref.target_critics[0](torch.cat([NS_, ref.target_actor(NS_)], -1))
            y = torch.clamp(y, -1 / (1 - cfg.gamma), 0.0)
        ref.critics[0].zero_grad()
        torch.nn.functional.mse_loss(ref.critics[0](torch.cat([S_, A_], -1)), y).backward()
        rg_c = _flat_grads(ref.critics[0]).clone()
        ref.critic_opts[0].step()
        ref.actor.zero_grad()
        (-ref.critics[0](torch.cat([S_, ref.actor(S_)], -1)).mean()).backward()
        np.savez(os.path.join(out_dir, "ref.npz"), g_c=rg_c.numpy(), g_a=_flat_grads(ref.actor).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_big_batch(tmp_path, gcrl):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1, ref = (np.load(tmp_path / f) for f in ("rank0.npz", "rank1.npz", "ref.npz"))
    assert np.array_equal(r0["g_c"], r1["g_c"]) and np.array_equal(r0["g_a"], r1["g_a"])   # replicas stay identical
    assert np.array_equal(r0["critic"], r1["critic"])
    assert np.allclose(r0["g_c"], ref["g_c"], rtol=1e-5, atol=1e-7)
    assert np.allclose(r0["g_a"], ref["g_a"], rtol=1e-4, atol=1e-7)
    assert bool(r0["shard_ok"][0]) and int(r0["seed"][0]) == 1898 and int(r1["seed"][0]) == 1899


def test_shard_env_streams_partition():
    from gcrl_amd.src.dp import shard_env_streams
    for nenvs, world in [(64, 8), (10, 4), (3, 8), (8, 1)]:
        parts = [list(shard_env_streams(nenvs, r, world)) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(nenvs))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
