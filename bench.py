#!/usr/bin/env python3
"""bench.py — gradient-steps/sec of the HER-sample + critic/actor update path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one agent.update(step): HER batch draw + gather, critic update, actor update when
due, Polyak when due (reference src/agent.py:1378-1404 / src/env.py:384-385).  Steps are issued
as the trainer issues them, `gradient_step` (40) at a time with the buffer untouched in between.
Inputs (the replay ring) are resident in HBM before the timed region.

N > 1: one process per GPU — under torchrun (RANK / WORLD_SIZE in the environment) or, when started as
plain `python bench.py --gpus N`, N rank processes spawned by this script before it touches the GPU.
Each rank owns a local ring and draws its own batch of B rows; gradients are all-reduced over RCCL —
once per overlapped DDPG step (critic and actor blocks are adjacent), twice per step otherwise.
Per-GPU work is fixed -> "scaling": "weak"; value = N x synchronised optimiser steps / s (batch-B-equivalent
steps: one synchronised step consumes N x B rows — `value_semantics` in the line says so).  Two extra
untimed-by-the-headline legs at N > 1: `strong_scaling` (the batch of B rows split over the ranks) and
`cfg5_sac_slide_b512` (BASELINE.json configs[4]).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline         the dominant kernel.  DDPG (row-block path): rowchain_ddpg_kernel, the forward and
                   input-gradient chains of a step's two phases in one launch — algorithmic flops per launch /
                   the kernel's average duration vs the fp32 MFMA peak.  Other agents: the whole step.
  roofline_gather  the HER sample (gather) kernel against 8 TB/s, always: algorithmic bytes of a trainer cycle's MAIN gather
                   launch / its average duration
  roofline_flush   the HER relabel + flush kernel (bytes per episode / duration), and push-side rows/s
  cpu_baseline     the oracle (CPU restatement of the reference) timed on this host, same workload; `cpu_baseline_n1e5`
                   the same with a 1e5-row deque (BASELINE.md's sanity row: random.sample(deque) is O(N))
Kernel durations are THE PROFILER'S: at N = 1 this script first runs itself once as a child under
`rocprofv3 --kernel-trace --stats` (before this process touches the GPU; same workload, fewer steps) and reads the per-kernel
average from the tool's own summary — `profiler.kernel_stats` in the line; `--keep-profile DIR` keeps the CSV.  The HIP-event
pair around the same launches (on the launch stream, separate untimed leg) is reported next to it; it includes ~3-4 us of
dispatch per bracketed launch.  If the tool cannot run, the line says so and the fractions come from the event time.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TF = 157.3

WORKLOADS = {
    # north-star headline: PandaPickAndPlace-v3 DDPG+HER, batch 256 (config_ddpg_pickplace.yaml
    # hyper-parameters, batch overridden to the metric's 256)
    "ddpg_pickplace_b256": dict(kind="DDPG", S=23, A=4, H=256, L=3, B=256, cap=1_000_000, k=8, gamma=0.98,
                                tau=0.05, grad_clip=10.0, freq=1, gstep=40, lr=1e-3),
    # BASELINE.json configs[1]: Reach, B=1024, buffer 1e6, MLP(256,256)
    "ddpg_reach_b1024": dict(kind="DDPG", S=10, A=3, H=256, L=2, B=1024, cap=1_000_000, k=4, gamma=0.98,
                             tau=0.05, grad_clip=10.0, freq=1, gstep=40, lr=1e-3),
    # configs[0]: the reference's CPU-runnable case
    "ddpg_reach_b256": dict(kind="DDPG", S=10, A=3, H=64, L=3, B=256, cap=100_000, k=4, gamma=0.98,
                            tau=0.05, grad_clip=10.0, freq=1, gstep=40, lr=1e-3),
    "td3_pickplace_b2048": dict(kind="TD3", S=23, A=4, H=256, L=3, B=2048, cap=1_000_000, k=8, gamma=0.98,
                                tau=0.005, grad_clip=2.0, freq=2, gstep=40, lr=1e-3, policy_noise=0.2, noise_clamp=0.5),
    "tqc_push_b2048": dict(kind="TQC", S=22, A=3, H=512, L=3, B=2048, cap=1_000_000, k=4, gamma=0.95,
                           tau=0.05, grad_clip=5.0, freq=1, gstep=40, lr=3e-4),
    # BASELINE.json configs[3] as WORDED ("25 quantiles x 2 critics, top-2 truncate"): the distributional variant, which the
    # reference does not contain (its TQC is the 5-scalar-critic ensemble above) — no reference parity, oracle-pinned only
    "tqc_quantile_push_b2048": dict(kind="TQC", S=22, A=3, H=512, L=3, B=2048, cap=1_000_000, k=4, gamma=0.95,
                                    tau=0.05, grad_clip=5.0, freq=1, gstep=40, lr=3e-4, n_quantiles=25, num_critics=2, top_drop=2),
    "sac_slide_b512": dict(kind="SAC", S=22, A=3, H=256, L=3, B=512, cap=1_000_000, k=4, gamma=0.98,
                           tau=0.005, grad_clip=2.0, freq=1, gstep=40, lr=5e-4),
}


def make_cfg(w):
    from gcrl_amd.src.synthetic import agent_config as make_config
    return make_config(w["kind"], hidden_dim=w["H"], layer_count=w["L"], batch_size=w["B"], max_len=w["cap"],
                       k_future=w["k"], gamma=w["gamma"], tau=w["tau"], grad_clip=w["grad_clip"],
                       ac_update_freq=w["freq"], actor_lr=w["lr"], actor_lr_min=w["lr"], critic_lr=w["lr"],
                       critic_lr_min=w["lr"], policy_noise=w.get("policy_noise", 0.0),
                       noise_clamp=w.get("noise_clamp", 0.5))


def episode_pool(w, n, seed):
    from gcrl_amd.src.synthetic import synthetic_episode
    gen = np.random.default_rng(seed)
    return [synthetic_episode(gen, 50, w["S"], w["A"]) for _ in range(n)]


def episode_arrays(steps):
    s, a, ns, r, d, dg, ag = zip(*steps)
    return (np.array(s, np.float32), np.array(a, np.float32), np.array(ns, np.float32),
            np.array(r, np.float32), np.zeros(len(steps), np.float32), np.array(ag, np.float32))


def flops_per_step(w, actor_step=None):
    """fwd = dX = dW = 2*B*P per network pass.  actor_step None: the MEAN over a trainer cycle — TD3 steps its
    actor every `ac_update_freq`-th update only (src/agent.py:281-317), the other agents every update."""
    if actor_step is None and w["kind"] == "TD3":
        f = int(w.get("freq", 2))
        return (flops_per_step(w, True) + (f - 1) * flops_per_step(w, False)) / f
    actor_step = True if actor_step is None else actor_step
    S, A, H, L, B = w["S"], w["A"], w["H"], w["L"], w["B"]
    Pa = S * H + (L - 1) * H * H + H * A
    Pc = (S + A) * H + (L - 1) * H * H + H
    kind = w["kind"]
    if kind == "DDPG":
        return 2 * B * (4 * Pa + 6 * Pc)
    if kind == "TD3":
        return 2 * B * ((4 * Pa + 10 * Pc) if actor_step else (Pa + 8 * Pc))
    if kind == "SAC":
        return 2 * B * (4 * Pa + 12 * Pc)
    if w.get("n_quantiles", 1) > 1:
        Pc += H * (w["n_quantiles"] - 1)
        return 2 * B * (4 * Pa + 7 * w["num_critics"] * Pc)
    return 2 * B * (4 * Pa + 7 * 5 * Pc)


def head_batches(gstep, kind="DDPG"):
    """batches of a call's head gather launch (csrc/agent.hip build(): 4 for the pipelined DDPG step, 3 otherwise; GCRL_HEAD_BATCHES overrides)"""
    return min(int(gstep), max(1, min(8, int(os.environ.get("GCRL_HEAD_BATCHES", "4" if kind == "DDPG" else "3")))))


def chain_flops_per_launch(w):
    """Algorithmic flops of one overlapped row-block launch: critic phase K = target
    actor fwd + target critic fwd + critic fwd + critic dX (hidden layers); actor phase P = actor fwd
    + critic fwd + critic dX down to the action columns + actor dX (hidden layers).  2 flop per MAC."""
    S, A, H, L, B = w["S"], w["A"], w["H"], w["L"], w["B"]
    fa = S * H + (L - 1) * H * H + H * A          # actor forward MACs per row
    fc = (S + A) * H + (L - 1) * H * H + H        # critic forward
    dxc = (L - 1) * H * H + H                     # critic dX, head and hidden layers
    dxa = (L - 1) * H * H + H * A                 # actor dX
    k = fa + fc + fc + dxc
    p = fa + fc + dxc + H * A + dxa
    return 2 * B * (k + p)


def cpu_baseline(w, pool, budget_s):
    """The oracle (CPU port of the reference: deque + random.sample + eager torch) on the same
    workload, bounded to ~budget_s seconds of CPU work."""
    from oracle.agent_oracle import OracleAgent
    cfg = make_cfg(w)
    threads = torch.get_num_threads()
    orc = OracleAgent(w["kind"], w["S"], w["A"], cfg, nenvs=8, gradient_step=w["gstep"], rng=random.Random(1898))
    rows_per_ep = 50 + w["k"] * 49
    n_eps = -(-w["cap"] // rows_per_ep)
    t0 = time.perf_counter()
    for ep in range(n_eps):
        for st in pool[ep % len(pool)]:
            orc.push_her(ep % 8, *st)
    fill_s = time.perf_counter() - t0
    # eager torch on tiny layers is often fastest with few threads; a 256-thread default would
    # handicap the baseline, so probe a few settings and time the best one
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    step, best = 1, (0.0, 1)
    for cand in sorted({1, min(4, avail), min(8, avail), min(16, avail)}):
        torch.set_num_threads(cand)
        orc.update(step); step += 1
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            orc.update(step); step += 1; k += 1
        rate = k / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, cand)
    threads = best[1]
    torch.set_num_threads(threads)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.update(step + n)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 2000:
            break
    return dict(value=n / el, unit="gradient-steps/s", cores=threads, kind="port", host=host_info(),
                sample=f"{n} oracle update() calls in {el:.1f}s after filling the deque to {len(orc.buffer)} rows "
                       f"({fill_s:.1f}s); {w['kind']} B={w['B']} H={w['H']} L={w['L']}; torch {threads} threads")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE anything in
    this process touches the GPU, wait for them, relay rank 0's JSON line, exit with the worst status.  (Never
    exec: a process that has initialised the GPU must not be replaced, and this one has not even done that.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; the children are polled so that one failing rank ends the others (a rank that
    # died leaves its peers blocked in a collective) instead of hanging this launcher
    import threading
    chunks = []
    t = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    t.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = (r, p.returncode)
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    rcs = [p.wait() for p in procs]
    t.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed: {bad} (first: {failed})")


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return dict(cpu_model=model, logical_cpus=os.cpu_count(), usable_cpus=avail)


def build_agent(w, args, rank, local_rank, batch=None):
    import gcrl_amd
    from gcrl_amd.src.dp import rank_seed
    cfg = make_cfg(dict(w, B=batch or w["B"]))
    cls = dict(DDPG=gcrl_amd.DDPG, TD3=gcrl_amd.TD3Agent, SAC=gcrl_amd.SACAgent, TQC=gcrl_amd.TQCAgent)[w["kind"]]
    extra = {}
    if w.get("n_quantiles", 1) > 1:
        extra = dict(n_quantiles=w["n_quantiles"], num_critics=w["num_critics"], top_quantiles_to_drop=w["top_drop"])
    agent = cls(w["S"], w["A"], cfg, None, nenvs=8, gradient_step=w["gstep"], use_graph=not args.no_graph,
                pipeline=(True if args.pipeline < 0 else args.pipeline), rng=args.rng, seed=rank_seed(1898, rank),
                device_index=local_rank, **extra)
    pool = episode_pool(w, 64, seed=1898 + rank)
    arrays = [episode_arrays(ep) for ep in pool]
    rows_per_ep = 50 + w["k"] * 49
    n_eps = -(-w["cap"] // rows_per_ep)          # fill to capacity before timing
    t_fill = time.perf_counter()
    for ep in range(n_eps):
        s, a, ns, r, d, ag = arrays[ep % len(arrays)]
        agent.buffer.push_episode(ep % 8, s, a, ns, r, d, ag)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill
    assert len(agent.buffer) == min(w["cap"], n_eps * rows_per_ep)
    return agent, pool, t_fill


def timed_region(agent, dp, w, steps, warmup, step0=1):
    """W warm-up steps, one untimed call of every chunk size the timed region will issue (lazily built graphs and
    buffers exist before the clock starts), then exactly `steps` steps between barrier + synchronize pairs."""
    import torch.distributed as dist
    gstep = w["gstep"]

    def run(s0, n):
        done = 0
        while done < n:
            m = min(gstep, n - done)
            (agent if dp is None else dp).update_many(s0 + done, m)
            done += m

    if dp is not None:
        dist.barrier()          # the ranks finish filling their rings seconds apart: start the first exchange together
    run(step0, warmup)
    extra = 0
    for m in sorted({min(gstep, steps), steps % gstep} - {0}):
        run(step0 + warmup + extra, m)
        extra += m
    s0 = step0 + warmup + extra
    torch.cuda.synchronize()
    if dp is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(s0, steps)
    torch.cuda.synchronize()
    if dp is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, s0 + steps, extra, run


def _meeting_forms(agent):
    try:
        m = int(agent.meetings())
    except Exception:
        return None
    names = []
    if m & 1:
        names.append("BatchNorm slab row groups")
    if m & 2:
        names.append("row-chain roles (DDPG: the critic phase as two roles of the fused launch)")
    if m & 4:
        names.append("weight-slice DDPG launch (GCRL_ROWTILE=1)")
    if m & 8:
        names.append("fused dW + clip + optimiser launch (csrc/dw_adam.hip: norm slots swept inside the launch)")
    return names


def pmc_traffic(workload, kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/r0N_pmc_traffic_<workload>.json — the newest round's — made by tools/pmc_traffic.sh: FETCH_SIZE x 2 + WRITE_SIZE per
    the guide's gfx950 correction).  bench.py cannot run the counter tool on itself."""
    for rnd in ("r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{workload}.json")
        if not os.path.exists(path):
            continue
        for name, d in json.load(open(path)).items():
            if kernel_substr in name and "hbm_bytes_per_launch_corrected" in d:
                return dict(d, source=os.path.basename(path))
    return None


KERNELS = {   # short name -> substring of the profiler's kernel name
    "gather_main": "her_gather_update_kernel<false", "gather_head": "her_gather_update_kernel<true",
    "flush_single": "her_flush_kernel<false>", "flush_multi": "her_flush_kernel<true>", "rowchain": "rowchain_ddpg_kernel",
    "dw_adam": "dw_adam_kernel", "dw_gemm": "gemm_batch_kernel<1, 1, 4>", "adam_pair": "adam_pair_kernel",
}


def profiler_child(args, w):
    """Run this same workload once as a child under `rocprofv3 --kernel-trace --stats` and return the tool's per-kernel
    summary {short name: {calls, avg_us, min_us, max_us}} (+ the command).  Must be called BEFORE this process touches the
    GPU.  The child is the program itself after `--` (never a shell or env hop).  Any failure -> (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    # under a profiler already (rocprofv3 -- python3 bench.py ...: its preloaded tool library has initialised the GPU before this
    # interpreter started) a child may not be exec'ed from here: report the event clock instead
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process itself runs under a profiler"
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 not found"
    keep = args.keep_profile
    out_dir = os.path.abspath(keep) if keep else tempfile.mkdtemp(prefix="gcrl_prof_", dir="/tmp")
    os.makedirs(out_dir, exist_ok=True)
    steps = 20 * w["gstep"]
    cmd = [tool, "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "-o", "p", "--", sys.executable,
           os.path.abspath(__file__), "--workload", args.workload, "--steps", str(steps), "--warmup", str(2 * w["gstep"]),
           "--no-cpu-baseline", "--no-profiler", "--rng", args.rng, "--pipeline", str(args.pipeline)] + (["--no-graph"] if args.no_graph else [])
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    except (OSError, subprocess.TimeoutExpired) as e:
        return None, f"{type(e).__name__}: {e}"[:300]
    files = glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True)
    if r.returncode != 0 or not files:
        return None, f"rocprofv3 rc={r.returncode}: " + r.stderr.decode("utf-8", "replace")[-300:]
    stats = {}
    for row in csv.DictReader(open(files[0])):
        for short, sub in KERNELS.items():
            if sub in row["Name"]:
                stats[short] = dict(calls=int(row["Calls"]), avg_us=float(row["AverageNs"]) / 1e3, min_us=float(row["MinNs"]) / 1e3,
                                    max_us=float(row["MaxNs"]) / 1e3)
    for f in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)                         # tens of MB; the statistics are what is read (and kept)
    if not keep:
        shutil.rmtree(out_dir, ignore_errors=True)
    return dict(kernel_stats=stats, command=" ".join(["rocprofv3 --kernel-trace --stats --"] + [os.path.basename(c) if c == sys.executable else c for c in cmd[cmd.index("--") + 1:]]).replace(ROOT + "/", ""),
                steps=steps), None


def push_leg(w):
    """Push-side rates on a small ring of the workload's shape: (a) whole-episode pushes (one upload + one single-episode
    flush launch each), (b) the trainer's vector-env step (`push_batch`, 8 envs in lock-step: every 50th step flushes 8
    episodes in ONE launch).  Wall clock around enqueue + completion; rows = ring rows appended."""
    import gcrl_amd
    out = {}
    pool = [episode_arrays(ep) for ep in episode_pool(w, 16, seed=7)]
    rows_per_ep = 50 + w["k"] * 49
    buf = gcrl_amd.HERBuffer(400_000, 50, 8, k_future=w["k"], rng="engine", seed=3)
    for warm in (True, False):
        n_eps = 64 if warm else 640
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(n_eps):
            s, a, ns, r, d, ag = pool[ep % len(pool)]
            buf.push_episode(ep % 8, s, a, ns, r, d, ag)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    out["episode_push"] = dict(rows_per_s=n_eps * rows_per_ep / el, transitions_per_s=n_eps * 50 / el, us_per_episode=1e6 * el / n_eps)
    buf2 = gcrl_amd.HERBuffer(400_000, 50, 8, k_future=w["k"], rng="engine", seed=3)
    stk = [np.stack([pool[e][j] for e in range(8)]) for j in range(6)]     # [8][50][...]
    dev_s = torch.from_numpy(stk[0]).cuda(); dev_ns = torch.from_numpy(stk[2]).cuda()
    for warm in (True, False):
        n_steps = 100 if warm else 1000
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            t = i % 50
            buf2.push_batch(dev_s[:, t], stk[1][:, t], dev_ns[:, t], stk[3][:, t], np.zeros(8, bool), stk[5][:, t])
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    out["vector_step_push"] = dict(envs=8, rows_per_s=(n_steps // 50) * 8 * rows_per_ep / el, transitions_per_s=n_steps * 8 / el,
                                   us_per_vector_step=1e6 * el / n_steps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--workload", default="ddpg_pickplace_b256", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--rng", default="engine", choices=["engine", "device"], help="index streams: CPython-exact MT (default) or the counter hash")
    ap.add_argument("--pipeline", type=int, default=-1, help="schedule level (see gcrl_agent_config.pipeline_steps); -1: the default")
    ap.add_argument("--no-extra-legs", action="store_true", help="N > 1: skip the strong-scaling and cfg-5 (SAC) legs")
    ap.add_argument("--no-profiler", action="store_true", help="skip the rocprofv3 child run (kernel durations then come from HIP events)")
    ap.add_argument("--keep-profile", default=None, help="directory that keeps the child's rocprofv3 summary (kernel_stats.csv)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)       # nothing above has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("GCRL_BENCH_SPAWN_ECHO"):   # CPU test of the launch plumbing: report the rank environment, touch nothing
        if rank == 0:
            print(json.dumps(dict(rank=rank, world=world, master=os.environ.get("MASTER_ADDR"), port=os.environ.get("MASTER_PORT"))))
        raise SystemExit(int(os.environ["GCRL_BENCH_SPAWN_ECHO"]) if rank == world - 1 else 0)
    prof, prof_why = None, "disabled (--no-profiler)" if args.no_profiler else "N > 1"
    if world == 1 and not args.no_profiler and not int(os.environ.get("GCRL_FORCE_DP", "0")):
        prof, prof_why = profiler_child(args, WORKLOADS[args.workload])      # nothing above has touched the GPU
    # one rank per GPU; GCRL_DIST_BACKEND=gloo lets several ranks rehearse the DP path on ONE GPU
    backend = os.environ.get("GCRL_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > max(1, ndev):
        raise SystemExit(f"--gpus {world} over RCCL needs {world} GPUs, {ndev} visible (GCRL_DIST_BACKEND=gloo shares one GPU)")
    local_rank = local_rank % max(1, ndev)
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # GCRL_FORCE_DP=1: take the data-parallel path (process group + exchanges) even at world size 1,
    # to rehearse the RCCL plumbing on a one-GPU box; never set by the driver
    force_dp = bool(int(os.environ.get("GCRL_FORCE_DP", "0")))
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import gcrl_amd  # noqa: F401
    from gcrl_amd._ffi import lib, check
    from gcrl_amd.src.dp import DataParallelUpdater

    w = WORKLOADS[args.workload]
    gstep = w["gstep"]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        agent, pool, t_fill = build_agent(w, args, rank, local_rank)
        # over RCCL the gradient exchange must be the engine's own (a broken native path may not hide behind the slower
        # per-exchange Python loop); the fallback is taken only when asked for (GCRL_DP_PYTHON_EXCHANGE, gloo rehearsals)
        strict = backend == "nccl" and "GCRL_DIST_BACKEND" not in os.environ and not int(os.environ.get("GCRL_DP_PYTHON_EXCHANGE", "0"))
        dp = DataParallelUpdater(agent, require_native=strict) if (world > 1 or force_dp) else None
        elapsed, next_step, extra_warm, run = timed_region(agent, dp, w, args.steps, args.warmup)

        # ---- untimed measurement legs -------------------------------------------------------
        # HER gather kernel: a hipEvent pair around every gather launch of 10 trainer cycles (on the launch stream)
        her = agent.buffer.handle
        check(lib.gcrl_her_profile_enable(her, 1))
        run(next_step, 10 * gstep)
        next_step += 10 * gstep
        launches, ms, rows, dev_ms = C.c_int64(), C.c_double(), C.c_int64(), C.c_double()
        check(lib.gcrl_her_profile_read(her, C.byref(launches), C.byref(ms), C.byref(rows), C.byref(dev_ms)))
        check(lib.gcrl_her_profile_enable(her, 0))
        # row-chain kernel: the engine's measurement hooks (plain launches, every row-chain launch bracketed by
        # hipEvents on its stream + device clock stamps)
        rc = None
        if w["kind"] in ("DDPG", "TD3") and dp is None and w["H"] % 4 == 0:
            check(lib.gcrl_agent_profile_enable(agent._h, 1))
            run(next_step, 10 * gstep)
            next_step += 10 * gstep
            n_l, ev_ms, clk_ms = C.c_int64(), C.c_double(), C.c_double()
            check(lib.gcrl_agent_profile_read(agent._h, C.byref(n_l), C.byref(ev_ms), C.byref(clk_ms)))
            check(lib.gcrl_agent_profile_enable(agent._h, 0))
            if n_l.value:
                rc = (n_l.value, ev_ms.value * 1e3 / n_l.value, clk_ms.value * 1e3 / n_l.value)
        # a last metrics fetch proves the steps really ran to completion
        last = [float(x) for x in (agent.update_many(next_step, 1)[0] if dp is None else dp.update(next_step))]
        next_step += 1
        assert all(np.isfinite(last)), last

        push = None
        if rank == 0 and world == 1 and not force_dp:
            try:
                push = push_leg(w)
            except Exception as e:   # noqa: BLE001  (a reporting leg never costs the headline line)
                push = dict(error=f"{type(e).__name__}: {e}"[:300])

    out = None
    if rank == 0:
        R = (2 * w["S"] + w["A"] + 2) * 4
        alg_bytes_per_row = 2 * R                      # read the record + write the batch row
        rows_per_launch = rows.value / max(1, launches.value)
        ev_us = ms.value * 1e3 / max(1, launches.value)
        ks = (prof or {}).get("kernel_stats", {})
        g_us = ks.get("gather_main", {}).get("avg_us")
        clock = "rocprofv3 kernel duration (child run of this command, profiler.command)" if g_us else \
                f"HIP-event pair (profiler unavailable: {prof_why}); includes dispatch overhead"
        use_us = g_us or ev_us
        achieved = (alg_bytes_per_row * rows_per_launch) / (use_us * 1e-6) / 1e9 if use_us > 0 else 0.0
        g_pmc = pmc_traffic(args.workload, "her_gather_update_kernel<false") or pmc_traffic(args.workload, "her_gather")
        traffic = None
        if g_pmc and g_pmc.get("rows_per_launch"):
            traffic = g_pmc["hbm_bytes_per_launch_corrected"] / g_pmc["rows_per_launch"] * rows_per_launch
        sync_rate = args.steps / elapsed
        value = world * sync_rate
        out = {
            "metric": "gradient-steps/sec (HER sample + critic+actor update)",
            "value": value, "unit": "gradient-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "agent": w["kind"], "batch_per_gpu": w["B"], "buffer_rows": len(agent.buffer),
                       "state_dim": w["S"], "action_dim": w["A"], "hidden": w["H"], "layers": w["L"], "k_future": w["k"],
                       "gradient_step": gstep, "parallelism": f"dp{world}" if world > 1 else "single",
                       "hip_graph": not args.no_graph,
                       # launch forms with in-kernel waits that are ACTIVE (csrc/meet.h; gcrl_agent_set_meetings' mask)
                       "in_kernel_meetings": _meeting_forms(agent),
                       "rng": "cpython-mt19937 (host) indices" if args.rng == "engine" else "counter-hash indices (in-kernel HER picks)",
                       "reward": "built-in sparse goal-distance reward (panda-gym absent: the only parity-unpinned seam, DESIGN.md §2)"},
            "value_semantics": ("one agent.update(step) on a batch of %d rows per second" % w["B"]) if world == 1 else
                               ("weak scaling: every rank draws its own batch of %d rows and the gradients are all-reduced, so ONE "
                                "synchronised optimiser step consumes %d x %d rows; value = samples/s / %d = n_gpus x "
                                "sync_optimizer_steps_per_s (batch-%d-equivalent gradient steps)" % (w["B"], world, w["B"], w["B"], w["B"])),
            "sync_optimizer_steps_per_s": sync_rate,
            "warmup_extra_steps": extra_warm,
            "roofline_gather": {"kernel": "her_gather_update_kernel<false> (a trainer cycle's main gather)", "bound": "hbm", "achieved": achieved,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                "traffic_source": (g_pmc or {}).get("source"),
                                "avg_launch_us": use_us, "rows_per_launch": rows_per_launch,
                                "algorithmic_bytes_per_row": alg_bytes_per_row, "timing": clock,
                                "hip_event_us": ev_us, "hip_event_launches": launches.value,
                                "achieved_hip_event": (alg_bytes_per_row * rows_per_launch) / (ev_us * 1e-6) / 1e9 if ev_us > 0 else None,
                                "head_launch": ({"kernel": "her_gather_update_kernel<true>", "rows": head_batches(gstep, w["kind"]) * w["B"], **ks["gather_head"]} if "gather_head" in ks else None),
                                "profiler": ks.get("gather_main"),
                                "note": "a call's first launch gathers its first %d batches (indices read from the pinned upload block, carries the control "
                                        "block) so that the first steps start while the host draws the rest; the main launch gathers the other "
                                        "gradient_step - %d batches." % (head_batches(gstep, w["kind"]), head_batches(gstep, w["kind"])) + "  Below ~1e5 rows a launch is bounded by the ~1.3 us dispatch + two dependent "
                                        "memory latencies, not by bandwidth (profiles/r03_gather_rows_curve.txt)"},
            "update_flops": {"gflop_per_step": flops_per_step(w) / 1e9,
                             "achieved_tflops": flops_per_step(w) * args.steps / elapsed / 1e12,
                             "peak_tflops": FP32_MFMA_PEAK_TF},
            "fill_s": t_fill, "last_metrics": last, "host": host_info(),
        }
        if dp is not None:
            out["dp_exchange"] = dp.exchange
            out["dp_exchange_reason"] = getattr(dp, "exchange_reason", "")
        # HER relabel + flush kernel: per episode read T*(2S+A+2+G)*4 B of staging, write (T + k(T-1)) rows of R bytes
        T, G = 50, 3
        ep_bytes = T * (2 * w["S"] + w["A"] + 2 + G) * 4 + (T + w["k"] * (T - 1)) * R
        fl = {"kernel": "her_flush_kernel", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_episode": ep_bytes,
              "fill_rows_per_s": len(agent.buffer) / t_fill if t_fill > 0 else None,
              "note": "launch-bound by construction: an episode is %d KB (0.01 us at 8 TB/s) against a ~10 us launch; a vector-env "
                      "step's episodes share one launch (up to 8)" % (ep_bytes // 1024)}
        for key, eps in (("flush_single", 1), ("flush_multi", 8)):
            if key in ks:
                us = ks[key]["avg_us"]
                fl[key] = dict(episodes_per_launch=eps, **ks[key], achieved=eps * ep_bytes / (us * 1e-6) / 1e9, frac=eps * ep_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        if "flush_single" in fl:
            fl["achieved"], fl["frac"] = fl["flush_single"]["achieved"], fl["flush_single"]["frac"]
        if push is not None:
            fl["push"] = push
        out["roofline_flush"] = fl
        if prof is not None:
            out["profiler"] = {"tool": "rocprofv3 --kernel-trace --stats", "command": prof["command"], "steps": prof["steps"], "kernel_stats": ks}
        else:
            out["profiler"] = {"unavailable": prof_why}
        if rc is not None and w["kind"] == "DDPG":
            fl_ = chain_flops_per_launch(w)
            r_pmc = pmc_traffic(args.workload, "rowchain")
            r_us = ks.get("rowchain", {}).get("avg_us") or rc[1]
            out["roofline"] = {
                "kernel": "rowchain_ddpg_kernel", "bound": "mfma", "achieved": fl_ / (r_us * 1e-6) / 1e12,
                "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": fl_ / (r_us * 1e-6) / 1e12 / FP32_MFMA_PEAK_TF,
                "traffic": r_pmc["hbm_bytes_per_launch_corrected"] if r_pmc else None,
                "traffic_source": (r_pmc or {}).get("source"),
                "l2_to_cu_read_requests_per_launch": r_pmc.get("TCP_TCC_READ_REQ_sum") if r_pmc else None,
                "avg_launch_us": r_us, "algorithmic_flops_per_launch": fl_,
                "timing": "rocprofv3 kernel duration (child run of this command)" if "rowchain" in ks else "HIP-event pair (profiler unavailable)",
                "profiler": ks.get("rowchain"),
                "hip_event_us": rc[1], "hip_event_launches": rc[0], "device_clock_us": rc[2],
                "share_of_step_time": r_us / (1e6 * elapsed / args.steps),
                "note": "latency-bound chain (10 dependent 256-wide layer passes per phase, 128 of 256 CUs at B=256), see DESIGN.md §4"}
            if "dw_adam" in ks:
                # the step's other launch (round 5): every dW | db GEMM, the global-norm clip, Adam, Polyak, weight copies, metrics
                S_, A_, H_, L_ = w["S"], w["A"], w["H"], w["L"]
                n_par = (S_ * H_ + (L_ - 1) * H_ * H_ + H_ * A_ + L_ * H_ + A_) + ((S_ + A_) * H_ + (L_ - 1) * H_ * H_ + H_ + L_ * H_ + 1)   # actor + critic, weights and biases
                o_us = ks["dw_adam"]["avg_us"]
                out["optimiser_launch"] = {
                    "kernel": "dw_adam_kernel (dW | db of both nets + clip + Adam + Polyak + [in][out] copies + metrics + control advance)",
                    "bound": "latency (one cold operand round trip, two hand-offs through memory inside the launch)",
                    "avg_launch_us": o_us, "profiler": ks["dw_adam"],
                    "algorithmic_flops_per_launch": 2 * w["B"] * n_par, "achieved_tflops": 2 * w["B"] * n_par / (o_us * 1e-6) / 1e12,
                    "algorithmic_bytes_per_launch": 28 * n_par + 2 * 4 * w["B"] * (w["H"] * 2 * w["L"]),
                    "replaces": "gemm_batch_kernel<1,1,4> (7.2 us) + adam_pair_kernel (8.8 us): profiles/r04_kernel_stats_ddpg_pickplace_b256.csv",
                    "share_of_step_time": o_us / (1e6 * elapsed / args.steps)}
        else:
            # many-kernel steps (TD3 / SAC / TQC): the step as a whole against the MFMA roofline, not one kernel
            tf = flops_per_step(w) * args.steps / elapsed / 1e12
            out["roofline"] = {"kernel": "whole update step (all launches; per-kernel shares in profiles/)", "bound": "mfma",
                               "achieved": tf, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TF,
                               "traffic": None, "timing": "algorithmic flops per step / measured step time"}
        if not args.no_cpu_baseline and world == 1 and w.get("n_quantiles", 1) == 1:
            out["cpu_baseline"] = cpu_baseline(w, pool, args.cpu_seconds)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
            if w["cap"] > 100_000:
                # BASELINE.md's sanity row is quoted with a 1e5-row deque: random.sample(deque) walks the deque, so the 1e6-row
                # baseline above is slower for a reason that has nothing to do with the update arithmetic — report both
                out["cpu_baseline_n1e5"] = cpu_baseline(dict(w, cap=100_000), pool, max(4.0, args.cpu_seconds / 2))
                out["speedup_vs_cpu_port_n1e5"] = value / out["cpu_baseline_n1e5"]["value"]
        print(json.dumps(out))
        sys.stdout.flush()

    if world > 1 and not args.no_extra_legs:
        # Extra legs AFTER the headline line is out (a leg that fails or hangs can no longer cost it); their results go to
        # stderr as one line.  After each leg every rank learns whether all ranks finished it (MIN over a flag): a rank-local
        # failure ends the legs on every rank together instead of leaving the others inside a collective.
        legs = {}
        with torch.cuda.stream(stream):
            def leg(name, fn):
                ok = 1
                try:
                    res = fn()
                except Exception as e:   # noqa: BLE001
                    res, ok = dict(error=f"{type(e).__name__}: {e}"[:400]), 0
                flag = torch.tensor([ok], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                legs[name] = res if int(flag.item()) or not ok else dict(error="another rank failed this leg")
                return bool(int(flag.item()))

            def strong():
                # strong scaling: the metric's batch of B rows split over the ranks (B/N rows each), same ring size
                ag_s, _, _ = build_agent(w, args, rank, local_rank, batch=w["B"] // world)
                dp_s = DataParallelUpdater(ag_s, require_native=strict)
                n_s = max(gstep, min(args.steps, 10 * gstep))
                el_s, _, _, _ = timed_region(ag_s, dp_s, w, n_s, min(args.warmup, 2 * gstep))
                return dict(global_batch=w["B"], batch_per_gpu=w["B"] // world, steps=n_s, value=n_s / el_s, unit="gradient-steps/s",
                            ms_per_step=1e3 * el_s / n_s,
                            note="fixed global batch: a latency-bound chain of 256-wide layers does not get shorter with fewer rows "
                                 "per GPU, and gains two exchanges (DESIGN.md §7)")

            def cfg5():
                # BASELINE cfg 5: SAC Slide, B=512 per GPU, 64 env streams over the ranks
                w5 = WORKLOADS["sac_slide_b512"]
                ag5, _, _ = build_agent(w5, args, rank, local_rank)
                dp5 = DataParallelUpdater(ag5, require_native=strict)
                n5 = max(w5["gstep"], min(args.steps, 5 * w5["gstep"]))
                el5, _, _, _ = timed_region(ag5, dp5, w5, n5, min(args.warmup, 2 * w5["gstep"]))
                return dict(batch_per_gpu=w5["B"], steps=n5, sync_optimizer_steps_per_s=n5 / el5, value=world * n5 / el5,
                            unit="gradient-steps/s (batch-512 equivalents)", ms_per_step=1e3 * el5 / n5,
                            batchnorm="local statistics per rank (DESIGN.md §7)", dp_exchange=dp5.exchange)

            alive = True
            if w["B"] % world == 0:
                alive = leg("strong_scaling", strong)
            if alive:
                leg("cfg5_sac_slide_b512", cfg5)
        if rank == 0:
            sys.stderr.write("bench-legs: " + json.dumps(legs) + "\n")
            sys.stderr.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
