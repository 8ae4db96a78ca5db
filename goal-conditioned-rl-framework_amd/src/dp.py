"""Data-parallel update over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  New design — the reference is single-process.

One process per GPU.  Env streams are sharded `nenvs / world` per rank; each rank owns a local
HER ring and draws its own batch of B rows (seed + rank); parameters, optimiser state and
targets are replicas kept identical by construction: identical initial parameters (broadcast
from rank 0) and identical post-exchange gradients.  The step is cut at the two gradient
exchanges (csrc/agent.hip phases):

    phase 0  sample + critic forward/backward            -> all-reduce(sum) critic gradients
    phase 1  critic clip/Adam/Polyak + actor fwd/bwd     -> all-reduce(sum) actor (+log_alpha) grads
    phase 2  actor clip/Adam (+alpha step, actor Polyak)

with `grad_scale = 1/world` applied inside the optimiser kernels, so G ranks x B rows equal one
rank x G*B rows up to fp32 summation order (DDPG/TD3).  SAC/TQC actors keep LOCAL BatchNorm
statistics per rank (stated divergence from a single big batch).  Each exchange is ONE flat
buffer: messages are 37 KB - 11 MB, latency-bound on point-to-point xGMI, so fewer, larger
collectives beat per-tensor ones.

Round 4: the default on one node is the engine's own peer-to-peer exchange kernel over IPC-mapped gradient arenas
(`exchange="ipc"`, csrc/xchg_ipc.hip): the all-reduce is a launch of the update's own sequence, so `update_many` under data
parallelism IS the single-GPU entry point (multi-step hipGraphs, control-advance riders, deferred draws) with two more kernels per
step (one per overlapped DDPG step), and the clip norm's partials come back with the reduced gradients.

Over RCCL (`exchange="rccl"`) the exchange lives in the engine too: `DataParallelUpdater` creates a library-owned
communicator (`gcrl_dp_create`, csrc/dp_rccl.cc; the 128-byte id travels through torch.distributed's store)
and a whole trainer cycle is ONE native call (`gcrl_agent_dp_run_all`) that enqueues every graph segment and
every all-reduce on the engine's stream — no Python round trip per exchange.  With any other backend (gloo in
the tests) the Python loop below moves the bytes instead; the schedule is the same.

For runs of plain DDPG steps the engine schedules the software-pipelined form instead (actor phase
of step i in the same launches as the critic phase of step i+1): both gradient blocks are ready at
the same point and adjacent in memory, so a step costs ONE all-reduce.  The schedule lives in the
engine (`gcrl_agent_dp_run` names the block to exchange after each segment); this module only
moves the bytes, with whatever torch.distributed backend the process group has.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist


def shard_env_streams(nenvs: int, rank: int, world: int) -> range:
    """Contiguous block of env ids owned by `rank` (64 envs / 8 ranks -> 8 each)."""
    base, extra = divmod(nenvs, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def rank_seed(seed: int, rank: int) -> int:
    return int(seed) + int(rank)


def allreduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks, then divide: what grad_scale does inside the engine."""
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    return flat


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    dist.broadcast(flat, src=src, group=group)
    return flat


class _DevVec:
    """Zero-copy torch view of a device vector owned by the engine."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_view(ptr: int, numel: int) -> torch.Tensor:
    return torch.as_tensor(_DevVec(ptr, numel), device="cuda")


class DataParallelUpdater:
    """Drives agent.update(step) across ranks.  `agent` is a gcrl_amd agent on this rank's GPU."""

    def __init__(self, agent, group=None, require_native: bool = False, sync_bn: bool = False, exchange: str = "auto"):
        """`exchange` — who moves the gradient bytes (`self.exchange` says which one runs):
          "ipc"     "engine-ipc": the engine's own peer-to-peer kernel over IPC-mapped gradient arenas (csrc/xchg_ipc.hip): the
                    exchange is a launch of the step's sequence — hipGraph replay, multi-step graphs and the control-advance riders
                    of the single-GPU path stay on, the clip norm's partials come back with the reduced gradients.  One node,
                    world <= 8; any torch.distributed backend (it only carries the handles);
          "rccl"    "engine-rccl": a library-owned RCCL communicator, one native call per trainer cycle (csrc/dp_rccl.cc; needs
                    the nccl backend and one GPU per rank);
          "python"  torch.distributed calls between the engine's segments (any backend; the tests' gloo rehearsals);
          "auto"    ipc when every rank is on this host, else rccl over the nccl backend, else python.
        GCRL_DP_EXCHANGE in the environment overrides the argument (bench.py's A/B legs); GCRL_DP_PYTHON_EXCHANGE=1 = "python".
        `require_native`: raise on EVERY rank when the asked-for in-engine exchange cannot be set up on any of them instead of
        falling back to torch.distributed calls (bench.py: a broken native path must not hide behind a slower one).
        `sync_bn` (SACAgent / TQCAgent): BatchNorm statistics over the concatenated batch of all ranks (gcrl_agent_dp_sync_bn)
        instead of each rank's own rows — G ranks x B rows then equal 1 rank x G*B rows for these agents too, at the price of
        one small exchange per BatchNorm layer and pass on the step's critical path."""
        from .. import _ffi
        self._ffi = _ffi
        self.agent = agent
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.scale = 1.0 / self.world
        lib = _ffi.lib
        self._views = {}
        self._blocks = []
        self._block_ptrs = []
        self._native = None
        self._xchg = None
        exchange = os.environ.get("GCRL_DP_EXCHANGE", exchange)
        if int(os.environ.get("GCRL_DP_PYTHON_EXCHANGE", "0")):
            exchange = "python"
        if exchange not in ("auto", "ipc", "rccl", "python"):
            raise ValueError(f"exchange must be auto | ipc | rccl | python, got {exchange!r}")
        nccl = dist.get_backend(group) == "nccl"
        flag_dev = "cuda" if nccl else "cpu"

        def all_ok(ok: bool) -> bool:      # every rank must take the same path
            t = torch.tensor([1 if ok else 0], device=flag_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        # who shares what: hosts (IPC handles only mean something on one node) and devices (launch forms whose workgroups wait
        # for each other inside a kernel assume the process has the GPU to itself: csrc/meet.h)
        import socket
        props = torch.cuda.get_device_properties(agent.device_index)
        dev_id = str(getattr(props, "uuid", "")) or f"{getattr(props, 'pci_bus_id', agent.device_index)}"
        where = [None] * self.world
        dist.all_gather_object(where, (socket.gethostname(), dev_id), group=group)
        one_host = len({h for h, _ in where}) == 1
        shared_gpu = len(set(where)) < self.world
        auto = exchange == "auto"
        self.exchange_reason = "asked for" if not auto else ""
        if auto:
            # Under the nccl backend `auto` takes the library collective: the peer-to-peer exchange has only ever run
            # with its ranks on ONE device (1-GPU boxes) — opt in with exchange="ipc" / GCRL_DP_EXCHANGE=ipc; its three-round
            # self-test then decides, and a failure falls back to RCCL on every rank together.  Under gloo (CPU rendezvous, ranks
            # sharing a GPU: the tests) the peer-to-peer exchange is the in-engine path there is.
            if nccl:
                exchange, self.exchange_reason = "rccl", "auto: nccl backend (the IPC exchange is opt-in until validated across devices)"
            elif one_host and self.world <= 8:
                exchange, self.exchange_reason = "ipc", "auto: one host, gloo backend"
            else:
                exchange, self.exchange_reason = "python", "auto: several hosts without the nccl backend"
        if exchange == "ipc":
            why = ""
            x = lib.gcrl_agent_xchg_create(agent._h, self.rank, self.world) if (one_host and self.world <= 8) else None
            rec = (C.c_uint8 * _ffi.XCHG_HANDLE_BYTES)()
            ok = bool(x) and lib.gcrl_xchg_handles(x, rec, _ffi.XCHG_HANDLE_BYTES) == 0
            if not ok:
                why = _ffi.last_error() or "the ranks are not on one host / more than 8 ranks"
            recs = [None] * self.world
            dist.all_gather_object(recs, bytes(rec) if ok else b"", group=group)
            ok = all(len(r) == _ffi.XCHG_HANDLE_BYTES for r in recs)
            if ok:
                ok = lib.gcrl_xchg_connect(x, b"".join(recs), _ffi.XCHG_HANDLE_BYTES * self.world) == 0
                why = why or (_ffi.last_error() if not ok else "")
            if all_ok(ok):      # a known pattern through the real kernel, before anything depends on it (collective)
                dist.barrier(group=group)
                ok = lib.gcrl_xchg_selftest(x, _ffi.stream_handle()) == 0
                why = why or (_ffi.last_error() if not ok else "")
                if not all_ok(ok):
                    ok = False
                    if x:
                        lib.gcrl_xchg_reset(x)
            else:
                ok = False
            if ok:
                self._xchg = x
                _ffi.check(lib.gcrl_agent_set_exchange(agent._h, x))
            else:
                if x:
                    lib.gcrl_xchg_destroy(x)
                if require_native and not nccl:      # (over RCCL the other in-engine exchange is tried next)
                    raise _ffi.GcrlError("gcrl_amd.dp: the in-engine IPC exchange could not be set up on every rank (this rank: " + (why or "ok") +
                                         "); refusing a fallback (require_native)")
                import warnings
                warnings.warn("gcrl_amd.dp: in-engine IPC exchange unavailable (" + (why or "another rank failed") + "); falling back")
                exchange = "rccl" if nccl else "python"
                self.exchange_reason = "fallback from ipc: " + (why or "another rank failed")
        if exchange == "rccl" and nccl:
            # the librccl PyTorch itself uses (two RCCL / HIP runtime copies in one process do not mix)
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
            uid = (C.c_uint8 * 128)()
            ok = 1
            if self.rank == 0:
                ok = 1 if lib.gcrl_dp_unique_id(uid, path) == 0 else 0
            box = [bytes(uid), ok]
            dist.broadcast_object_list(box, src=0, group=group)
            h = lib.gcrl_dp_create(self.rank, self.world, box[0], agent.device_index, path) if box[1] else None
            # every rank must take the same path: the in-engine exchange only if ALL ranks got their communicator
            if all_ok(bool(h)):
                self._native = h
            else:
                why = _ffi.last_error()
                if h:
                    lib.gcrl_dp_destroy(h)
                if require_native:
                    raise _ffi.GcrlError("gcrl_amd.dp: the in-engine RCCL communicator could not be created on every rank (this rank: "
                                         + (why or "ok") + "); refusing the torch.distributed fallback (require_native)")
                import warnings
                warnings.warn("gcrl_amd.dp: in-engine RCCL communicator unavailable (" + why +
                              "); exchanging gradients through torch.distributed instead")
        elif exchange == "rccl" and require_native:
            raise _ffi.GcrlError("gcrl_amd.dp: exchange='rccl' needs the nccl backend")
        self.exchange = "engine-ipc" if self._xchg else ("engine-rccl" if self._native else "python")
        # waits between workgroups inside a launch (csrc/meet.h) assume an exclusive GPU: off when ranks share a device, and
        # under the Python exchange (torch's own collectives and copies run beside the engine's stream)
        if shared_gpu or self.exchange == "python":
            if shared_gpu:
                lib.gcrl_set_shared_device(1)
            _ffi.check(lib.gcrl_agent_set_meetings(agent._h, 0))
        self.sync_bn = bool(sync_bn) and agent._sac and self.world > 1
        self._bn_cb = None
        self._bn_xchg = None
        self.sync_bn_exchange = None
        if self.sync_bn and self._xchg:
            # round 5: the BatchNorm partials through the same peer-to-peer kernel as the gradients (an exchange handle over the
            # partials' own arena: connect, self-test, collective fallback to the paths below) — hipGraphs and multi-step graphs stay on
            bx = lib.gcrl_agent_bn_xchg_create(agent._h, self.rank, self.world)
            rec = (C.c_uint8 * _ffi.XCHG_HANDLE_BYTES)()
            ok = bool(bx) and lib.gcrl_xchg_handles(bx, rec, _ffi.XCHG_HANDLE_BYTES) == 0
            recs = [None] * self.world
            dist.all_gather_object(recs, bytes(rec) if ok else b"", group=group)
            ok = ok and all(len(r) == _ffi.XCHG_HANDLE_BYTES for r in recs)
            if ok:
                ok = lib.gcrl_xchg_connect(bx, b"".join(recs), _ffi.XCHG_HANDLE_BYTES * self.world) == 0
            if all_ok(ok):
                dist.barrier(group=group)
                ok = lib.gcrl_xchg_selftest(bx, _ffi.stream_handle()) == 0
                ok = all_ok(ok)
            else:
                ok = False
            if ok:
                self._bn_xchg = bx
                _ffi.check(lib.gcrl_agent_dp_sync_bn_xchg(agent._h, self.world, self.rank, bx))
                self.sync_bn_exchange = "engine-ipc"
            elif bx:
                lib.gcrl_xchg_destroy(bx)
        if self.sync_bn and not self._bn_xchg:
            self.sync_bn_exchange = "engine-rccl" if self._native else "python"
            if self._native:
                _ffi.check(lib.gcrl_agent_dp_sync_bn(agent._h, self.world, dist.get_rank(group), self._native, None, None))
            else:
                def _exchange(ptr, n, _stream, _user, self=self):
                    try:
                        dist.all_reduce(device_view(ptr, n), op=dist.ReduceOp.SUM, group=self.group)
                        return 0
                    except BaseException as e:   # noqa: BLE001  (must not propagate through the C frames)
                        self._bn_exc = e
                        return 1
                self._bn_exc = None
                self._bn_cb = _ffi.EXCHANGE_FN(_exchange)
                _ffi.check(lib.gcrl_agent_dp_sync_bn(agent._h, self.world, dist.get_rank(group), None, C.cast(self._bn_cb, C.c_void_p), None))
        for phase in (0, 1):
            p, n = C.c_void_p(), C.c_int64()
            _ffi.check(lib.gcrl_agent_grad_ptr(agent._h, phase, C.byref(p), C.byref(n)))
            self._blocks.append(device_view(p.value, n.value))
            self._block_ptrs.append((p.value, n.value))
        self.sync_parameters()

    def __del__(self):
        if getattr(self, "sync_bn", False) and getattr(self.agent, "_h", None):
            self._ffi.lib.gcrl_agent_dp_sync_bn(self.agent._h, 1, 0, None, None, None)   # the agent outlives the callback / communicator
        bx, self._bn_xchg = getattr(self, "_bn_xchg", None), None
        if bx:
            self._ffi.lib.gcrl_xchg_destroy(bx)
        x, self._xchg = getattr(self, "_xchg", None), None
        if x:
            if getattr(self.agent, "_h", None):
                self._ffi.lib.gcrl_agent_set_exchange(self.agent._h, None)
            self._ffi.lib.gcrl_xchg_destroy(x)
        h, self._native = getattr(self, "_native", None), None
        if h:
            self._ffi.lib.gcrl_dp_destroy(h)

    def recover(self):
        """COLLECTIVE: after a timed-out exchange (`GcrlError: ... exchange timed out waiting for a peer`) every rank calls this —
        barrier, counters of the peer-to-peer exchange back to their common initial state (gcrl_xchg_reset), barrier.  The step
        that timed out took NaN gradients: reload the last checkpoint (`agent.load_state`) on every rank afterwards, as a trainer
        would after any failed step (reference: src/agent.py:659-699 raises and stops).  A no-op for the RCCL / Python exchanges."""
        dist.barrier(group=self.group)
        if self._xchg:
            self._ffi.check(self._ffi.lib.gcrl_xchg_reset(self._xchg))
        if self._bn_xchg:
            self._ffi.check(self._ffi.lib.gcrl_xchg_reset(self._bn_xchg))
        dist.barrier(group=self.group)

    def _allreduce_block(self, i: int, st):
        if self._native:
            p, n = self._block_ptrs[i]
            self._ffi.check(self._ffi.lib.gcrl_dp_allreduce_sum(self._native, p, n, st))
        else:
            dist.all_reduce(self._blocks[i], op=dist.ReduceOp.SUM, group=self.group)

    def sync_parameters(self):
        """Rank 0's parameters (and BN statistics, log_alpha) become everyone's."""
        lib, ffi = self._ffi.lib, self._ffi
        names = ["actor"] + [f"critic_{i}" for i in range(self.agent.num_critics)]
        if self.agent._sac:
            names += ["log_alpha", "bn_running_mean", "bn_running_var"]
        for name in names:
            p, n = C.c_void_p(), C.c_int64()
            ffi.check(lib.gcrl_agent_dev_ptr(self.agent._h, name.encode(), C.byref(p), C.byref(n)))
            broadcast_(device_view(p.value, n.value), 0, self.group)
        torch.cuda.synchronize()
        self.agent.update_target_network()

    def update_many(self, step0: int, n: int):
        """One trainer cycle (src/env.py:384-385) across ranks: the n batches of this rank are drawn
        and gathered by ONE launch; the engine then hands back graph segments and the gradient
        block to all-reduce after each (two per ordinary step, one per pipelined DDPG step)."""
        a, lib, ffi = self.agent, self._ffi.lib, self._ffi
        if self._xchg:      # the exchanges are launches of the engine's own sequence: the ordinary entry point, graphs and all
            return a.update_many(step0, n)
        her = a.buffer.handle
        st = ffi.stream_handle()
        tickets, lens = (C.c_int64 * n)(), (C.c_int32 * n)()
        a.buffer.rng.pull()
        ffi.check(lib.gcrl_agent_dp_begin(a._h, her, int(step0), int(n), self.scale, tickets, lens, st))
        a.buffer.rng.push_back()
        if self._native:   # segments and collectives enqueued by one native call
            ffi.check(lib.gcrl_agent_dp_run_all(a._h, self._native, st))
            return [a._tuple(int(t), int(l)) for t, l in zip(tickets, lens)]
        # the engine owns the schedule: run a segment, all-reduce the gradient block it names, repeat
        ptr, numel = C.c_void_p(), C.c_int64()
        more = 1
        while more:
            more = ffi.check(lib.gcrl_agent_dp_run(a._h, C.byref(ptr), C.byref(numel), st))
            if numel.value:
                key = (ptr.value, numel.value)
                view = self._views.get(key)
                if view is None:
                    view = self._views[key] = device_view(ptr.value, numel.value)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        return [a._tuple(int(t), int(l)) for t, l in zip(tickets, lens)]

    def update(self, step: int, batch=None, noise=None, eps_next=None, eps_cur=None):
        """One step.  `batch` = (s, a, r, ns, d) cuda tensors injects this rank's rows, `noise` / `eps_*` its
        TD3 smoothing noise / SAC-TQC reparameterisation draws (tests)."""
        a, lib, ffi = self.agent, self._ffi.lib, self._ffi
        if self._xchg:
            return a.update(step, batch=batch, noise=noise, eps_next=eps_next, eps_cur=eps_cur)
        ticket = C.c_int64(-1)
        inputs, keep = a._inject(batch, noise, eps_next, eps_cur)
        her = a.buffer.handle if batch is None else None
        if batch is None:
            a.buffer.rng.pull()
        st = ffi.stream_handle()
        n = ffi.check(lib.gcrl_agent_update_phase(a._h, her, int(step), 0, C.byref(inputs) if inputs is not None else None,
                                                  self.scale, C.byref(ticket), st))
        if batch is None:
            a.buffer.rng.push_back()
        self._allreduce_block(0, st)
        ffi.check(lib.gcrl_agent_update_phase(a._h, her, int(step), 1, None, self.scale, None, st))
        if n == {0: 6, 1: 8, 2: 9, 3: 9}[ffi_kind(a)]:   # tuple length of an actor step
            self._allreduce_block(1, st)
        ffi.check(lib.gcrl_agent_update_phase(a._h, her, int(step), 2, None, self.scale, None, st))
        return a._tuple(ticket.value, n)


def ffi_kind(agent) -> int:
    from .agent import KIND
    return KIND[agent.KIND_NAME]
