"""ctypes binding of libgcrl_hip.so (C ABI: include/gcrl.h).

The library is the product: there is no Python/CPU fallback.  Importing this module loads the
.so (building it is `__graft_entry__.build()` / `make -C csrc`); a missing library raises
ImportError, and every device entry point raises RuntimeError when no GPU is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCRL_HIP_LIB: development knob (A/B builds of the same library); the default is the in-tree build
LIB_PATH = os.environ.get("GCRL_HIP_LIB") or os.path.join(_HERE, "libgcrl_hip.so")

STREAM_LEGACY = 1  # GCRL_STREAM_LEGACY
XCHG_HANDLE_BYTES = 256  # GCRL_XCHG_HANDLE_BYTES


class GcrlError(RuntimeError):
    pass


class NotEnoughSamples(AssertionError):
    """len(buffer) < batch_size — the reference asserts (src/buffer.py:122)."""


class HerConfig(C.Structure):
    _fields_ = [
        ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("goal_dim", C.c_int32),
        ("capacity", C.c_int64), ("nenvs", C.c_int32), ("k_future", C.c_int32),
        ("flush_len", C.c_int32), ("reward_kind", C.c_int32), ("reward_threshold", C.c_float),
        ("device", C.c_int32), ("rng_mode", C.c_int32), ("seed", C.c_uint64),
    ]


class AgentConfig(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("obs_dim", C.c_int32), ("ac_dim", C.c_int32),
        ("hidden_dim", C.c_int32), ("layer_count", C.c_int32), ("batch_size", C.c_int32),
        ("num_critics", C.c_int32), ("top_drop", C.c_int32), ("ac_update_freq", C.c_int32),
        ("gradient_step", C.c_int32), ("polyak_every", C.c_int32),
        ("gamma", C.c_double), ("tau", C.c_double), ("grad_clip", C.c_double),
        ("policy_noise", C.c_double), ("noise_clamp", C.c_double),
        ("actor_lr", C.c_double), ("actor_lr_min", C.c_double), ("critic_lr", C.c_double),
        ("critic_lr_min", C.c_double), ("alpha_lr", C.c_double),
        ("ac_scheduler_steps", C.c_int64), ("cr_scheduler_steps", C.c_int64),
        ("alpha_min_steps", C.c_double),
        ("device", C.c_int32), ("use_graph", C.c_int32), ("seed", C.c_uint64),
        ("pipeline_steps", C.c_int32), ("n_quantiles", C.c_int32),
    ]


class UpdateInputs(C.Structure):
    _fields_ = [
        ("s_dev", C.c_void_p), ("ld_s", C.c_int32),
        ("a_dev", C.c_void_p), ("ld_a", C.c_int32),
        ("r_dev", C.c_void_p),
        ("ns_dev", C.c_void_p), ("ld_ns", C.c_int32),
        ("d_dev", C.c_void_p),
        ("noise_dev", C.c_void_p), ("eps_next_dev", C.c_void_p), ("eps_cur_dev", C.c_void_p),
        ("idx_host", C.c_void_p), ("weights_host", C.c_void_p),
    ]


_vp, _i32, _i64, _u32, _u64, _f32, _f64 = (C.c_void_p, C.c_int32, C.c_int64, C.c_uint32,
                                           C.c_uint64, C.c_float, C.c_double)
_cp = C.c_char_p

# name -> (restype, argtypes): every symbol include/gcrl.h declares
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
REWARD_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p)

PROTOTYPES = {
    "gcrl_last_error": (_cp, []),
    "gcrl_abi_version": (C.c_int, []),
    "gcrl_device_count": (C.c_int, []),
    "gcrl_mt_create": (_vp, []),
    "gcrl_mt_destroy": (None, [_vp]),
    "gcrl_mt_seed": (C.c_int, [_vp, _u64]),
    "gcrl_mt_get_state": (C.c_int, [_vp, _vp]),
    "gcrl_mt_set_state": (C.c_int, [_vp, _vp]),
    "gcrl_mt_getrandbits": (_u32, [_vp, C.c_int]),
    "gcrl_mt_randbelow": (_u32, [_vp, _u32]),
    "gcrl_mt_randint": (_i64, [_vp, _i64, _i64]),
    "gcrl_mt_random": (_f64, [_vp]),
    "gcrl_mt_sample_indices": (C.c_int, [_vp, _u32, _u32, _vp]),
    "gcrl_mt_future_indices": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "gcrl_cosine_lr_next": (_f64, [_f64, _f64, _f64, _i64, _i64]),
    "gcrl_ringbook_sim": (C.c_int, [_i64, _i64, _i64, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "gcrl_her_create": (_vp, [C.POINTER(HerConfig), _vp]),
    "gcrl_her_destroy": (None, [_vp]),
    "gcrl_her_len": (_i64, [_vp]),
    "gcrl_her_head": (_i64, [_vp]),
    "gcrl_her_staged": (_i32, [_vp, C.c_int]),
    "gcrl_her_stream": (_vp, [_vp]),
    "gcrl_her_push": (_i64, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _f32, C.c_int, _vp, _vp, _vp]),
    "gcrl_her_append": (_i64, [_vp, _vp, C.c_int, _vp, _f32, _vp, C.c_int, C.c_int, _vp]),
    "gcrl_her_push_batch": (_i64, [_vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "gcrl_her_push_episode": (_i64, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcrl_her_sample": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp]),
    "gcrl_her_read_rows": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gcrl_her_state_size": (_i64, [_vp]),
    "gcrl_her_save_state": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_her_load_state": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_her_profile_enable": (C.c_int, [_vp, C.c_int]),
    "gcrl_her_profile_read": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_i64), C.POINTER(_f64)]),
    "gcrl_agent_create": (_vp, [C.POINTER(AgentConfig)]),
    "gcrl_agent_destroy": (None, [_vp]),
    "gcrl_agent_profile_enable": (C.c_int, [_vp, C.c_int]),
    "gcrl_agent_profile_read": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_f64), C.POINTER(_f64)]),
    "gcrl_agent_stream": (_vp, [_vp]),
    "gcrl_agent_numel": (_i64, [_vp, _cp]),
    "gcrl_agent_get": (C.c_int, [_vp, _cp, _vp, _i64]),
    "gcrl_agent_set": (C.c_int, [_vp, _cp, _vp, _i64]),
    "gcrl_agent_init_weights": (C.c_int, [_vp, _u64, C.c_int]),
    "gcrl_agent_hard_update_targets": (C.c_int, [_vp]),
    "gcrl_agent_soft_update_targets": (C.c_int, [_vp, _f64, _vp]),
    "gcrl_agent_state_size": (_i64, [_vp]),
    "gcrl_agent_save_state": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_agent_load_state": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_agent_update": (C.c_int, [_vp, _vp, _i64, C.POINTER(UpdateInputs), C.POINTER(_i64), _vp]),
    "gcrl_agent_update_n": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp, _vp, _vp]),
    "gcrl_agent_metrics": (C.c_int, [_vp, _i64, _vp, C.c_int]),
    "gcrl_agent_update_phase": (C.c_int, [_vp, _vp, _i64, C.c_int, C.POINTER(UpdateInputs), _f32, C.POINTER(_i64), _vp]),
    "gcrl_agent_grad_ptr": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_i64)]),
    "gcrl_agent_dp_begin": (C.c_int, [_vp, _vp, _i64, C.c_int, _f32, _vp, _vp, _vp]),
    "gcrl_agent_dp_phase": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "gcrl_agent_dp_end": (C.c_int, [_vp, _vp]),
    "gcrl_agent_dp_run": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i64), _vp]),
    "gcrl_dp_unique_id": (C.c_int, [_vp, _cp]),
    "gcrl_dp_create": (_vp, [C.c_int, C.c_int, _vp, C.c_int, _cp]),
    "gcrl_dp_destroy": (None, [_vp]),
    "gcrl_dp_world": (C.c_int, [_vp]),
    "gcrl_dp_allreduce_sum": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gcrl_dp_broadcast": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "gcrl_agent_dp_run_all": (C.c_int, [_vp, _vp, _vp]),
    "gcrl_agent_dev_ptr": (C.c_int, [_vp, _cp, C.POINTER(_vp), C.POINTER(_i64)]),
    "gcrl_agent_act": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "gcrl_agent_act_host": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "gcrl_normalizer_create": (_vp, [C.c_int, _f64, _f64, C.c_int]),
    "gcrl_normalizer_destroy": (None, [_vp]),
    "gcrl_normalizer_size": (C.c_int, [_vp]),
    "gcrl_normalizer_update": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gcrl_normalizer_normalize": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "gcrl_normalizer_get": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gcrl_normalizer_set": (C.c_int, [_vp, _vp, _vp, _f64, _f64]),
    "gcrl_normalizer_set_float32": (C.c_int, [_vp, C.c_int]),
    "gcrl_normalizer_is_float32": (C.c_int, [_vp]),
    "gcrl_normalizer_set_rows_float64": (C.c_int, [_vp, C.c_int]),
    "gcrl_agent_observe_act": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]),
    "gcrl_her_process_step": (_i64, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gcrl_her_process_step_g": (_i64, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "gcrl_sort_truncate_mean": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp, _vp, _vp]),
    "gcrl_gemm_f32": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "gcrl_gemm_dw_split_f32": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "gcrl_bn_relu_fwd_f32": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gcrl_bn_relu_bwd_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "gcrl_bn_linear_slab_fwd_f32": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "gcrl_bn_linear_slab_bwd_f32": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "gcrl_her_set_reward_callback": (C.c_int, [_vp, _vp, _vp]),
    "gcrl_agent_dp_sync_bn": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "gcrl_agent_bn_xchg_create": (_vp, [_vp, C.c_int, C.c_int]),
    "gcrl_agent_dp_sync_bn_xchg": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "gcrl_xchg_create": (_vp, [_vp, _i64, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gcrl_xchg_destroy": (None, [_vp]),
    "gcrl_xchg_handles": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_xchg_connect": (C.c_int, [_vp, _vp, _i64]),
    "gcrl_xchg_world": (C.c_int, [_vp]),
    "gcrl_xchg_allreduce": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "gcrl_xchg_reset": (C.c_int, [_vp]),
    "gcrl_xchg_selftest": (C.c_int, [_vp, _vp]),
    "gcrl_xchg_read": (C.c_int, [_vp, _i64, _i64, _vp]),
    "gcrl_xchg_get_partials": (C.c_int, [_vp, C.c_int, _vp, C.c_int]),
    "gcrl_agent_xchg_create": (_vp, [_vp, C.c_int, C.c_int]),
    "gcrl_agent_set_exchange": (C.c_int, [_vp, _vp]),
    "gcrl_set_shared_device": (C.c_int, [C.c_int]),
    "gcrl_agent_set_meetings": (C.c_int, [_vp, C.c_int]),
    "gcrl_agent_get_meetings": (C.c_int, [_vp]),
    "gcrl_agent_debug_meet_fault": (C.c_int, [_vp]),
    "gcrl_hash_normal_fill": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int64, _vp, _vp]),
    "gcrl_event_create": (_vp, []),
    "gcrl_event_destroy": (None, [_vp]),
    "gcrl_event_record": (C.c_int, [_vp, _vp]),
    "gcrl_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(_f32)]),
    "gcrl_stream_synchronize": (C.c_int, [_vp]),
    "gcrl_malloc": (_vp, [C.c_size_t]),
    "gcrl_free": (None, [_vp]),
    "gcrl_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "gcrl_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C csrc). "
            "There is no CPU fallback.")
    # torch first: its bundled HIP runtime must be the one in the process (streams and device
    # pointers are shared with torch); loading ours first left torch.cuda unavailable
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

GCRL_OK, GCRL_ERR_ARG, GCRL_ERR_HIP, GCRL_ERR_NOT_ENOUGH, GCRL_ERR_STATE = 0, -1, -2, -3, -4


def last_error() -> str:
    return lib.gcrl_last_error().decode("utf-8", "replace")


def check(rc: int) -> int:
    """Raise the Python exception that mirrors the reference's behaviour for a status code."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == GCRL_ERR_NOT_ENOUGH:
        raise NotEnoughSamples(msg)
    if rc == GCRL_ERR_ARG:
        raise ValueError(msg)
    raise GcrlError(f"[gcrl status {rc}] {msg}")


def check_ptr(p, what: str):
    if not p:
        raise GcrlError(f"{what} failed: {last_error()}")
    return p


def stream_handle(torch_stream=None) -> int:
    """hipStream_t of torch's current stream as the ABI wants it (0 -> legacy sentinel)."""
    import torch
    if torch_stream is None:
        try:   # raw handle straight from the C extension: torch.cuda.current_stream() costs ~10 us per call
            h = int(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
            return h if h != 0 else STREAM_LEGACY
        except AttributeError:
            torch_stream = torch.cuda.current_stream()
    h = int(torch_stream.cuda_stream)
    return h if h != 0 else STREAM_LEGACY
