"""MI355X-native HER replay + actor-critic update engine (host side).

Layout:
  csrc/   HIP kernels + the C ABI (include/gcrl.h) -> libgcrl_hip.so
  _ffi.py ctypes binding of that library (no CPU fallback)
  src/    the reference's hot-path interface, same module and class names as its src/
          (buffer.HERBuffer, agent.DDPG/TD3Agent/SACAgent/TQCAgent, model.*, utils.*)

The directory name carries a hyphen, so import it through the repo-root shim: `import gcrl_amd`.
"""
from . import _ffi  # noqa: F401  (loads the shared library; ImportError if it was not built)
from .src import agent, buffer, model, utils  # noqa: F401
from .src.agent import DDPG, SACAgent, TD3Agent, TQCAgent  # noqa: F401
from .src.buffer import HERBuffer, MTStream, PERBuffer, ReplayBuffer  # noqa: F401

__all__ = ["DDPG", "TD3Agent", "SACAgent", "TQCAgent", "HERBuffer", "ReplayBuffer", "PERBuffer", "MTStream", "agent", "buffer",
           "model", "utils"]
