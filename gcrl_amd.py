"""Import shim: `import gcrl_amd` loads the package in ./goal-conditioned-rl-framework_amd/
(a hyphenated directory name cannot appear in an import statement)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "goal-conditioned-rl-framework_amd")
_spec = _u.spec_from_file_location("gcrl_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["gcrl_amd"] = _mod
_spec.loader.exec_module(_mod)
