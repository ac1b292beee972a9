"""Import alias: `import opus_pllm_amd` -> the package in ./opus-pllm_amd/ (hyphenated directory)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "opus-pllm_amd")
_spec = _ilu.spec_from_file_location("opus_pllm_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["opus_pllm_amd"] = _mod
_spec.loader.exec_module(_mod)
