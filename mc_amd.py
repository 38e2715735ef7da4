"""Import alias for the package directory `marching-cube-for-implicit-surfaces_amd/`
(its name, fixed by the project layout, is not a Python identifier)."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "marching-cube-for-implicit-surfaces_amd"
_spec = importlib.util.spec_from_file_location("mc_amd_pkg", _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mc_amd_pkg"] = _mod
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
