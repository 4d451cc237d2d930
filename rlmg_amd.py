"""Import alias: the product package lives in `reinforcement-learning-in-music-generation_amd/`
(a directory name Python cannot import directly); `import rlmg_amd` loads that directory as the
package `rlmg_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reinforcement-learning-in-music-generation_amd")
_spec = importlib.util.spec_from_file_location(
    "rlmg_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rlmg_amd"] = _mod
_spec.loader.exec_module(_mod)
