"""Importable alias for the package directory ``dgvit-depth-goal-guided-vision-transformer-_amd/``
(its mandated name is not a valid Python identifier).  ``import dgvit_amd`` loads that directory as the
package ``dgvit_amd``; submodules resolve inside it (``dgvit_amd.got_sac_network`` ...)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dgvit-depth-goal-guided-vision-transformer-_amd")
_spec = importlib.util.spec_from_file_location("dgvit_amd", os.path.join(_DIR, "__init__.py"),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dgvit_amd"] = _mod
_spec.loader.exec_module(_mod)
