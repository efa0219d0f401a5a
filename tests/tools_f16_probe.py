"""(test helper) the survivor-matching function of tools/f16_trained_probe.py, importable from tests/."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location("f16_trained_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                 "tools", "f16_trained_probe.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
agreement = _mod.agreement
