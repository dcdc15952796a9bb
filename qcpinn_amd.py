"""Importable alias: ``import qcpinn_amd`` == the package directory
``qcpinn-convection-diffusion-qiskit_amd/`` (whose name is not a Python identifier)."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("qcpinn-convection-diffusion-qiskit_amd")
sys.modules[__name__] = _pkg
