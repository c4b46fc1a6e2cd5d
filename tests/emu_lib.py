"""Loads the TEST-ONLY kernel simulator build (tests/emu) behind the same ctypes prototypes as the product library."""
import importlib.util
import os

from skani_amd import _binding

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = {}


def emu_lib(variant=None):
    """variant: None, or "bigpad" (tests/emu/build_emu.py VARIANTS: contigs ~2^30 padded coordinates apart)"""
    if variant not in _LIB:
        spec = importlib.util.spec_from_file_location("build_emu", os.path.join(_HERE, "emu", "build_emu.py"))
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
        _LIB[variant] = _binding.load(m.build(variant=variant))
    return _LIB[variant]
