"""CPU-side checks of the drop-in boundary: the HIP library builds, loads and exports every symbol that
include/skani_hip.h declares (no compute calls without a GPU), and refuses to run without a device."""
import ctypes
import os
import re

import pytest

import skani_amd as sk
from skani_amd import _binding
from skani_amd.build import build_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    return build_hip()


def test_header_symbols_exported(lib_path):
    hdr = open(os.path.join(ROOT, "include", "skani_hip.h")).read()
    declared = set(re.findall(r"\b(skh_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_binding.EXPORTS), declared ^ set(_binding.EXPORTS)
    L = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(L, name), name


def test_no_cpu_fallback(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sk.SkaniHipError):
        sk.Context(0)


def test_product_does_not_use_oracle_or_simulator():
    """The product path must never import, link or call the oracle / the kernel simulator."""
    banned = ("oracle_py", "libskani_oracle", "skani_oracle.h", "libskani_emu", "from oracle", "import oracle", "emu_lib", "ora_")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "skani_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for b in banned:
                    assert b not in txt.replace("ora_chain_stats", ""), (f, b)
