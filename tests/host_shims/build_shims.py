"""Builds tests/host_shims/libskani_host_test.so: the C++ host sources (FASTA reader, writers, on-disk formats) behind the test-only extern "C" hooks of
capi_test.cpp, so the CPU suite can call them through ctypes.  TEST-ONLY: the product's host library is skani_amd/libskani_host.so (skani_amd/build.py)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HOST = os.path.join(ROOT, "skani_amd", "host")
LIB = os.path.join(HERE, "libskani_host_test.so")


def build(force=False):
    srcs = [os.path.join(HOST, f) for f in ("fastx.cpp", "writers.cpp", "formats.cpp")] + [os.path.join(HERE, "capi_test.cpp")]
    deps = srcs + [os.path.join(HOST, "host.hpp")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off", "-shared", "-o", LIB] + srcs + ["-lz", "-pthread"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed:\n" + r.stderr[-6000:])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
