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


CLI_EMU = os.path.join(HERE, "skani-hip-emu")


def build_cli_emu(force=False):
    """TEST-ONLY: the product's CLI sources (host/main.cpp, node.cpp, ...) linked against the kernel simulator build instead of libskani_hip.so, so that the drivers --
    `triangle --gpus N` with its launcher and shared-memory collectives above all -- run in a container without a GPU.  rccl_stub.cpp stands in for the two RCCL entry
    points the simulator build lacks (they fail: the ranks then agree on host collectives, the path `--one-device` takes on a one-GPU box)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_emu", os.path.join(ROOT, "tests", "emu", "build_emu.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    emu = m.build()
    srcs = [os.path.join(HOST, f) for f in ("main.cpp", "fastx.cpp", "writers.cpp", "formats.cpp", "node.cpp")] + [os.path.join(HERE, "rccl_stub.cpp")]
    deps = srcs + [os.path.join(HOST, "host.hpp"), emu]
    if force or not os.path.exists(CLI_EMU) or any(os.path.getmtime(d) > os.path.getmtime(CLI_EMU) for d in deps):
        r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", "-o", CLI_EMU] + srcs +
                           ["-L", os.path.dirname(emu), "-lskani_emu", "-lz", "-pthread", "-Wl,-rpath," + os.path.dirname(emu)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed:\n" + r.stderr[-6000:])
    return CLI_EMU


if __name__ == "__main__":
    print(build(force=True))
