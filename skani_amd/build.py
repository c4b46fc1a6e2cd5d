"""Builds libskani_hip.so (hipcc, gfx950) in-tree.  `python -m skani_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["alloc.hip", "scan.hip", "sort.hip", "pack_seed.hip", "sketch_build.hip", "screen.hip", "screen_keys.hip", "chain.hip", "dist.hip", "rccl_transport.hip", "capi.hip"]
LIB = os.path.join(HERE, "libskani_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-ffp-contract=off"]


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "skani_hip.h")]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_hip(force=False, verbose=False):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = _deps()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s); obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + deps):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout[-4000:], r.stderr[-8000:]))
        return r.stderr
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn:
                print(warn[-2000:])
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


HOST = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HERE, "libskani_host.so")
HOST_BIN = os.path.join(HERE, "bin", "skani-hip")


def build_host(force=False):
    """C++ host side (FASTA ingest, writers, triangle/dist drivers): g++ only, links the C ABI library."""
    srcs = [os.path.join(HOST, f) for f in ("fastx.cpp", "writers.cpp", "formats.cpp", "node.cpp")]
    deps = srcs + [os.path.join(HOST, "host.hpp"), os.path.join(HOST, "capi_db.cpp"), os.path.join(HOST, "main.cpp"), LIB]
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed: %s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off"]
    if force or _stale(HOST_LIB, deps):
        run(common + ["-shared", "-o", HOST_LIB] + srcs + [os.path.join(HOST, "capi_db.cpp"), "-lz", "-pthread"])   # (capi_db.cpp: what skani_amd/formats.py binds)
    if force or _stale(HOST_BIN, deps):
        run(common + ["-o", HOST_BIN, os.path.join(HOST, "main.cpp")] + srcs + ["-L", HERE, "-lskani_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN/.."])
    return HOST_LIB, HOST_BIN


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
