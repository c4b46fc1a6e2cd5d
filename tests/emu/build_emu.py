"""Builds tests/emu/libskani_emu.so: the kernel sources of skani_amd/csrc compiled by g++ against the
lockstep simulator (emu.h).  TEST-ONLY -- see emu.h."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "skani_amd", "csrc")
SOURCES = ["scan.hip", "pack_seed.hip", "sketch_build.hip", "screen.hip", "screen_keys.hip", "chain.hip", "dist.hip", "capi.hip"]   # not part of the simulator build: rccl_transport.hip (RCCL, GPU only), sort.hip (rocPRIM; emu_sort.cpp stands in), alloc.hip
LIB = os.path.join(HERE, "libskani_emu.so")
FLAGS = ["-O2", "-g", "-ffp-contract=off", "-std=c++17", "-fPIC", "-DSKANI_EMU", "-I", HERE, "-I", CSRC, "-pthread", "-Wall", "-Wno-unknown-pragmas",
         "-Wno-attributes", "-fno-strict-aliasing"]


# variant "bigpad": contigs 2^30 - 5450 padded coordinates apart instead of 8192 (common.h SKH_CTG_PAD): two contigs are a wide genome, the fifth contig starts beyond
# 2^32 -- the 64-bit paths (high words of the ring DP's reference coordinates above all) meet coordinates that really need them, on inputs of a few kilobases
VARIANTS = {None: [], "bigpad": ["-DSKH_CTG_PAD=1073736374"]}


def build(force=False, variant=None):
    global LIB
    extra = VARIANTS[variant]
    objdir = os.path.join(HERE, "build" + ("_" + variant if variant else "")); os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(HERE, "libskani_emu%s.so" % ("_" + variant if variant else ""))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "emu.h"), os.path.join(HERE, "emu_dev.h"),
            os.path.join(ROOT, "include", "skani_hip.h")]
    jobs = []
    srcs = [(os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))) for s in SOURCES]
    srcs.append((os.path.join(HERE, "emu.cpp"), os.path.join(objdir, "emu.o")))
    srcs.append((os.path.join(HERE, "emu_sort.cpp"), os.path.join(objdir, "emu_sort.o")))
    for src, obj in srcs:
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in [src] + deps):
            jobs.append(["g++"] + FLAGS + extra + ["-x", "c++", "-c", src, "-o", obj])
    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("g++ failed: %s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    objs = [o for _, o in srcs]
    if force or jobs or not os.path.exists(lib):
        run(["g++", "-shared", "-fPIC", "-pthread", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force=True))
