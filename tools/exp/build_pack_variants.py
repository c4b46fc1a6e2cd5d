"""Builds libskani_hip variants that differ in pack_seed.hip's PACK_ROUNDS_N (tools/exp/variants/libskani_hip_r<N>.so) for an A/B run on the GPU box."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from skani_amd import build as B
B.build_hip()
objdir = os.path.join(B.CSRC, "build"); out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants")
for n in sys.argv[1:]:
    obj = os.path.join(out, "pack_seed_r%s.o" % n)
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DPACK_ROUNDS_N=%s" % n, "-c", os.path.join(B.CSRC, "pack_seed.hip"), "-o", obj])
    objs = [obj if s == "pack_seed.hip" else os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libskani_hip_r%s.so" % n)] + objs + ["-ldl"])
    print("built", n)
