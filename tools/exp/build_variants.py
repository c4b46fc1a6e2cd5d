"""Builds libskani_hip variants that differ in one -D of one source: python tools/exp/build_variants.py <source.hip> <MACRO> <value>... -> tools/exp/variants/libskani_hip_<MACRO>_<value>.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from skani_amd import build as B
B.build_hip()
objdir = os.path.join(B.CSRC, "build"); out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants")
os.makedirs(out, exist_ok=True)
src, macro = sys.argv[1], sys.argv[2]
for v in sys.argv[3:]:
    obj = os.path.join(out, "%s_%s_%s.o" % (src.replace(".hip", ""), macro, v))
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-D%s=%s" % (macro, v), "-c", os.path.join(B.CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(objdir, s.replace(".hip", ".o")) for s in B.SOURCES]
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "libskani_hip_%s_%s.so" % (macro, v))] + objs + ["-ldl"])
    print("built", macro, v)
