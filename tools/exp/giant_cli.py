"""Two 2.3 Gbp genomes (three contigs each, 1 % apart) as FASTA files on a RAM disk through `skani-hip dist`: wall time and the driver's phase clock."""
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, '.')
from tests.helpers import big_random_genome, big_mutate
import skani_amd as sk
from skani_amd.build import build_host
_, exe = build_host()
total = 2_300_000_000
a = big_random_genome(total, 71); b = big_mutate(a, 0.01, 72)
cuts = [0, 900_000_000, 1_700_000_000, total]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
names = []
for tag, x in (("g0", a), ("g1", b)):
    p = os.path.join(d, tag + ".fa"); names.append(p)
    with open(p, "wb") as f:
        for i in range(3):
            f.write(b">chr%d\n" % i); f.write(memoryview(x[cuts[i]:cuts[i + 1]])); f.write(b"\n")
del a, b
env = dict(os.environ, SKH_TIMING="1", SKANI_HIP_DATA=os.path.join(os.path.dirname(sk.library_path()), "data"))
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "dist", "-t", "8", "-q", names[1], "-r", names[0]], capture_output=True, text=True, env=env)
    print("run %d: %.2f s wall, rc %d" % (rep, time.perf_counter() - t0, r.returncode))
    print(r.stdout.strip()[-400:]); print(r.stderr.strip()[-600:])
for p in names: os.remove(p)
os.rmdir(d)
