import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import skani_amd as sk
from parity_cases import synthetic_clades, random_genome, mutate
from oracle import oracle_py as ora
ctx = sk.Context(0)
genomes = synthetic_clades(n_clades=2, members=2, length=60000, seed=31) + [[("few", random_genome(9000, 77))]]
names = ["s%02d.fa" % i for i in range(len(genomes))]
sp = sk.SketchParams(marker_c=200)
refs = ctx.sketch_records(genomes, sp, names)
orefs = [ora.sketch_records(g, 125, 15, 200, names[i], 1) for i, g in enumerate(genomes)]
for g in range(len(genomes)):
    mk = np.asarray(refs.export(g)["markers"]); o = orefs[g]; om = np.sort(o.markers())
    print(g, len(mk), len(om), bool(np.array_equal(mk, om)), bool(np.all(np.diff(mk.astype(np.int64)) > 0)))

queries = [genomes[0], [("q", mutate(genomes[2][0][1], 0.03, 5))], [("none", random_genome(40000, 1234))]]
qs = ctx.sketch_records(queries, sp, ["q0", "q1", "q2"])
oqs = [ora.sketch_records(g, 125, 15, 200, "q%d" % i, 1) for i, g in enumerate(queries)]
for rescue in (True, False):
    for rule in (0, 2):
        a, b = ctx.screen(refs, qs, 0.8, rule, rescue)
        exp = sorted((q, int(r)) for q in range(len(oqs)) for r in ora.screen_refs(orefs, oqs[q], 0.8, rule, rescue))
        print(rescue, rule, list(zip(a.tolist(), b.tolist())), exp)
