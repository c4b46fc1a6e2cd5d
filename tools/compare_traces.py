#!/usr/bin/env python3
"""Two kernel-trace summaries (tools/prof.sh: trace_<tag>.json) side by side: per kernel ms per step in each and the ratio; kernels that moved by more than 15 % are marked.
usage: compare_traces.py <a.json> <b.json> <steps in each trace> [name of a] [name of b]"""
import json
import sys

a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])); steps = int(sys.argv[3])
na, nb = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("a", "b")
rows = []
for k in sorted(set(a) | set(b)):
    if "skh::" not in k: continue
    x, y = a.get(k, {}).get("total_ms", 0.0) / steps, b.get(k, {}).get("total_ms", 0.0) / steps
    if max(x, y) < 0.05: continue
    rows.append((max(x, y), k.replace("void ", "").replace("skh::", "")[:70], x, y))
print("| kernel | %s ms/step | %s ms/step | ratio | |\n|---|---|---|---|---|" % (na, nb))
for _, k, x, y in sorted(rows, reverse=True):
    r = y / x if x else float("inf")
    print("| %s | %.3f | %.3f | %.2f | %s |" % (k, x, y, r, "moved" if (r > 1.15 or r < 1 / 1.15) else ""))
print("| all skh kernels | %.3f | %.3f | %.2f | |" % (sum(r[2] for r in rows), sum(r[3] for r in rows), sum(r[3] for r in rows) / max(sum(r[2] for r in rows), 1e-9)))
