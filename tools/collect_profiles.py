#!/usr/bin/env python3
"""Copies what a `tools/gpu_job.sh TAG ...` run left in gpurun_out/ into profiles/ under the round's names and stamps the three derived files with the commit the job ran on.
usage: collect_profiles.py <tag> <round, e.g. r06> <commit the job ran on>
(the GPU box has no .git: seed_traffic.json, chain_traffic.json and stage_profile.json come back with an empty commit field)"""
import json
import os
import shutil
import sys

tag, rnd, commit = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = [("bench_%s.json", "%s_bench.json"), ("trace_%s.md", "%s_kernel_trace.md"), ("pmc_%s.md", "%s_pmc.md"), ("pmc_%s.json", "%s_pmc.json"), ("timeline_%s.txt", "%s_timeline.txt"),
         ("gaps_%s.txt", "%s_gaps.txt"), ("%s_variants.json", "%s_variants.json"), ("%s_config4_n1.json", "%s_bench_config4_n1.json"), ("%s_config4_dist_w1.json", "%s_bench_config4_dist_world1.json"),
         ("%s_search_65k.json", "%s_search_config5.json"), ("%s_search_113k.json", "%s_search_113k.json"), ("%s_fd.json", "%s_bench_force_dist.json"), ("%s_8ranks.json", "%s_bench_8ranks_one_device.json"),
         ("%s_wide_sweep0.json", "%s_bench_forced_wide.json"), ("%s_wide_sweep1.json", "%s_bench_forced_wide_sweep_dp.json"), ("%s_fuzz.txt", "%s_fuzz.txt"), ("%s_predict_inputs.json", "%s_predict_inputs.json")]
for src, dst in pairs:
    s = os.path.join(G, src % tag)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst % rnd)); print("copied", src % tag, "->", dst % rnd)
for src, dst in (("seed_traffic_%s.json", "seed_traffic.json"), ("chain_traffic_%s.json", "chain_traffic.json"), ("stage_profile_%s.json", "stage_profile.json")):
    s = os.path.join(G, src % tag)
    if not os.path.exists(s): continue
    d = json.load(open(s)); d["commit"] = "%s (the tree job %s ran on)" % (commit, tag)
    if "source" in d: d["source"] = d["source"].replace("r05_pmc", rnd + "_pmc")
    json.dump(d, open(os.path.join(P, dst), "w"), indent=1); print("stamped", dst)
log = os.path.join(G, "gpu_tests_%s.log" % tag)
if os.path.exists(log):
    keep = [l for l in open(log, errors="replace").read().splitlines() if not l.startswith(("Hostname", "Librccl", "RCCL", "HIP", "ROCm")) and "amdgpu.ids" not in l][-12:]
    open(os.path.join(P, "%s_gpu_tests.txt" % rnd), "w").write("python -m pytest tests -m gpu -x -q --durations=6   (job %s, commit %s)\n" % (tag, commit) + "\n".join(keep) + "\n")
