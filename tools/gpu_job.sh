#!/bin/bash
# usage (on the GPU box, from the repo root; through gpurun): tools/gpu_job.sh <tag> <section> [<section> ...]
# One parameterised job instead of a script per GPU run.  Everything lands in gpurun_out/ under names that carry <tag>.
# sections: suite | suite:<pytest -k expr> | smoke | bench | benchquick | trace | gaps | pmc | traffic | forcedist | config4 | config4dist | touch | ranks8 | search | search113k | presets | fuzz[:rounds] | mergejoin | wide | predict | cli | sortab | skeysavg | alloctrace | config4trace | hosttrace | screenab | overlap | probe | stageprofile | realdense | densetraces | variants (after bench) | benchmini (first: stops the job when bench.py fails)
mkdir -p gpurun_out
tag=$1; shift
short() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
cb = d.get("cpu_baseline") or {}
print(sys.argv[1], "ms/step", round(d["ms_per_step"], 3), {k: round(v, 2) for k, v in d["phase_ms_per_step"].items()}, "value", round(d["value"]),
      {k: d["config"].get(k) for k in ("chained_pairs", "hits", "library_live_gb") if d["config"].get(k) is not None},
      "cpu", round(cb["value"]) if cb.get("value") else None, cb.get("delta_vs_oracle"), "strong", (d.get("strong") or {}).get("ms_per_step"),
      "steps", d.get("step_wall_ms_rank0"))
PY
}
for sec in "$@"; do
  echo "== $sec"; date
  case $sec in
    suite) timeout 2700 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/gpu_tests_$tag.log 2>&1; grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" gpurun_out/gpu_tests_$tag.log | tail -12 ;;
    suite:*) timeout 2700 python -m pytest tests -m gpu -x -q -k "${sec#suite:}" --durations=6 > gpurun_out/gpu_tests_$tag.log 2>&1; grep -v "^Hostname\|^Librccl\|^RCCL\|^HIP\|^ROCm" gpurun_out/gpu_tests_$tag.log | tail -25 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    bench) timeout 1200 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err; short gpurun_out/bench_$tag.json
           python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); r=d['roofline']; print('roofline', r['frac'], r.get('frac_rocprof'), r.get('valu_frac'), r['ms_per_launch'], r['traffic'], (d.get('roofline_chain') or {}).get('traffic'), 'e2e', (d.get('e2e') or {}).get('wall_s'), (d.get('e2e') or {}).get('error'))" ;;
    benchquick) timeout 600 python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --no-variants --steps 20 > gpurun_out/benchq_$tag.json 2> gpurun_out/benchq_$tag.err || tail -5 gpurun_out/benchq_$tag.err; short gpurun_out/benchq_$tag.json ;;
    trace) tools/prof.sh $tag --no-e2e --strong-collection 0 > /dev/null 2>&1; head -40 gpurun_out/trace_$tag.txt | cut -c1-66,70-110 ;;
    gaps) db=$(find /tmp/prof_$tag -name "*.db" | head -1); python tools/rocpd_gaps.py $db gpurun_out/gaps_$tag.txt | head -3; python tools/rocpd_timeline.py $db gpurun_out/timeline_$tag.txt > /dev/null ;;
    pmc) tools/pmc.sh $tag > gpurun_out/pmc_$tag.log 2>&1; tail -2 gpurun_out/pmc_$tag.log | cut -c1-160 ;;
    traffic) timeout 300 python bench.py --no-e2e --cpu-clades 0 --strong-collection 0 > gpurun_out/bench_pre_$tag.json 2>/dev/null
             python tools/make_seed_traffic.py gpurun_out/pmc_$tag.json "profiles/r06_pmc.json (tools/pmc.sh, $tag)" gpurun_out/trace_$tag.md | tail -3
             python tools/make_chain_traffic.py gpurun_out/pmc_$tag.json gpurun_out/bench_pre_$tag.json "profiles/r06_pmc.json (tools/pmc.sh, $tag)" | tail -2
             cp profiles/seed_traffic.json gpurun_out/seed_traffic_$tag.json; cp profiles/chain_traffic.json gpurun_out/chain_traffic_$tag.json ;;
    forcedist) timeout 300 python bench.py --force-dist --cpu-clades 0 --no-e2e --steps 20 > gpurun_out/${tag}_fd.json 2> gpurun_out/${tag}_fd.err; short gpurun_out/${tag}_fd.json ;;
    config4) timeout 900 python bench.py --collection 10000 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_n1.json 2> gpurun_out/${tag}_config4_n1.err || tail -5 gpurun_out/${tag}_config4_n1.err; short gpurun_out/${tag}_config4_n1.json ;;
    config4dist) SKH_TUNE_DIST_KEY_RANGE_W1=1 timeout 900 python bench.py --force-dist --collection 10000 --no-e2e --cpu-clades 0 --steps 8 --warmup 4 > gpurun_out/${tag}_config4_dist_w1.json 2> gpurun_out/${tag}_config4_dist_w1.err || tail -5 gpurun_out/${tag}_config4_dist_w1.err
                 short gpurun_out/${tag}_config4_dist_w1.json
                 SKH_TUNE_DIST_KEY_RANGE_W1=1 SKH_TRACE=1 timeout 600 python bench.py --force-dist --no-e2e --cpu-clades 0 --collection 10000 --steps 1 --warmup 2 2>&1 >/dev/null | grep "skh trace\] dist" | tail -12 ;;
    touch) for t in 0 1; do SKH_TUNE_ALLOC_TOUCH=$t timeout 900 python bench.py --collection 10000 --steps 7 --warmup 1 --cpu-clades 0 --no-e2e > gpurun_out/${tag}_touch$t.json 2> gpurun_out/${tag}_touch$t.err || tail -3 gpurun_out/${tag}_touch$t.err; short gpurun_out/${tag}_touch$t.json; done ;;
    ranks8) timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --one-device --steps 3 --warmup 1 --cpu-sample-clades 5 > gpurun_out/${tag}_8ranks.json 2> gpurun_out/${tag}_8ranks.err || tail -5 gpurun_out/${tag}_8ranks.err
            short gpurun_out/${tag}_8ranks.json ;;
    search) timeout 1200 python bench.py --workload search --db-genomes 65000 --queries 1000 --steps 3 --warmup 1 > gpurun_out/${tag}_search_65k.json 2> gpurun_out/${tag}_search_65k.err && short gpurun_out/${tag}_search_65k.json || tail -5 gpurun_out/${tag}_search_65k.err
            python -c "
import json; d=json.load(open('gpurun_out/${tag}_search_65k.json')); print(json.dumps(d['roofline'])[:600]); print(json.dumps(d['cpu_baseline'])[:1500])" ;;
    search113k) timeout 1200 python bench.py --workload search --db-genomes 113000 --queries 1000 --steps 3 --warmup 1 --cpu-queries 0 > gpurun_out/${tag}_search_113k.json 2> gpurun_out/${tag}_search_113k.err && short gpurun_out/${tag}_search_113k.json || tail -5 gpurun_out/${tag}_search_113k.err ;;
    presets) for c in 30 70 200; do timeout 300 python bench.py --c $c --cpu-clades 0 --no-e2e --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c=$c', round(d['ms_per_step'],2))"; done
             timeout 300 python bench.py --genomes-per-gpu 5000 --cpu-clades 0 --no-e2e --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n5000', round(d['ms_per_step'],2), round(d['value']/1e6,1))"
             timeout 300 python bench.py --clade 1000 --cpu-clades 0 --no-e2e --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense', round(d['ms_per_step'],1), d['config']['chained_pairs'])" ;;
    fuzz*) n=${sec#fuzz}; n=${n#:}; seed=$RANDOM; { echo "tools/fuzz_parity.py ${n:-400} $seed   (rounds, seed)  on $(date -u +%FT%TZ), sources $(cat skani_amd/csrc/*.hip skani_amd/csrc/*.h | sha256sum | cut -c1-16)"; timeout 1200 python tools/fuzz_parity.py ${n:-400} $seed 2>&1 | tail -4; } > gpurun_out/${tag}_fuzz.txt; cat gpurun_out/${tag}_fuzz.txt ;;
    stageprofile) python tools/make_stage_profile.py gpurun_out/trace_$tag.json gpurun_out/pmc_$tag.json 3 "tools/prof.sh + tools/pmc.sh, $tag" | tail -12; cp profiles/stage_profile.json gpurun_out/stage_profile_$tag.json ;;
    realdense) timeout 1500 python bench.py --root-fasta tests/golden/e.coli-W.fasta.gz --clade 1000 --cpu-genomes 24 --steps 2 --warmup 1 --strong-collection 0 --no-e2e > gpurun_out/${tag}_real_dense.json 2> gpurun_out/${tag}_real_dense.err || tail -5 gpurun_out/${tag}_real_dense.err
               short gpurun_out/${tag}_real_dense.json; python -c "
import json; d=json.load(open('gpurun_out/${tag}_real_dense.json')); print(json.dumps(d.get('real_sequence'))[:900]); print(json.dumps(d.get('units')))" ;;
    densetraces) tools/prof.sh ${tag}_dense_syn --no-e2e --clade 1000 > /dev/null 2>&1; tools/prof.sh ${tag}_dense_real --no-e2e --clade 1000 --root-fasta $PWD/tests/golden/e.coli-W.fasta.gz > /dev/null 2>&1
                 python tools/compare_traces.py gpurun_out/trace_${tag}_dense_syn.json gpurun_out/trace_${tag}_dense_real.json 3 "synthetic dense" "E. coli W derivatives" > gpurun_out/${tag}_dense_compare.md; cat gpurun_out/${tag}_dense_compare.md ;;
    variants) python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); v=d.get('variants'); json.dump({'commit_note': 'the variants block of gpurun_out/bench_$tag.json (bench.py default run)', 'headline_ms_per_step': d['ms_per_step'], 'variants': v}, open('gpurun_out/${tag}_variants.json','w'), indent=1)
for x in v or []: print(x.get('variant'), round(x.get('ms_per_step', 0), 2), x.get('chained_pairs'), round(x.get('chained_pairs_per_s', 0)), x.get('error'))" ;;
    wide) for sw in 1 0; do SKH_TUNE_WIDE_SPAN=0 SKH_TUNE_WIDE_SWEEP_DP=$sw timeout 600 python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --steps 10 > gpurun_out/${tag}_wide_sweep$sw.json 2> gpurun_out/${tag}_wide_sweep$sw.err || tail -3 gpurun_out/${tag}_wide_sweep$sw.err; short gpurun_out/${tag}_wide_sweep$sw.json; done
          SKH_TUNE_WIDE_SPAN=0 timeout 900 python bench.py --cpu-clades 5 --no-e2e --strong-collection 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced wide vs oracle', d['cpu_baseline']['delta_vs_oracle'])" ;;
    sortab) for r in 1 0; do SKH_TUNE_SCREEN_SORT_RADIX=$r timeout 600 python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --steps 20 > gpurun_out/${tag}_sort_radix$r.json 2> gpurun_out/${tag}_sort_radix$r.err || tail -3 gpurun_out/${tag}_sort_radix$r.err; short gpurun_out/${tag}_sort_radix$r.json; done ;;
    skeysavg) for a in 350 700 2800; do SKH_TUNE_SKEYS_AVG=$a timeout 600 python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --steps 20 > gpurun_out/${tag}_skeys_avg$a.json 2> gpurun_out/${tag}_skeys_avg$a.err || tail -3 gpurun_out/${tag}_skeys_avg$a.err; short gpurun_out/${tag}_skeys_avg$a.json; done ;;
    alloctrace) SKH_TRACE_ALLOC=1 BENCH_STEP_TIMES=1 timeout 600 python bench.py --no-e2e --cpu-clades 0 --steps 3 --warmup 2 > /dev/null 2> gpurun_out/${tag}_alloctrace.err; grep -n "skh alloc\|^step " gpurun_out/${tag}_alloctrace.err | tail -60 | cut -c1-150 ;;
    config4trace) tools/prof.sh ${tag}c4 --no-e2e --collection 10000 > /dev/null 2>&1; head -24 gpurun_out/trace_${tag}c4.txt | cut -c1-66,70-110 ;;
    benchmini) timeout 600 python bench.py --steps 2 --warmup 1 --cpu-clades 2 --strong-steps 2 > gpurun_out/benchmini_$tag.json 2> gpurun_out/benchmini_$tag.err || { tail -20 gpurun_out/benchmini_$tag.err; echo "bench.py is broken: stopping the job"; exit 1; }; short gpurun_out/benchmini_$tag.json ;;
    hosttrace) SKH_TRACE=2 timeout 300 python bench.py --cpu-clades 0 --no-e2e --strong-collection 0 --steps 3 --warmup 2 2>&1 >/dev/null | grep "skh trace" | tail -75 > gpurun_out/${tag}_hosttrace.txt; tail -75 gpurun_out/${tag}_hosttrace.txt ;;
    predict) timeout 900 python tools/predict_scaling.py > gpurun_out/${tag}_predict_inputs.json 2> gpurun_out/${tag}_predict.err || tail -3 gpurun_out/${tag}_predict.err; cut -c1-400 gpurun_out/${tag}_predict_inputs.json ;;
    mergejoin) timeout 300 tools/exp/merge_join > gpurun_out/mergejoin_$tag.txt 2>&1; cat gpurun_out/mergejoin_$tag.txt ;;
    screenab) for o in clade shuffled; do for v in "0 0" "1 0" "1 1" "1 0" "1 1"; do set -- $v; SKH_TUNE_SCREEN_COUNT_ROWS=$1 SKH_TUNE_SCREEN_COL_ORDER=$2 timeout 600 python bench.py --order $o --cpu-clades 0 --no-e2e --strong-collection 0 --steps 20 > gpurun_out/${tag}_screen_rows$1_order$2_$o.json 2> gpurun_out/${tag}_screen_rows$1_order$2_$o.err || tail -3 gpurun_out/${tag}_screen_rows$1_order$2_$o.err; short gpurun_out/${tag}_screen_rows$1_order$2_$o.json | cut -c1-200; done; done
              for v in "0 0" "1 0" "1 1"; do set -- $v; SKH_TUNE_SCREEN_COUNT_ROWS=$1 SKH_TUNE_SCREEN_COL_ORDER=$2 timeout 900 python bench.py --collection 10000 --steps 4 --warmup 2 --cpu-clades 0 --no-e2e > gpurun_out/${tag}_c4_screen_rows$1_order$2.json 2> gpurun_out/${tag}_c4_screen_rows$1_order$2.err || tail -3 gpurun_out/${tag}_c4_screen_rows$1_order$2.err; short gpurun_out/${tag}_c4_screen_rows$1_order$2.json | cut -c1-200; done ;;
    overlap) timeout 600 python tools/exp/two_halves.py > gpurun_out/${tag}_two_halves.txt 2> gpurun_out/${tag}_two_halves.err || tail -5 gpurun_out/${tag}_two_halves.err; cat gpurun_out/${tag}_two_halves.txt ;;
    probe) { echo "gfx950 agents: $(rocminfo 2>/dev/null | grep -c 'Name: *gfx950')"; rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Name: *gfx" | head -20; echo "-- amd-smi partition"; timeout 60 amd-smi partition 2>&1 | head -40; echo "-- rocm-smi"; timeout 60 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20; python -c "import torch; print('torch devices', torch.cuda.device_count())"; nproc; } > gpurun_out/${tag}_probe.txt 2>&1; cat gpurun_out/${tag}_probe.txt ;;
    cli) timeout 900 python -m pytest tests/test_host_cpp.py -m gpu -x -q 2>&1 | tail -5 ;;
    *) echo "unknown section $sec" ;;
  esac
done
date
