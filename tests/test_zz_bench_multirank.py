"""bench.py's multi-rank branch on ONE MI355X (`--one-device`): the driver's 8-GPU lease is then not the first time those lines run.
All ranks share cuda:0 and talk through host collectives over gloo (RCCL refuses two ranks on one GPU); everything else -- the launch through
torch.distributed.run, the control plane, the collection, the deferred tables, skh_triangle_distributed, the reduction of the times, the
per-rank block of the JSON line -- is what `bench.py --gpus 8` runs with one GPU per rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _bench(n_ranks, extra, timeout):
    cmd = [sys.executable]
    if n_ranks > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # the contract: ONE JSON line on stdout, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_eight_ranks_one_device_config4():
    """The command the driver runs on an 8-GPU node, with the eight ranks on one device: BASELINE config 4 (10,000 genomes, 1,250 per rank, shuffled); the oracle beside
    it on two sampled clades (cpu_baseline scaled to the collection, delta_vs_oracle over their pairs), the end-to-end leg through `skani-hip triangle --gpus 8 --one-device`,
    and the `strong` block (at eight ranks the weak default IS the fixed collection: the same figures)."""
    out = _bench(8, ["--one-device", "--steps", "2", "--warmup", "1", "--cpu-sample-clades", "2"], 2400)
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    cfg = out["config"]
    assert cfg["genomes"] == 10000 and cfg["genomes_per_gpu"] == 1250 and cfg["pairs"] == 10000 * 9999 // 2 and cfg["chained_pairs"] == 95000 and cfg["kept_pairs"] == 95000
    assert cfg["one_device"] is True and cfg["transport"] == "torch" and cfg["order"] == "shuffled"
    pr = out["per_rank"]
    assert all(len(v) == 8 for v in pr.values()) and sum(pr["chained_pairs"]) == 95000
    mean = 95000 / 8
    assert max(abs(x - mean) for x in pr["chained_pairs"]) <= 0.03 * mean, pr["chained_pairs"]
    assert sum(pr["sketches_received"]) <= 10000 and all(b > 0 for b in pr["bytes_received"]) and all(b > 5.5e9 for b in pr["bases"])
    assert abs(out["value"] - cfg["pairs"] / (out["ms_per_step"] * 1e-3)) <= 1e-6 * out["value"]
    assert out["chained_pairs_per_s_per_gpu"] > 0 and out["bases_per_s_per_gpu"] > 0
    assert cfg["result_rows"] == "gathered on rank 0" and pr["rows_returned"][0] == 95000 and sum(pr["rows_returned"][1:]) == 95000 - pr["chained_pairs"][0]
    cb = out["cpu_baseline"]; d = cb["delta_vs_oracle"]
    assert cb["kind"] == "port" and cb["seconds_scaled_to_collection"]["total"] > 0 and cb["value"] > 0 and "10000-genome collection" in cb["sample"]
    assert d["pairs_compared"] == 2 * 190 and d["same_pair_set"] and d["max_abs_d_ani"] <= 1e-4 and d["int_fields_equal"]
    st = out["strong"]
    assert st["collection"] == 10000 and st["genomes_per_gpu"] == 1250 and st["ms_per_step"] == out["ms_per_step"] and st["chained_pairs"] == 95000
    e = out["e2e"]
    assert "error" not in e, e
    assert e["gpus"] == 8 and e["genomes"] == 40 and e["matrix_rows"] == 40 and "--gpus 8 --one-device" in e["command"] and e["phases_s"]["triangle_s"] > 0


@pytest.mark.gpu
def test_bench_strong_mode_small_collection():
    """`--collection`: the same shuffled collection at every N.  80 genomes of 1 Mbp on one rank (with the oracle on sampled clades: delta_vs_oracle) and
    on two ranks sharing the device: the same pairs chained, the per-GPU rates present, "scaling": "strong"."""
    common = ["--collection", "80", "--mean-len", "1000000", "--steps", "2", "--warmup", "1", "--no-e2e"]
    one = _bench(1, common + ["--cpu-sample-clades", "2"], 900)
    two = _bench(2, common + ["--one-device"], 900)
    for out, n in ((one, 1), (two, 2)):
        assert out["scaling"] == "strong" and out["n_gpus"] == n and out["config"]["genomes"] == 80 and out["config"]["genomes_per_gpu"] == 80 // n
        assert out["config"]["order"] == "shuffled" and out["config"]["chained_pairs"] == 4 * 190
        assert out["chained_pairs_per_s_per_gpu"] > 0 and out["bases_per_s_per_gpu"] > 0
    d = one["cpu_baseline"]["delta_vs_oracle"]
    assert d["pairs_compared"] == 2 * 190 and d["same_pair_set"] and d["max_abs_d_ani"] <= 1e-4 and d["max_abs_d_af_ref"] <= 1e-4 and d["int_fields_equal"]
    assert one["cpu_baseline"]["seconds_scaled_to_collection"] is not None and one["cpu_baseline"]["threads"] == one["cpu_baseline"]["cores"]
    assert len(two["per_rank"]["chained_pairs"]) == 2 and sum(two["per_rank"]["chained_pairs"]) == 4 * 190
    d2 = two["cpu_baseline"]["delta_vs_oracle"]                            # several ranks: rank 0 makes the sampled clades for the oracle itself (here: all four)
    assert d2["pairs_compared"] == 4 * 190 and d2["same_pair_set"] and d2["max_abs_d_ani"] <= 1e-4 and d2["int_fields_equal"] and "strong" not in two


@pytest.mark.gpu
def test_bench_weak_line_carries_a_strong_block():
    """The default (weak) mode measures the fixed collection at the same N as well: `strong` = that run's step time and per-GPU rates.  Small shapes: 2 ranks x 40 genomes of
    1 Mbp, a fixed collection of 120."""
    out = _bench(2, ["--one-device", "--genomes-per-gpu", "40", "--mean-len", "1000000", "--steps", "2", "--warmup", "1", "--cpu-sample-clades", "2", "--no-e2e",
                     "--strong-collection", "120", "--strong-steps", "2"], 900)
    assert out["scaling"] == "weak" and out["config"]["genomes"] == 80 and out["config"]["chained_pairs"] == 4 * 190
    st = out["strong"]
    assert st["collection"] == 120 and st["genomes_per_gpu"] == 60 and st["chained_pairs"] == 6 * 190 and st["steps"] == 2 and st["ms_per_step"] > 0
    assert st["chained_pairs_per_s_per_gpu"] > 0 and len(st["per_rank"]["chained_pairs"]) == 2 and sum(st["per_rank"]["chained_pairs"]) == 6 * 190
    assert out["cpu_baseline"]["delta_vs_oracle"]["same_pair_set"]


@pytest.mark.gpu
def test_bench_force_dist_rccl_world_one():
    """`--force-dist`: the library's RCCL communicator (self-test included) with a world of one, through bench.py's multi-rank branch."""
    out = _bench(1, ["--force-dist", "--genomes-per-gpu", "40", "--mean-len", "1000000", "--steps", "2", "--warmup", "1", "--cpu-clades", "0"], 900)
    assert out["config"]["transport"] == "rccl" and out["config"]["chained_pairs"] == 2 * 190 and len(out["per_rank"]["chained_pairs"]) == 1
