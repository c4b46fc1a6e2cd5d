#!/usr/bin/env python3
"""Static VALU instruction mix of one kernel, priced with the issue rates measured on gfx950 (profiles/r02_valu_rates.md).

usage: isa_mix.py <source.hip> <kernel-name-substring> [--dynamic-valu N] [--json out.json]
The source is compiled to gfx950 assembly (hipcc -S --cuda-device-only, the library's flags); every VALU instruction of the kernel body is put into a
rate class: full rate 2.17 cycles per wave-instruction per SIMD, half rate 4.27, carry / 64-bit-compare / 32x32->64 multiply-add 4.5.  The result is the
number of VALU issue cycles ONE wave needs: the static sum (straight-line kernels), or -- with --dynamic-valu N, the measured SQ_INSTS_VALU per wave --
N times the static mix's mean cost.  bench.py divides (waves x issue cycles per wave) by (SIMDs x clock x kernel time): roofline.valu_frac."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

FULL = {"v_mov_b32", "v_not_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_add_f32", "v_mul_f32", "v_sub_f32",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_nop"}
SLOW = {"v_mad_u64_u32", "v_mad_i64_i32", "v_cmp_gt_u64", "v_cmp_lt_u64", "v_cmp_ge_u64", "v_cmp_le_u64", "v_cmp_eq_u64", "v_cmp_ne_u64", "v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32"}
CYC = {"full": 2.17, "half": 4.27, "slow": 4.5}


def classify(m):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m)
    if m.endswith(("_sdwa", "_dpp")):
        return "half"
    if base in SLOW:
        return "slow"
    if base in FULL:
        return "full" if not (m.endswith("_e64") and base in ("v_xor_b32", "v_and_b32", "v_or_b32")) else "full"
    return "half"


def main():
    src, name = os.path.abspath(sys.argv[1]), sys.argv[2]
    dyn = float(sys.argv[sys.argv.index("--dynamic-valu") + 1]) if "--dynamic-valu" in sys.argv else None
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", asm, src],
                       check=True, capture_output=True, cwd=td)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    counts, per_mnemonic, others = {"full": 0, "half": 0, "slow": 0}, {}, {"salu": 0, "vmem": 0, "lds": 0, "smem": 0}
    for l in lines[start + 1:end]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        m = t[0]
        if m.startswith("v_"):
            c = classify(m); counts[c] += 1; per_mnemonic[m] = per_mnemonic.get(m, 0) + 1
        elif m.startswith(("global_", "buffer_", "flat_", "scratch_")):
            others["vmem"] += 1
        elif m.startswith("ds_"):
            others["lds"] += 1
        elif m.startswith("s_load") or m.startswith("s_buffer_load"):
            others["smem"] += 1
        elif m.startswith("s_"):
            others["salu"] += 1
    n_static = sum(counts.values())
    cyc_static = sum(counts[c] * CYC[c] for c in counts)
    res = {"kernel": name, "source": os.path.relpath(src), "source_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           "static_valu": n_static, "static_by_class": counts, "static_issue_cycles": cyc_static, "mean_cycles_per_valu": cyc_static / max(n_static, 1),
           "cycles_per_class": CYC, "other_instructions": others,
           "top_mnemonics": dict(sorted(per_mnemonic.items(), key=lambda kv: -kv[1])[:16])}
    if dyn is not None:
        res["dynamic_valu_per_wave"] = dyn
        res["issue_cycles_per_wave"] = dyn * res["mean_cycles_per_valu"]
    else:
        res["issue_cycles_per_wave"] = cyc_static
    print(json.dumps(res, indent=1))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
