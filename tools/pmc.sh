#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc.sh <tag>  -> gpurun_out/pmc_<tag>.md / .json
# Four separate counter passes (the rocprofv3 PMC rules of MI355X_MICROARCH.md: no mixing with runtime/sys traces).
tag=$1
R=$PWD
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_$tag
run() { name=$1; shift; rm -rf /tmp/pmc_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o pmc -- python $R/bench.py --steps 1 --warmup 0 --cpu-clades 0 --no-e2e --strong-collection 0 --no-units --no-variants > /tmp/pmc_$name.log 2>&1; echo $name rc=$?; python $R/tools/rocpd_pmc_summary.py /tmp/pmc_$name/pmc_results.db $R/gpurun_out/pmc_$tag/$name.json > /dev/null; }
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run SQ2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS
python $R/tools/pmc_table.py $R/gpurun_out/pmc_$tag $R/gpurun_out/pmc_$tag.md $R/gpurun_out/pmc_$tag.json "rocprofv3 --pmc <set> --kernel-trace -- python bench.py --steps 1 --warmup 0 --cpu-clades 0 --no-e2e --strong-collection 0 (N=1000 x 5 Mbp, one step, 4 separate counter passes; $tag kernels)"
