set -x
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
timeout 300 python bench.py > gpurun_out/bench_r02_a.json 2> gpurun_out/bench_r02_a.err; tail -c 1500 gpurun_out/bench_r02_a.json; tail -3 gpurun_out/bench_r02_a.err
timeout 300 python bench.py --force-dist --cpu-clades 0 > gpurun_out/bench_r02_fd.json 2> gpurun_out/bench_r02_fd.err; tail -c 1200 gpurun_out/bench_r02_fd.json; tail -5 gpurun_out/bench_r02_fd.err
tools/exp/valu_rates2.sh > /dev/null 2>&1
