set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a_tests.log; tail -3 gpurun_out/r3a_tests.log
grep -q "tests rc=0" gpurun_out/r3a_tests.log || exit 1
timeout -k 10 200 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r3a_bench.json
timeout -k 10 300 python tools/bench_configs.py --stats config4 config4_uniform_starts config5 config5_uniform_starts --iters 10 > gpurun_out/r3a_configs.jsonl 2> gpurun_out/r3a_configs.err; echo "configs rc=$?"; cut -c1-600 gpurun_out/r3a_configs.jsonl
timeout -k 10 200 python tools/experiments/small_edge_batches.py > gpurun_out/r3a_small_edges.txt 2>&1; echo "small rc=$?"; cat gpurun_out/r3a_small_edges.txt
PROFILE_PROG=tools/bench_configs.py timeout -k 10 400 bash tools/profile.sh r3a_config4 config4 --iters 5 > gpurun_out/r3a_prof_c4.log 2>&1; echo "c4 prof rc=$?"
