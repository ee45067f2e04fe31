set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3j_tests.log; tail -3 gpurun_out/r3j_tests.log
grep -q "tests rc=0" gpurun_out/r3j_tests.log || exit 1
bash tools/ab_bench.sh --no-cpu-baseline --no-two-streams --no-shard-probe --steps 200 --warmup 20 > gpurun_out/r3j_ab_bench.txt 2>&1; cat gpurun_out/r3j_ab_bench.txt
bash tools/ab_robots.sh > gpurun_out/r3j_ab_robots.txt 2>&1; cat gpurun_out/r3j_ab_robots.txt
bash tools/ab_configs.sh config3 config4 config5 --iters 10 > gpurun_out/r3j_ab_configs.txt 2>&1; cat gpurun_out/r3j_ab_configs.txt
