set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "validate_batch_bit_exact or validate_motion_batch_bit_exact or edge_cases or large_primitive or counted_loop or ill_formed" > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3i_tests.log; tail -3 gpurun_out/r3i_tests.log
grep -q "tests rc=0" gpurun_out/r3i_tests.log || exit 1
bash tools/ab_bench.sh --no-cpu-baseline --no-two-streams --steps 200 --warmup 20 > gpurun_out/r3i_ab_bench.txt 2>&1; cat gpurun_out/r3i_ab_bench.txt
bash tools/ab_configs.sh config3 config4 config5 --iters 10 > gpurun_out/r3i_ab_configs.txt 2>&1; cat gpurun_out/r3i_ab_configs.txt
bash tools/ab_robots.sh > gpurun_out/r3i_ab_robots.txt 2>&1; cat gpurun_out/r3i_ab_robots.txt
