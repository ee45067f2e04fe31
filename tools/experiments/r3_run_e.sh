set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3e_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3e_tests.log; tail -3 gpurun_out/r3e_tests.log
grep -q "tests rc=0" gpurun_out/r3e_tests.log || exit 1
bash tools/ab_configs.sh config3 config5 config5_uniform_starts --iters 10 > gpurun_out/r3e_ab_configs.txt 2>&1; cat gpurun_out/r3e_ab_configs.txt
bash tools/ab_configs.sh config3 config5 --iters 10 > gpurun_out/r3e_ab_configs2.txt 2>&1; cat gpurun_out/r3e_ab_configs2.txt
