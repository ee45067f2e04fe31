set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r2b_tests.log; tail -3 gpurun_out/r2b_tests.log
grep -q "tests rc=0" gpurun_out/r2b_tests.log || exit 1
timeout -k 10 400 bash tools/profile.sh r02_final > gpurun_out/r2b_prof.log 2>&1; echo "profile rc=$?"
PROFILE_PROG=tools/bench_configs.py timeout -k 10 300 bash tools/profile.sh r02_config3 config3 --iters 5 > gpurun_out/r2b_prof_c3.log 2>&1; echo "c3 rc=$?"
PROFILE_PROG=tools/bench_configs.py timeout -k 10 300 bash tools/profile.sh r02_config5 config5 --iters 5 > gpurun_out/r2b_prof_c5.log 2>&1; echo "c5 rc=$?"
timeout -k 10 200 python bench.py > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err; echo "bench rc=$?"; cat gpurun_out/r2b_bench.json | cut -c1-400
timeout -k 10 200 python tools/bench_configs.py config3 config4 config5 --iters 10 > gpurun_out/r2b_configs.jsonl 2>/dev/null; cat gpurun_out/r2b_configs.jsonl | cut -c1-130
timeout -k 10 200 python tools/bench_robots.py > gpurun_out/r2b_robots.jsonl 2>/dev/null; cat gpurun_out/r2b_robots.jsonl | cut -c1-130
