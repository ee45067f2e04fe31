set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3b_tests.log; tail -3 gpurun_out/r3b_tests.log
grep -q "tests rc=0" gpurun_out/r3b_tests.log || exit 1
EDGE_MODES=0,1,2 timeout -k 10 300 python tools/experiments/small_edge_batches.py > gpurun_out/r3b_small_edges.txt 2>&1; echo "small rc=$?"; cat gpurun_out/r3b_small_edges.txt
timeout -k 10 200 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3b_bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d.get('shard_probe',{}).get('shards')))"
timeout -k 10 300 python tools/bench_planners.py --json gpurun_out/r3b_planners.json > gpurun_out/r3b_planners.txt 2>&1; echo "planners rc=$?"; cat gpurun_out/r3b_planners.txt
