# Round-3 closing run on the GPU box (one gpurun call): parity suite, soak, headline bench, PMC profile of the bench command,
# bench again (so that the line quotes the PMC summary of this very build), the other BASELINE configs, planners, edge sizes.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3z_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3z_tests.log; tail -3 gpurun_out/r3z_tests.log
grep -q "tests rc=0" gpurun_out/r3z_tests.log || exit 1
timeout -k 10 200 python bench.py > gpurun_out/r3z_bench_first.json 2> gpurun_out/r3z_bench_first.err; echo "bench rc=$?"
timeout -k 10 400 bash tools/profile.sh r3z_bench > gpurun_out/r3z_prof.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r3z_prof.log
cp profiles/r03_pmc.json gpurun_out/r3z_r03_pmc.json
timeout -k 10 200 python bench.py > gpurun_out/r3z_bench.json 2> gpurun_out/r3z_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3z_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['other_kernels_ms'], r['frac'], r['traffic'], r['pmc_note'], d['two_streams']['value'], json.dumps(d['shard_probe']['shards']))"
timeout -k 10 400 python tools/bench_configs.py --stats --shard-probe config2 config3 config4 config4_uniform_starts config5 config5_uniform_starts --iters 10 > gpurun_out/r3z_configs.jsonl 2> gpurun_out/r3z_configs.err; echo "configs rc=$?"; python3 -c "
import json
for l in open('gpurun_out/r3z_configs.jsonl'):
    d=json.loads(l); print(d['config'], d['robot'], round(d['ms'],4), '%.3e'%d['value'], d['unit'], round(d['valid_fraction'],3), json.dumps(d.get('shard_probe')))"
timeout -k 10 200 python tools/bench_robots.py > gpurun_out/r3z_robots.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r3z_robots.jsonl
timeout -k 10 300 python tools/bench_planners.py --json gpurun_out/r3z_planners.json > gpurun_out/r3z_planners.txt 2>&1; echo "planners rc=$?"; grep -v amdgpu gpurun_out/r3z_planners.txt
timeout -k 10 200 python tools/experiments/small_edge_batches.py > gpurun_out/r3z_small_edges.txt 2>&1; grep -v amdgpu gpurun_out/r3z_small_edges.txt
timeout -k 10 800 python tools/fuzz_gpu.py --minutes 12 --seed 32 > gpurun_out/r3z_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r3z_fuzz.log
