# Round-3 closing run on the GPU box (one gpurun call): parity suite, PMC profile of the bench command, the bench line (quoting that PMC summary), the other BASELINE configs with rake statistics and shard probe, robots, a short soak
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3y_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3y_tests.log; tail -3 gpurun_out/r3y_tests.log
grep -q "tests rc=0" gpurun_out/r3y_tests.log || exit 1
timeout -k 10 400 bash tools/profile.sh r3y_bench > gpurun_out/r3y_prof.log 2>&1; echo "profile rc=$?"; tail -2 gpurun_out/r3y_prof.log
cp profiles/r03_pmc.json gpurun_out/r3y_r03_pmc.json
timeout -k 10 200 python bench.py > gpurun_out/r3y_bench.json 2> gpurun_out/r3y_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3y_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['other_kernels_ms'], r['frac'], r['traffic'], r['pmc_note'], d['two_streams']['value'], json.dumps(d['shard_probe']['shards']))"
timeout -k 10 400 python tools/bench_configs.py --stats --shard-probe config2 config3 config4 config4_uniform_starts config5 config5_uniform_starts mixed_panda mixed_ur5 mixed_fetch mixed_baxter --iters 10 > gpurun_out/r3y_configs.jsonl 2> gpurun_out/r3y_configs.err; echo "configs rc=$?"; python3 -c "
import json
for l in open('gpurun_out/r3y_configs.jsonl'):
    d=json.loads(l); print(d['config'], d['robot'], round(d['ms'],4), '%.3e'%d['value'], d['unit'], round(d['valid_fraction'],3))"
timeout -k 10 200 python tools/bench_robots.py > gpurun_out/r3y_robots.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r3y_robots.jsonl
timeout -k 10 400 python tools/fuzz_gpu.py --minutes 5 --seed 33 > gpurun_out/r3y_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r3y_fuzz.log
timeout -k 10 300 python tools/bench_configs.py prm_panda prm_fetch prm_baxter > gpurun_out/r3y_prm_robots.jsonl 2>/dev/null; echo "prm rc=$?"; python3 -c "
import json
for l in open('gpurun_out/r3y_prm_robots.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config'], d['robot'], round(d['ms'],4), '%.3e'%d['value'], d['unit'], round(d['valid_fraction'],3))"
