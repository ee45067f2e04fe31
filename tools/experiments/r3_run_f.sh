set -o pipefail
mkdir -p gpurun_out
for g in 2 3 4 5 6; do
  VMV_SELF_GROUP=$g timeout -k 10 120 python bench.py --no-cpu-baseline --no-two-streams --no-shard-probe --steps 100 > gpurun_out/r3f_bench_group$g.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r3f_bench_group$g.json')); print('group $g', round(d['value']/1e9,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), d['roofline']['other_kernels_ms'])"
done
timeout -k 10 200 python tools/bench_configs.py --shard-probe --stats config4 config4_uniform_starts --iters 10 > gpurun_out/r3f_shard_probe_edges.jsonl 2>/dev/null; echo "edges rc=$?"; cut -c1-200 gpurun_out/r3f_shard_probe_edges.jsonl; python3 -c "
import json
for l in open('gpurun_out/r3f_shard_probe_edges.jsonl'):
    d=json.loads(l); print(d['config'], d['ms'], json.dumps(d['shard_probe']))"
timeout -k 10 200 python tools/bench_robots.py > gpurun_out/r3f_robots.jsonl 2>/dev/null; cut -c1-140 gpurun_out/r3f_robots.jsonl
