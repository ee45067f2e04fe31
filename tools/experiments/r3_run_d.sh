set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3d_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3d_tests.log; tail -3 gpurun_out/r3d_tests.log
grep -q "tests rc=0" gpurun_out/r3d_tests.log || exit 1
timeout -k 10 200 python tools/experiments/small_edge_batches.py 256 2048 16384 131072 1048576 > gpurun_out/r3d_small_edges.txt 2>&1; echo "small rc=$?"; grep -v amdgpu gpurun_out/r3d_small_edges.txt
VMV_FUSED_KERNEL=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-two-streams > gpurun_out/r3d_bench_fused.json 2> gpurun_out/r3d_bench_fused.err; echo "fused rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3d_bench_fused.json')); print('fused', d['value'], d['ms_per_step'], json.dumps(d.get('shard_probe',{}).get('shards')))"
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r3d_bench.json')); print('two kernels', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels_ms'], d['two_streams']['value'], json.dumps(d.get('shard_probe',{}).get('shards')))"
timeout -k 10 400 bash tools/profile.sh r3d_bench > gpurun_out/r3d_prof.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/r3d_prof.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d_small_prof -o run -- python3 tools/experiments/small_edge_batches.py 2048 > gpurun_out/r3d_small_prof.txt 2>&1; echo "prof rc=$?"
find gpurun_out/r3d_small_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3d_small_kernel_stats.csv \;
find gpurun_out/r3d_small_prof -name "*.csv" -size +2M -delete; find gpurun_out/r3d_small_prof -name "*.db" -delete
cut -c1-60,200-330 gpurun_out/r3d_small_kernel_stats.csv | head -12
timeout -k 10 330 python tools/fuzz_gpu.py --minutes 5 --seed 31 > gpurun_out/r3d_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r3d_fuzz.log
