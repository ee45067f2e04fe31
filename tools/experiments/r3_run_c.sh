set -o pipefail
mkdir -p gpurun_out
for m in 0 1 2; do
  VMV_EDGE_TASKS=$m timeout -k 10 300 python tools/bench_configs.py config4 config5 config5_uniform_starts --iters 10 > gpurun_out/r3c_configs_mode$m.jsonl 2> gpurun_out/r3c_configs_mode$m.err; echo "mode $m rc=$?"; cut -c1-160 gpurun_out/r3c_configs_mode$m.jsonl
done
VMV_EDGE_TASKS=1 PROFILE_PROG=tools/bench_configs.py timeout -k 10 400 bash tools/profile.sh r3c_config4_tasks config4 --iters 5 > gpurun_out/r3c_prof_c4.log 2>&1; echo "c4 prof rc=$?"
VMV_EDGE_TASKS=1 PROFILE_PROG=tools/bench_configs.py timeout -k 10 400 bash tools/profile.sh r3c_config5_tasks config5 --iters 5 > gpurun_out/r3c_prof_c5.log 2>&1; echo "c5 prof rc=$?"
