for r in panda ur5; do
python tools/experiments/fused_small.py $r 2>/dev/null > gpurun_out/r3g_two_$r.txt; VMV_FUSED_KERNEL=1 python tools/experiments/fused_small.py $r 2>/dev/null > gpurun_out/r3g_fused_$r.txt
paste -d'\n' gpurun_out/r3g_two_$r.txt gpurun_out/r3g_fused_$r.txt
done
