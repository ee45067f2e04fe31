bash tools/ab_bench.sh --no-cpu-baseline --no-two-streams --no-shard-probe --steps 200 --warmup 20 > gpurun_out/r3m_ab_bench.txt 2>&1; cat gpurun_out/r3m_ab_bench.txt
bash tools/ab_robots.sh > gpurun_out/r3m_ab_robots.txt 2>&1; cat gpurun_out/r3m_ab_robots.txt
bash tools/ab_configs.sh config4 --iters 10 > gpurun_out/r3m_ab_configs.txt 2>&1; cat gpurun_out/r3m_ab_configs.txt
