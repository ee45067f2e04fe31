set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "edge_schedules or full_size_baseline_configs or validate_motion_batch_bit_exact" > gpurun_out/r3h_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3h_tests.log; tail -3 gpurun_out/r3h_tests.log
grep -q "tests rc=0" gpurun_out/r3h_tests.log || exit 1
EDGE_MODES=1,3 timeout -k 10 300 python tools/experiments/small_edge_batches.py 256 2048 8192 16384 32768 65536 131072 262144 > gpurun_out/r3h_small_edges.txt 2>&1; echo "small rc=$?"; grep -v amdgpu gpurun_out/r3h_small_edges.txt
