#!/bin/bash
# Collects, on the GPU box, the evidence bench.py's roofline object is checked against:
#   1. rocprofv3 --kernel-trace --stats of the bench command          -> gpurun_out/$TAG/stats
#   2. rocprofv3 --pmc passes (separate runs, counters only)          -> gpurun_out/$TAG/pmc*
#   3. tools/pmc_summary.py over the passes                           -> gpurun_out/$TAG/pmc.json
# usage (from the repo root, under gpurun):  bash tools/profile.sh TAG [bench args...]
# Copy the summaries you want judged from gpurun_out/$TAG into profiles/.
set -o pipefail
TAG=${1:-prof}; shift
ARGS=${@:---steps 30 --warmup 5 --no-cpu-baseline --no-two-streams --no-shard-probe}
PROG=${PROFILE_PROG:-bench.py}   # e.g. PROFILE_PROG=tools/bench_configs.py bash tools/profile.sh r02_config3 config3
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $PROG $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
echo "stats done" >&2
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc$i -o run -- python3 $PROG $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.log || echo "pmc pass $i failed" >&2
  echo "pmc pass $i done" >&2
done
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc6 --json $OUT/pmc.json > $OUT/pmc_summary.txt
if [ "$PROG" = "bench.py" ]; then python3 tools/make_pmc_profile.py $OUT/pmc.json && cp profiles/r03_pmc.json $OUT/r03_pmc.json; fi
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge small: drop the raw traces
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
tail -40 $OUT/pmc_summary.txt
