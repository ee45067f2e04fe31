#!/bin/bash
# Developer experiment: is BASELINE config 3's environment kernel bound by the vector-memory path (TA / TCP) rather than by
# latency?  usage (under gpurun): bash tools/experiments/config3_ta_probe.sh [library]
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/c3_ta
mkdir -p $OUT
[ -n "$1" ] && export VMV_LIBRARY=$1
rocprofv3 -L > $OUT/avail.txt 2>&1 || true
grep -o -E "\b(TA|TCP|TD|SQ|SQC)_[A-Za-z0-9_]+" $OUT/avail.txt | sort -u > $OUT/counter_names.txt
i=0
for CTRS in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" \
            "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -o run -- python3 tools/bench_configs.py config3 > $OUT/p$i.json 2> $OUT/p$i.log || echo "pass $i failed: $CTRS"
  echo "pass $i done"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/c3_ta/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "validate_env_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in tot: print(d, k, "per launch %.4g" % (tot[k] / n[k]), "launches", n[k])
PY
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
