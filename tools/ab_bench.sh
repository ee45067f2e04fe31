#!/bin/bash
# A/B on the GPU box: the in-tree library and every variants/<name>/libvamp_mvt_amd.so, same bench command.
# usage: bash tools/ab_bench.sh [bench args]   (prints: name  env_ms  self_ms  checks/s)
ARGS=${@:---no-cpu-baseline --steps 100 --warmup 20}
run() {
  VMV_LIBRARY=$2 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['roofline']['kernel_ms'],4), round(d['roofline']['other_kernels_ms']['validate_self_kernel'],4), '%.4g' % d['value'])"
}
run base ""
for v in variants/*/libvamp_mvt_amd.so; do [ -f "$v" ] && run $(basename $(dirname $v)) $PWD/$v; done
run base-again ""
