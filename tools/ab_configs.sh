#!/bin/bash
# A/B of tools/bench_configs.py over the in-tree library and variants/*   usage: bash tools/ab_configs.sh config4 config5
run() { echo "== $1"; VMV_LIBRARY=$2 python tools/bench_configs.py "${@:3}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %-8s %-7s %.4f ms  %.3e %s' % (d['config'], d['robot'], d['ms'], d['value'], d['unit']))"; }
run base "" "$@"
for v in variants/*/libvamp_mvt_amd.so; do [ -f "$v" ] && run $(basename $(dirname $v)) $PWD/$v "$@"; done
