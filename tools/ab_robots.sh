#!/bin/bash
# A/B of every robot's two kernels (tools/bench_robots.py) over the in-tree library and variants/*
run() { echo "== $1"; VMV_LIBRARY=$2 python tools/bench_robots.py ${ROBOTS:-ur5 fetch baxter} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  %-7s env %.4f self %.4f  %.3e/s' % (d['robot'], d['env_ms'], d['self_ms'], d['checks_per_s']))"; }
run base ""
for v in variants/*/libvamp_mvt_amd.so; do [ -f "$v" ] && run $(basename $(dirname $v)) $PWD/$v; done
