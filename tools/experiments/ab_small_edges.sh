#!/bin/bash
# A/B of planner-sized edge batches over the in-tree library and variants/*   usage: bash tools/experiments/ab_small_edges.sh
for robot in panda ur5; do
  for v in base variants/*/libvamp_mvt_amd.so; do
    lib=""; name=base; [ "$v" != base ] && lib=$PWD/$v && name=$(basename $(dirname $v))
    echo "== $robot $name"
    SMALL_ROBOT=$robot SMALL_N=65536 VMV_LIBRARY=$lib python tools/experiments/small_edge_batches.py 256 2048 8192 2>/dev/null | grep -E "^prm|^uniform"
  done
done
