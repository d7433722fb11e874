#!/bin/bash
# Round 6: A/B of two restructurings of the fp32 MSDA forward against the shipped 4-query-workgroup kernel, on one box:
#   EGTR_MSDA_PERS = 0 (off) / <waves>x<workgroups per CU>: persistent workgroups, per-XCD query queues with stealing
#   EGTR_MSDA_BAND = 1: band mapping (XCD x serves the x-th horizontal eighth of every level)
# fused entry, encoder shape, jitter sweep + the plain entry at B = 4.  -> gpurun_out/r06_msda_pers_ab.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/r06_msda_pers_ab.txt
: > $out
run() {
  for j in 0 0.5 4; do
    echo "== $1 jitter=$j fused graph" >> $out
    env $1 python3 tools/msda_bench.py --fused --graph --jitter $j --iters 100 2>&1 | grep "^msda" >> $out
  done
  echo "== $1 jitter=0.5 plain entry, batch 4" >> $out
  env $1 python3 tools/msda_bench.py --graph --jitter 0.5 --iters 50 --batch 4 2>&1 | grep "^msda" >> $out
}
run "EGTR_MSDA_PERS=0"
run "EGTR_MSDA_PERS=0 EGTR_MSDA_BAND=1"
for cfg in 8x2 8x3 16x1 16x2; do run "EGTR_MSDA_PERS=$cfg"; done
run "EGTR_MSDA_PERS=0"
cat $out
