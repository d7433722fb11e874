#!/bin/bash
# Does MIOpen's find mode matter?  The bench headline and the stress forward under MIOPEN_FIND_MODE = default (dynamic hybrid),
# 1 (normal: every applicable solver is timed), 3 (hybrid); wall time of each process = what the search costs on a fresh box.
# (Disabling the assembly implicit-GEMM NHWC solver, MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0, was measured
# first: headline 341 -> 300 images/s, stress 697 -> 250 -- it is the good solver for all shapes but one.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for v in default 1 3; do
  if [ "$v" = "default" ]; then unset MIOPEN_FIND_MODE; else export MIOPEN_FIND_MODE=$v; fi
  echo "== MIOPEN_FIND_MODE: $v"
  ( time timeout 900 python3 tools/stress_bench.py --iters 10 --find 1 2>&1 | grep -E "HIP graph" ) 2>&1 | grep -E "stress|real"
done
unset MIOPEN_FIND_MODE
} > gpurun_out/miopen_solver_ab.txt 2>&1
cat gpurun_out/miopen_solver_ab.txt
