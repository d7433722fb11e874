#!/bin/bash
# Round-6, after the bottleneck-tail kernels: GPU test suite, forward breakdowns (headline + stress), the default bench command.
set -u
tag=r06
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest.log 2>&1; tail -3 gpurun_out/${tag}_gputest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
rm -rf gpurun_out/prof_fb
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_fb -o fb -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-probes --extras 0 > /dev/null 2>&1
python3 tools/forward_breakdown.py gpurun_out/prof_fb/fb_results.db 16 > gpurun_out/${tag}_forward_breakdown.txt 2>&1
rm -rf gpurun_out/prof_fb
bash tools/stress_breakdown.sh ${tag} > /dev/null 2>&1
python3 - <<'PY'
import re
p = "gpurun_out/r06_stress_forward_breakdown.txt"
lines = open(p).read().splitlines()
out, n = [], 0
for l in lines:          # keep the 24 largest kernels of each phase (the full list is 360 lines)
    if not l.strip() or not l.startswith("  "):
        n = 0
        out.append(l)
        continue
    n += 1
    if n <= 24:
        out.append(l)
open(p, "w").write("\n".join(out) + "\n")
PY
( time python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ) 2> gpurun_out/${tag}_bench_time.txt
cut -c1-700 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench_time.txt
python3 -c "
import json; d=json.load(open('gpurun_out/r06_bench.json'))
for k in ('eager','mixed_shapes','batched_bs8','train_step','stress_bf16'):
    v=d.get(k,{}); print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
print('parity', d.get('parity_vs_oracle'))"
