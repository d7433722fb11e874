#!/bin/bash
# A/B of the fused bottleneck tail (egtr_amd.backbone.CONV3_FUSED / CONV3_FUSED_BF16; csrc/conv_tail_x6.hip, conv_tail_bf16.hip):
# stand-alone per layer, the kernels' durations inside the forward (rocprofv3 kernel trace), then the bench headline and the
# stress forward, alternating on one box.  The switches are module attributes: the "off" runs patch them before bench.py starts.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run_bench() {   # $1 = 0 | 1, rest = bench.py arguments
  local f=$1; shift
  python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.CONV3_FUSED = b.CONV3_FUSED and bool($f); b.CONV2_X6 = b.CONV2_X6 and bool(int('${CONV2:-1}')); \
b.CONV3_FUSED_BF16 = getattr(b, 'CONV3_FUSED_BF16', False) and bool($f); sys.argv = ['bench.py'] + sys.argv[1:]; \
runpy.run_path('bench.py', run_name='__main__')" "$@"
}
{
timeout 300 python3 tools/conv3_fused_ab.py
for f in 1 0; do
  rm -rf gpurun_out/prof_c3
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_c3 -o c3 -- python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.CONV3_FUSED = bool($f); sys.argv = ['bench.py', '--steps', '100', '--warmup', '20', '--no-cpu-baseline', '--no-kernel-probes', '--extras', '0']; runpy.run_path('bench.py', run_name='__main__')" > /dev/null 2>&1
  echo "== in-forward kernel times, CONV3_FUSED=$f"
  python3 tools/forward_breakdown.py gpurun_out/prof_c3/c3_results.db 40 2>&1 | grep -E "one forward|backbone|conv_tail|bias_act"
  rm -rf gpurun_out/prof_c3
done
for i in 1 2; do
  for f in 0 1; do
    echo "== CONV3_FUSED=$f run $i: images/s, ms per step"
    run_bench $f --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-probes --extras 0 \
      | python3 -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
for i in 1 2; do
  for f in 0 1; do
    echo "== ${BF16_SWITCH:-CONV3_FUSED_BF16}=$f run $i: stress forward (800x1333, N=300, 8 decoder layers, bf16, bs 16)"
    python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.${BF16_SWITCH:-CONV3_FUSED_BF16} = bool($f); sys.argv = ['stress_bench.py', '--iters', '10']; runpy.run_path('tools/stress_bench.py', run_name='__main__')" 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
} > gpurun_out/conv3_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/conv3_ab.txt | cut -c1-230
