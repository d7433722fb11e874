#!/bin/bash
# A/B of a backbone switch (default egtr_amd.backbone.CONV2_X6, the own 3x3 convolutions, csrc/conv3x3_x6.hip; SWITCH=STEM_FUSED:
# the fused stem, csrc/stem_x6.hip): kernel times inside the forward and the bench headline, alternating on one box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
if [ "${SWITCH:-CONV2_X6}" = "CONV2_X6" ]; then timeout 300 python3 tools/conv3x3_ab.py; else timeout 300 python3 tools/stem_ab.py; fi
for f in 1 0; do
  rm -rf gpurun_out/prof_c2
  timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_c2 -o c2 -- python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.${SWITCH:-CONV2_X6} = bool($f); sys.argv = ['bench.py', '--steps', '100', '--warmup', '20', '--no-cpu-baseline', '--no-kernel-probes', '--extras', '0']; runpy.run_path('bench.py', run_name='__main__')" > /dev/null 2>&1
  echo "== in-forward kernel times, ${SWITCH:-CONV2_X6}=$f"
  python3 tools/forward_breakdown.py gpurun_out/prof_c2/c2_results.db 40 2>&1 | grep -E "one forward|backbone|conv3x3|conv_tail|igemm|grouped_conv|SubTensor|stem|Sp3Asm|maxpool"
  rm -rf gpurun_out/prof_c2
done
for i in 1 2; do
  for f in 0 1; do
    echo "== ${SWITCH:-CONV2_X6}=$f run $i: images/s, ms per step"
    python3 -c "import sys, runpy; import egtr_amd.backbone as b; b.${SWITCH:-CONV2_X6} = bool($f); sys.argv = ['bench.py', '--steps', '300', '--warmup', '30', '--no-cpu-baseline', '--no-kernel-probes', '--extras', '0']; runpy.run_path('bench.py', run_name='__main__')" \
      | python3 -c "import sys,json; [print(json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
  done
done
} > gpurun_out/${OUT:-conv2_ab}.txt 2>&1
grep -v amdgpu.ids gpurun_out/${OUT:-conv2_ab}.txt | cut -c1-200
