#!/bin/bash
# Matrix-pipe busy of the bf16 matrix kernels of the stress forward from their micro-benchmarks (same shapes as the model's
# launches: tools/ffn_bf16_bench.py, tools/rel_head_bf16_bench.py), one rocprofv3 --pmc pass each.
# -> gpurun_out/<tag>_stress_mfma_pmc.{json,txt}; copy to profiles/.
set -u
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_stress_mfma_${tag}
rm -rf $out; mkdir -p $out
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY"
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $out/ffn -o pmc -- python3 tools/ffn_bf16_bench.py --fused-only --iters 3 > $out/ffn.log 2>&1
echo "ffn pass rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $out/rel -o pmc -- python3 tools/rel_head_bf16_bench.py --iters 3 > $out/rel.log 2>&1
echo "rel pass rc=$?"
find $out -name "*.db" | head
python3 tools/mfma_busy.py $out --out gpurun_out/${tag}_stress_mfma_pmc.json > gpurun_out/${tag}_stress_mfma_pmc.txt 2>&1
cat gpurun_out/${tag}_stress_mfma_pmc.txt
find $out -name "*.db" -delete
