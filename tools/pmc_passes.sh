#!/bin/bash
# Collect PMC counters for a command in separate rocprofv3 passes (counters never combined with tracing other
# than --kernel-trace).  Usage: tools/pmc_passes.sh <outdir> <tag> <pass-set> -- <program> [args...]
#   pass-set: "mem" (FETCH/WRITE/L2/L1), "sq" (wave/instruction/LDS counters), "mfma" (matrix-core busy + sq), "all"
# Each pass runs under its own `timeout 150` (TA_* counters were seen to abort rocprofv3 on this pool: not used).
set -u
out=$1; tag=$2; set_=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
MEM=("FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum")
SQ=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
    "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD" \
    "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL")
MFMA=("SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY")
case "$set_" in mfma) P=("${MFMA[@]}" "${SQ[@]}");; mem) P=("${MEM[@]}");; sq) P=("${SQ[@]}");; *) P=("${MEM[@]}" "${SQ[@]}");; esac
i=0
for ctrs in "${P[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs -d "$out/${tag}_${set_}$i" -o pmc -- "$@" > "$out/${tag}_${set_}$i.log" 2>&1
  echo "pass $set_$i ($ctrs): rc=$?"
done
