#!/bin/bash
# PMC passes on single GEMM shapes (run on the GPU box).  usage: gemm_pmc.sh M N K tile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=$1; N=$2; K=$3; T=$4
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" ; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set -d /tmp/pm -o p -- python tools/gemm_probe.py $M $N $K $T 8 > /dev/null 2>&1
  python tools/pmc_per_kernel.py /tmp/pm/p_results.db | grep gemm | awk -F'|' '{printf "%-34s n=%s mean=%s\n", $3, $4, $6}'
done
