#!/bin/bash
# GPU box: SQ / TA counters of the one-kernel FFN forward on the micro-benchmark (three --pmc passes, summarised by
# tools/pmc_summary.py).  Output: gpurun_out/ffn1_pmc_summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/bench_kernels.py ${PMC_ARGS:-ffn1 524288}"
TAG=${PMC_TAG:-ffn1}
FILTER=${PMC_FILTER:-gemm_nt_mul}
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH"; do
  # (a pass with TA_* counters aborted rocprofv3 on this pool: not part of the script)
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/${TAG}_pmc_$i -o p --output-format csv -- $P > $R/gpurun_out/${TAG}_pmc_$i.log 2>&1 || echo "pass $i failed"
done
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_$i $FILTER; done > gpurun_out/${TAG}_pmc_summary.txt 2>&1
echo done
