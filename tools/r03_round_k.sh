#!/bin/bash
# round 3, step k: counters of cin_dw_bf3_k (two passes) -- what keeps the matrix pipe at 0.56
cd "$GRAFT_REPO_ROOT"
export DIR_BENCH_NO_SECONDARY=1
bash tools/pmc2.sh cdw_a cin_dw_bf3_k "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" -- --workload cin_backward --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_pmc_cdw_a.txt 2>&1; tail -14 gpurun_out/r03_pmc_cdw_a.txt
bash tools/pmc2.sh cdw_b cin_dw_bf3_k "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" -- --workload cin_backward --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_pmc_cdw_b.txt 2>&1; tail -14 gpurun_out/r03_pmc_cdw_b.txt
