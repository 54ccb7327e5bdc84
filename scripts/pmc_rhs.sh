#!/bin/bash
# SQ + HBM traffic passes for the elastic RHS element kernel (full step bench)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KPAT=${KPAT:-k_rhs_elem_elastic} KONLY=""
source <(sed -n '/^run()/,/^}/p' $R/scripts/pmc_k.sh)
cd /tmp
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM &&
run rd FETCH_SIZE && run wr WRITE_SIZE
