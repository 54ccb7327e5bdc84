#!/bin/bash
# LDS counters of k_assemble_s3_hex2 for the profiling builds named in HXV (petsc_amd/csrc/exp/libfemasm_hx<v>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
source <(sed -n '/^run()/,/^}/p' $R/scripts/pmc_k.sh)
cd /tmp; KONLY=--k-only; EXTRA=--perturb; export KPAT=k_assemble_s3_hex2
for v in $HXV; do
  export FEMASM_LIB=$R/petsc_amd/csrc/exp/libfemasm_hx$v.so
  echo "== variant $v"; run sq1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU || exit 1
done
