#!/bin/bash
# HBM traffic passes only (WRITE_SIZE, FETCH_SIZE) over the K-only bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
source <(sed -n '/^run()/,/^}/p' $R/scripts/pmc_k.sh)
cd /tmp; : ${KONLY=--k-only}
run wr WRITE_SIZE && run rd FETCH_SIZE
