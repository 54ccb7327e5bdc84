#!/bin/bash
# SQ counter passes (scripts/pmc_sq.sh) of the two kernels rebuilt in round 2, kept as text for profiles/rNN/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
KPAT=k_assemble_s3 bash $R/scripts/pmc_sq.sh > $R/gpurun_out/profiles/sq_counters_elastic3d_k_assemble_s3_unit.txt 2>&1 || exit 1
KPAT=k_assemble_ke EXTRA="--workload poisson3d --perturb" bash $R/scripts/pmc_sq.sh > $R/gpurun_out/profiles/sq_counters_poisson3d_general_k_assemble_ke_p3.txt 2>&1 || exit 1
KPAT=k_assemble_p3_rect EXTRA="--workload poisson3d" bash $R/scripts/pmc_sq.sh > $R/gpurun_out/profiles/sq_counters_poisson3d_k_assemble_p3_rect.txt 2>&1 || exit 1
tail -n 30 $R/gpurun_out/profiles/sq_counters_poisson3d_k_assemble_p3_rect.txt
