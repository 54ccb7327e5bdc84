#!/bin/bash
# every entry of profiles/rNN (scripts/make_profiles.sh per tag), then: python scripts/collect_profiles.py rNN
# usage: make_all_profiles.sh [a|b]   (a: the four kinds and their general-geometry variants, b: the enriched entries;
# one gpurun call holds about one half)
R=$GRAFT_REPO_ROOT
set -e
half=${1:-ab}
run() { tag=$1; shift; bash $R/scripts/make_profiles.sh $tag "$@" > $R/gpurun_out/mp_$tag.log 2>&1; echo $tag done; }
if [[ $half == *a* ]]; then
run elastic3d
run poisson3d --workload poisson3d
run elastic2d --workload elastic2d
run elastic3d_manufactured --manufactured-loads
run elastic3d_general --perturb
run poisson3d_general --workload poisson3d --perturb
run elastic2d_general --workload elastic2d --perturb
fi
if [[ $half == *b* ]]; then
run enriched3d --workload enriched3d
run enriched3d_general --workload enriched3d --perturb
run enriched3d_valu --workload enriched3d --enr-contraction valu
run enriched3d_mfma --workload enriched3d --enr-contraction mfma
fi
