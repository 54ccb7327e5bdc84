#!/bin/bash
# every entry of profiles/rNN in one go (scripts/make_profiles.sh per tag), then: python scripts/collect_profiles.py rNN
R=$GRAFT_REPO_ROOT
set -e
bash $R/scripts/make_profiles.sh elastic3d > $R/gpurun_out/mp_elastic3d.log 2>&1; echo elastic3d done
bash $R/scripts/make_profiles.sh poisson3d --workload poisson3d > $R/gpurun_out/mp_poisson3d.log 2>&1; echo poisson3d done
bash $R/scripts/make_profiles.sh elastic2d --workload elastic2d > $R/gpurun_out/mp_elastic2d.log 2>&1; echo elastic2d done
bash $R/scripts/make_profiles.sh elastic3d_manufactured --manufactured-loads > $R/gpurun_out/mp_e3m.log 2>&1; echo manufactured done
bash $R/scripts/make_profiles.sh elastic3d_general --perturb > $R/gpurun_out/mp_e3g.log 2>&1; echo elastic3d_general done
bash $R/scripts/make_profiles.sh poisson3d_general --workload poisson3d --perturb > $R/gpurun_out/mp_p3g.log 2>&1; echo poisson3d_general done
bash $R/scripts/make_profiles.sh elastic2d_general --workload elastic2d --perturb > $R/gpurun_out/mp_e2g.log 2>&1; echo elastic2d_general done
bash $R/scripts/make_profiles.sh enriched3d --workload enriched3d > $R/gpurun_out/mp_enr.log 2>&1; echo enriched3d done
bash $R/scripts/make_profiles.sh enriched3d_general --workload enriched3d --perturb > $R/gpurun_out/mp_enrg.log 2>&1; echo enriched3d_general done
bash $R/scripts/make_profiles.sh enriched3d_valu --workload enriched3d --enr-contraction valu > $R/gpurun_out/mp_enrv.log 2>&1; echo enriched3d_valu done
bash $R/scripts/make_profiles.sh enriched3d_mfma --workload enriched3d --enr-contraction mfma > $R/gpurun_out/mp_enrm.log 2>&1; echo enriched3d_mfma done
