set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc_step.sh r02 > gpurun_out/pmc_step.log 2>&1
bash tools/pmc_step_mfma.sh r02 > gpurun_out/pmc_mfma.log 2>&1
echo PMC done
