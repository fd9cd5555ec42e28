#!/bin/bash
# The evidence kept under profiles/, collected on one box: kernel-trace summaries of the headline bench, the sampling loop and the LoRA step, then the
# counter passes (HBM-side traffic, matrix-pipe utilisation) over the eager train step.  usage: tools/r3_final_profiles.sh <tag>
cd $GRAFT_REPO_ROOT
tag=${1:-r03}
bash tools/make_profile.sh $tag > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
echo "bench profile done"
bash tools/make_aux_profile.sh $tag > gpurun_out/prof_aux_$tag.log 2>&1 || { tail -5 gpurun_out/prof_aux_$tag.log; exit 1; }
echo "sampler / lora profiles done"
bash tools/pmc_step.sh $tag > gpurun_out/pmc_step_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_step_$tag.log; exit 1; }
echo "traffic passes done"
bash tools/pmc_step_mfma.sh $tag > gpurun_out/pmc_mfma_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_mfma_$tag.log; exit 1; }
echo "mfma pass done"
