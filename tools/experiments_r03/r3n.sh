mkdir -p gpurun_out/prof
rm -f gpurun_out/prof/*.progress
bash tools/profile_batch.sh r03 > gpurun_out/prof/r03_prof.log 2>&1; echo "profile_batch rc=$?"; tail -2 gpurun_out/prof/r03_prof.log
bash tools/pmc_batch.sh r03 > gpurun_out/prof/r03_pmc.log 2>&1; echo "pmc_batch rc=$?"; tail -12 gpurun_out/prof/r03_pmc.log
bash tools/profile_sequence.sh r03 > gpurun_out/prof/r03_seq.log 2>&1; echo "profile_sequence rc=$?"; tail -3 gpurun_out/prof/r03_seq.log
