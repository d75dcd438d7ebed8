mkdir -p gpurun_out/prof
rm -f gpurun_out/prof/*.progress
bash tools/profile_batch.sh r03 > gpurun_out/prof/r03_prof.log 2>&1; rc=$?; echo "profile_batch rc=$rc"; tail -2 gpurun_out/prof/r03_prof.log
[ $rc -ne 0 ] && exit $rc
bash tools/pmc_batch.sh r03 > gpurun_out/prof/r03_pmc.log 2>&1; rc=$?; echo "pmc_batch rc=$rc"; tail -12 gpurun_out/prof/r03_pmc.log
[ $rc -ne 0 ] && exit $rc
cp gpurun_out/prof/r03_icp_hbm_pmc.json profiles/r03_icp_hbm_pmc.json
timeout -k 10 500 python bench.py > gpurun_out/prof/r03_bench_default.json 2> gpurun_out/prof/r03_bench_default.err; rc=$?; echo "bench rc=$rc"; tail -c 1500 gpurun_out/prof/r03_bench_default.json
[ $rc -ne 0 ] && exit $rc
bash tools/profile_sequence.sh r03 > gpurun_out/prof/r03_seq.log 2>&1; echo "profile_sequence rc=$?"; tail -3 gpurun_out/prof/r03_seq.log
