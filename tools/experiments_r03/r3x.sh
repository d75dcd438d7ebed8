mkdir -p gpurun_out/prof
export PCS_TRACK_ENGINES=1
python3 -c "import bench; bench.load_sequences([0], 20, 160000)" || exit 1
bash tools/pmc_pass.sh r03_tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_GATE_EN1_sum; rc=$?; echo "tlb rc=$rc"
[ $rc -ne 0 ] && tail -5 gpurun_out/prof/r03_tlb.err
bash tools/pmc_pass.sh r03_tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum; rc=$?; echo "tcp rc=$rc"
[ $rc -ne 0 ] && tail -5 gpurun_out/prof/r03_tcp.err
bash tools/pmc_pass.sh r03_ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE; rc=$?; echo "ta rc=$rc"
[ $rc -ne 0 ] && tail -5 gpurun_out/prof/r03_ta.err
true
