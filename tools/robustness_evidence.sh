mkdir -p gpurun_out/r06x
(echo "# tools/soak.py on the round-6 sources"; timeout -k 10 300 python tools/soak.py 400 2>&1 | grep -v "^|\|Warn\|amdgpu") > gpurun_out/r06x/soak.log
(echo "# tools/long_chain.py on the round-6 sources: K chained launches of 1e7 + flush; every history accounted for, catches == bin counts"
 for c in "crustpinch 300" "lopnor 300" "sphere_deep 40"; do set -- $c; timeout -k 10 200 python tools/long_chain.py $1 9 10000000 $2 2>&1 | grep -v "^|\|Warn\|amdgpu"; done) > gpurun_out/r06x/long_chain.log
(echo "# python -m tests.margins: observed deviations of the engine from the oracle, TOA degree 4"; timeout -k 10 400 python -m tests.margins 2>&1 | grep -v "^|\|Warn\|amdgpu") > gpurun_out/r06x/margins.log
bash tools/comm_rccl_trace.sh gpurun_out/r06x > gpurun_out/r06x/comm_trace.log 2>&1
tail -3 gpurun_out/r06x/soak.log; cat gpurun_out/r06x/long_chain.log gpurun_out/r06x/margins.log; tail -12 gpurun_out/r06x/comm_trace.log; head -12 gpurun_out/r06x/kernel_stats_bench_launched_world1.csv
