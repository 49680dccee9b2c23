#!/bin/bash
# tools/pmc_run.sh OUTDIR "CTR1 CTR2" "CTR3 ..."   -- one rocprofv3 --pmc pass per argument
# (counters in their own runs, no other tracing), workload = tools/time_variant.py crustpinch 9 1e7
out=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d $out/p$i -o pmc --output-format csv -- python3 tools/time_variant.py ${PMC_WORKLOAD:-crustpinch 9 10000000} > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 tools/pmc_pass.py $out
