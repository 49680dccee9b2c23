#!/bin/bash
# tools/round_evidence.sh TAG -- everything a round's profiles/ directory holds, in one GPU call:
#   * tools/collect_profiles.sh TAG (kernel trace + 4 PMC passes per bench workload)
#   * the bench line of every workload (default flags) and the headline with the driver's flags
# Outputs under gpurun_out/profiles_TAG/; copy the summaries into profiles/<round>/ afterwards
# (tools/copy_evidence.py).
set -e
tag=${1:-run}
out=gpurun_out/profiles_$tag
bash tools/collect_profiles.sh $tag
for c in crustpinch halfspace lopnor sphere crustpinch_volume; do
  echo "== bench line: $c" >&2
  timeout -k 10 600 python3 bench.py --config $c > $out/bench_line_$c.json 2> $out/bench_line_$c.log
done
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_line_crustpinch_steps20_warmup5.json 2> $out/bench_line_crustpinch_steps20_warmup5.log
