#!/bin/bash
# tools/bench_lines.sh TAG -- the bench line of every workload (default flags) and the headline with the driver's flags,
# into gpurun_out/profiles_TAG/ (the second half of tools/round_evidence.sh: run once the counter files of these
# kernel sources are committed under profiles/<round>/, so that the lines carry `roofline.frac`).
set -e
tag=${1:-run}
out=gpurun_out/profiles_$tag
mkdir -p $out
for c in crustpinch halfspace lopnor sphere crustpinch_volume; do
  echo "== bench line: $c" >&2
  timeout -k 10 600 python3 bench.py --config $c > $out/bench_line_$c.json 2> $out/bench_line_$c.log
done
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $out/bench_line_crustpinch_steps20_warmup5.json 2> $out/bench_line_crustpinch_steps20_warmup5.log
