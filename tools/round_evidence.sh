#!/bin/bash
# tools/round_evidence.sh TAG ROUND -- everything a round's profiles/ directory holds, in one GPU call:
#   * tools/collect_profiles.sh TAG (kernel trace + 4 PMC passes per bench workload)
#   * the counter summaries copied into profiles/ROUND/ of THIS copy of the repository (bench.py reads them from there,
#     keyed by the kernel sources' hash), so that
#   * the bench line of every workload (default flags) and the headline with the driver's flags carry valu_busy / ta_busy.
# Outputs under gpurun_out/profiles_TAG/; copy the summaries into profiles/ROUND/ afterwards (tools/copy_evidence.py TAG ROUND).
# profiles/ROUND/phase_floor.json must be current (python tools/phase_floor.py profiles/ROUND/phase_floor.json, no GPU needed).
set -e
tag=${1:-run}
round=${2:-r06}
out=gpurun_out/profiles_$tag
bash tools/collect_profiles.sh $tag
mkdir -p profiles/$round
cp $out/pmc_*.json profiles/$round/
bash tools/bench_lines.sh $tag
