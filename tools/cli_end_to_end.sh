#!/bin/bash
# Wall time of the command-line program on the NSCP configuration (do-crustpinch.sh arguments),
# host-built tables (--host-tables) vs the default, tables built in HBM.  Run on the GPU box from the repo root.
A="--grid-compiled=5 --model-args=0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.01,0.20,0.3,1500,0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900,2.0,30.0,5.0,.3666667,.4736842,1,1 --source=SDR,22.5,90,0 --source-loc=0,0,-10 --frequency=2.0 --timetolive=600 --binsize=2.00 --toa-degree=9 --seis-p2p=0,67.5,0,950,67.5,0,1.0,2.0,40.0,160 --seis-p2p=0,112.5,0,950,112.5,0,1.0,2.0,40.0,160 --seis-p2p=0,90,0,950,90,0,1.0,2.0,40.0,160"
n=${1:-100M}
mkdir -p /tmp/r3d_cli && cd /tmp/r3d_cli
for extra in "--host-tables" ""; do
  rm -rf out && mkdir out
  s=$(date +%s%N)
  timeout -k 10 300 $GRAFT_REPO_ROOT/main $A --num-phonons=$n --output-dir=out $extra > log.txt 2> err.txt || { echo FAILED; tail -5 log.txt err.txt; exit 1; }
  e=$(date +%s%N)
  echo "main --num-phonons=$n $extra: $(( (e - s) / 1000000 )) ms wall; $(ls out | wc -l) files; $(grep -c . log.txt) stdout lines"
  grep -E "Loss surfaces|Timeout|Invalid" log.txt | tr '\n' ' '; echo
done
