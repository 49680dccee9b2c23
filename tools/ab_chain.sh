#!/bin/bash
# tools/ab_chain.sh [configs...] -- chained-launch kernel time per config of the shipped library
# (radiative3d_amd/lib/libr3d_hip.so); run on the GPU box from the repo root.  Compare builds within
# ONE call (tools/ab_variants.sh): boxes differ by a few per cent.
configs=${@:-crustpinch halfspace lopnor sphere_deep crustpinch_vids}
for c in $configs; do
  n=10000000; [ $c = sphere_deep ] && n=3000000
  timeout -k 10 200 python3 tools/time_chain.py $c 9 $n 5 2>&1 | grep "chained launch"
done
