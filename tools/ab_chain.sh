#!/bin/bash
# tools/ab_chain.sh [configs...] -- chained-launch kernel time per config, for each value of R3D_KERNEL given
# in $KERNELS (default "pool lanes"); run on the GPU box from the repo root.
configs=${@:-crustpinch halfspace lopnor sphere_deep crustpinch_vids}
for c in $configs; do
  n=10000000; [ $c = sphere_deep ] && n=3000000
  for k in ${KERNELS:-pool lanes}; do
    R3D_KERNEL=$k timeout -k 10 200 python3 tools/time_chain.py $c 9 $n 5 2>&1 | grep "chained launch" | sed "s/^/[$k] /"
  done
done
