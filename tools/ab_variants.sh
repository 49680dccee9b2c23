#!/bin/bash
# tools/ab_variants.sh "VARIANTS" [configs...] -- chained-launch kernel time of developer variants
# (radiative3d_amd/lib/variant_<NAME>.so; "main" = libr3d_hip.so) on the GPU box.
variants=$1; shift
configs=${@:-crustpinch lopnor sphere_deep}
for c in $configs; do
  n=10000000; [ $c = sphere_deep ] && n=3000000
  for v in $variants; do
    lib=radiative3d_amd/lib/variant_$v.so; [ $v = main ] && lib=radiative3d_amd/lib/libr3d_hip.so
    R3D_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 tools/time_chain.py $c 9 $n 5 2>&1 | grep "chained launch" | tee -a gpurun_out/ab.log
  done
done
