# tools/ab_round6_inline_rt.sh -- same-call A/B of the interface solve inside the MOVE phase (r3d_pool.h kInlineRt):
#   variant_TI0.so: -DR3D_TAIL_INLINE_RT=0 (layered models only, as first built)   variant_CI0.so: -DR3D_CYL_INLINE_RT=0
rm -f gpurun_out/ab.log
bash tools/ab_variants.sh "TI0 main CI0 TI0 main CI0" lopnor
bash tools/ab_variants.sh "TI0 main TI0 main" crustpinch sphere_deep crustpinch_vids
python tools/lone_history_time.py crustpinch 9 2937872 variant_TI0.so libr3d_hip.so 2>&1 | grep -v "^|"
python tools/lone_history_time.py lopnor 9 3142726 variant_CI0.so libr3d_hip.so 2>&1 | grep -v "^|"
