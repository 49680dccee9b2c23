#!/bin/bash
# tools/disasm.sh KIND [extra hipcc flags] -- disassembly with source lines of the traversal kernels of one cell
# kind (0 cylinder, 1 tetra, 2 sphere): /tmp/r3d_disasm_K/dev.lst (+ per-kernel files k_<mangled>.lst)
set -e
kind=$1; shift
out=/tmp/r3d_disasm_$kind; mkdir -p $out
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-bitwise-instead-of-logical \
   -mllvm -disable-machine-licm -gline-tables-only -DR3D_DEV_ONLY_KIND=$kind "$@" -shared -pthread -o $out/lib.so \
   radiative3d_amd/csrc/r3d_engine.hip radiative3d_amd/csrc/r3d_tables_build.hip
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $out/lib.so $out/fat.bin
$L/clang-offload-bundler --type=o --unbundle --input=$out/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$out/dev.co
$L/llvm-objdump -d -l --no-show-raw-insn $out/dev.co > $out/dev.lst
python3 - $out <<'PY'
import re, sys
out = sys.argv[1]
cur, buf = None, []
def flush():
    if cur and buf:
        open(f"{out}/k_{cur[:120]}.lst", "w").write("\n".join(buf))
for ln in open(f"{out}/dev.lst"):
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
    if m:
        flush(); cur, buf = m.group(1), []
    buf.append(ln.rstrip("\n"))
flush()
PY
ls $out | head -30
