#!/bin/bash
# tools/disasm.sh KIND [extra hipcc flags] -- disassembly with source lines of the traversal kernels of one cell
# kind (0 cylinder, 1 tetra, 2 sphere): /tmp/r3d_disasm_K/dev.lst (+ per-kernel files k_<mangled>.lst); built with
# the Makefile's flags (HIPFLAGS and the kind's own HIPFLAGS_CYL / _TET / _SPH).
set -e
kind=$1; shift
out=/tmp/r3d_disasm_$kind; mkdir -p $out
case $kind in 0) tag=CYL;; 1) tag=TET;; *) tag=SPH;; esac
kflags=$(make -s -f - print <<MK
include Makefile
print:
	@echo \$(HIPFLAGS) \$(HIPFLAGS_$tag)
MK
)
/opt/rocm/bin/hipcc $kflags -gline-tables-only -DR3D_KIND=$kind "$@" -c -o $out/unit.o \
   radiative3d_amd/csrc/r3d_kernels_kind.hip
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $out/unit.o $out/fat.bin
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
