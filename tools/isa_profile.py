"""Static instruction histogram of one kernel from `llvm-objdump -d -l` output (built with
-gline-tables-only): instructions per source line (innermost inlined location) and per
mnemonic class.  Usage: isa_profile.py listing.lst [topN]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = "?"
by_line = collections.Counter(); by_class = collections.Counter(); by_file = collections.Counter()
by_line_valu = collections.Counter()
n = 0
for ln in lines:
    m = re.match(r"^; (\S+):(\d+)", ln)
    if m:
        cur = (m.group(1).split("/")[-1], int(m.group(2))); continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
    if not m or "//" not in ln:
        continue
    op = m.group(1); n += 1
    by_line[cur] += 1
    by_file[cur[0]] += 1
    cls = op.split("_")[0] + "_" + (op.split("_")[1] if "_" in op else "")
    if op.startswith("v_") and ("f64" in op): cls = "v_*_f64"
    elif op.startswith("v_cndmask"): cls = "v_cndmask"
    elif op.startswith("v_cmp"): cls = "v_cmp"
    elif op.startswith("v_"): cls = "v_other"
    elif op.startswith("s_"): cls = "s_*"
    elif op.startswith("global_") or op.startswith("flat_"): cls = "global"
    elif op.startswith("ds_"): cls = "ds"
    elif op.startswith("scratch_") or op.startswith("buffer_"): cls = "scratch"
    by_class[cls] += 1
print("total instructions", n)
print("by class:", dict(by_class.most_common()))
print("by file:", dict(by_file.most_common()))
print("top lines:")
for (f, l), c in by_line.most_common(top):
    print("  %5d  %s:%d" % (c, f, l))
