"""What a drain waits for: the longest histories of a workload and the time a history takes per move
when it is (nearly) alone on the chip.
    python tools/longest_history.py crustpinch [toa_degree=9] [histories=4000000]
Traced runs (final records carry the move count) in passes of 1e6; then the single longest history is
run again on its own and timed: its moves x that time per move is the floor of any drain it is part of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else "crustpinch"
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
m = Model(CONFIGS[name](deg) + ["--device-tables"]); e = Engine(m)
moves = []
per = 1_000_000
for first in range(0, n, per):
    _, fin = e.run(min(per, n - first), first_id=first, trace=True)
    moves.append(np.frombuffer(fin, dtype=np.dtype([("f", "<f8", 9), ("moves", "<u4"), ("fate", "u1"), ("type", "u1"), ("ncatch", "<u2")]))["moves"].copy())
moves = np.concatenate(moves)
top = np.argsort(moves)[-5:][::-1]
q = np.quantile(moves, [0.5, 0.9, 0.99, 0.999, 0.9999, 0.99999])
print(f"{name} deg {deg}, {n} histories: moves mean {moves.mean():.1f}, median {q[0]:.0f}, 90 % {q[1]:.0f}, 99 % {q[2]:.0f}, "
      f"99.9 % {q[3]:.0f}, 99.99 % {q[4]:.0f}, 99.999 % {q[5]:.0f}, longest {moves[top].tolist()} (ids {top.tolist()})")
for hid in top[:3]:
    e.run(1, first_id=int(hid))
    ms = min((e.run(1, first_id=int(hid)), e.last_kernel_ms())[1] for _ in range(3))
    print(f"  history {hid}: {moves[hid]} moves alone on the chip in {ms:.3f} ms = {1e3 * ms / moves[hid]:.2f} us per move")
share = [float(moves[moves > t].sum()) / float(moves.sum()) for t in (128, 256, 512, 1024)]
print("  share of all moves made beyond move 128 / 256 / 512 / 1024 of their history: " + " / ".join(f"{100 * s:.2f} %" for s in share))
e.close()
