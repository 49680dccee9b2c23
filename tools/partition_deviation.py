"""How far two runs of the same histories, batched differently, differ in the bins (GPU): one launch against six
launches of a sixth each -- per element and per bin energy.  python tools/partition_deviation.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import CONFIGS
from radiative3d_amd.parallel import shard_range
for name, n in (("crustpinch", 240000), ("lopnor", 100000)):
    e = Engine(Model(CONFIGS[name](4)))
    whole = e.run(n, seed=5)
    parts = e.model.new_result()
    for r in range(6):
        lo, hi = shard_range(n, r, 6)
        e.run(hi - lo, first_id=lo, seed=5, result=parts)
    assert (whole.counts == parts.counts).all()
    d = np.abs(whole.energy - parts.energy)
    scale = whole.energy[:, :, 3:].sum(-1, keepdims=True)
    with np.errstate(divide='ignore', invalid='ignore'):
        rel_el = np.nanmax(np.where(whole.energy > 0, d / whole.energy, 0))
        rel_bin = np.nanmax(np.where(scale > 0, d / scale, 0))
    print(name, "max dev / element", rel_el, " max dev / bin energy", rel_bin, flush=True)
