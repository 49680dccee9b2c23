"""The bench's envelope statistic between two INDEPENDENT GPU samples with the bench's batch
structure (16 x 376k histories against 32 x 1.25M): what the estimator itself gives when both
sides come from the same code."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from radiative3d_amd import Model, Engine
from radiative3d_amd.configs import crustpinch
from bench import envelope_agreement, batch_moments
m = Model(crustpinch(9) + ["--device-tables"]); e = Engine(m, lib=os.environ.get("R3D_HIP_LIB"))
def batches(k, per, base):
    es, cs = [], []
    for b in range(k):
        r = e.run(per, first_id=base + b * per)
        es.append(r.energy / per); cs.append(r.counts)
    return batch_moments(es, cs)
small = batches(16, 376000, 1 << 50)
big = batches(32, 1250000, 1 << 44)
print("GPU 32 x 1.25M vs GPU 16 x 376k:", envelope_agreement(big, small, 32 * 1250000, 16 * 376000)["rms_sigma"])
small2 = batches(16, 376000, 1 << 52)
print("GPU 16 x 376k vs GPU 16 x 376k:", envelope_agreement(small2, small, 16 * 376000, 16 * 376000)["rms_sigma"])
