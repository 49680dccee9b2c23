"""Soak test of the pool's queue logic (drain with kept lanes, refill in pairs, carry chains): many launches of
random sizes on small-table models of all three cell kinds; every launch must account for its histories
(generated == n, lost + timeout + invalid == n) and a chain must equal one run of its ids.
    timeout -k 10 300 python tools/soak.py [launches=400]"""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiative3d_amd import Model, Engine
from radiative3d_amd.parallel import DeviceResult
from radiative3d_amd.configs import CONFIGS
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rnd = random.Random(4)
t0 = time.time()
for name in ("crustpinch", "lopnor", "sphere_deep", "halfspace", "crustpinch_vids"):
    m = Model(CONFIGS[name](4))
    for opts in ({}, {"pool_slots": 768, "accumulator_bits": 0}):
        e = Engine(m, **opts)
        total = 0
        for i in range(launches):
            n = rnd.choice([0, 1, 63, 64, 65, 127, 128, 129, 767, 1024, 5000, rnd.randint(1, 300000)])
            r = e.run(n, first_id=rnd.randint(0, 1 << 40), seed=rnd.randint(0, 1 << 30))
            assert r.events["generated"] == n and r.n_lost + r.n_timeout + r.n_invalid == n, (name, opts, i, n)
            total += n
        # a chain of uneven steps against one run of the same ids
        buf = DeviceResult(m, "cuda:0")
        sizes = [rnd.randint(0, 200000) for _ in range(40)]
        first = 123456
        for s in sizes:
            e.run_device(s, first, 77, *buf.pointers(), carry="carry")
            first += s
        e.run_device(0, 0, 77, *buf.pointers(), carry="final")
        torch.cuda.synchronize()
        got, want = buf.to_result(), e.run(sum(sizes), first_id=123456, seed=77)
        assert (got.counts == want.counts).all() and got.events == want.events, (name, opts)
        assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid)
        e.close()
        print(f"{name} {opts or 'default'}: {launches} launches, {total} histories, chain of {len(sizes)} = one run; {time.time() - t0:.0f} s", flush=True)
print("soak ok")
