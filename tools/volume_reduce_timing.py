"""What BASELINE config 5's grid reduction costs ONE rank of an N-rank job, measured on one GPU:
the rank's 10 GB grid after 1e8 / N histories, the compaction of the other owners' frames into
(index, count) pairs (r3d_volume_compact), and the add of as many pairs as the rank would receive
(r3d_volume_scatter_add: its own pairs of another frame range stand in for its peers').
    python tools/volume_reduce_timing.py [N=8] [toa_degree=9]
Prints occupancy, pairs, bytes on the wire (sparse and dense), and the kernels' times from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radiative3d_amd import Model, Engine, _ffi
from radiative3d_amd.configs import crustpinch_vids, CRUSTPINCH_VOLUME
from radiative3d_amd.parallel import DeviceVolume, DeviceResult

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 9
n = 100_000_000 // world
m = Model(crustpinch_vids(deg) + ["--device-tables"]); e = Engine(m)
vol = DeviceVolume(e, device="cuda:0", **CRUSTPINCH_VOLUME)
res = DeviceResult(m, "cuda:0")
e.run_device(n, 0, 0x5EED, *res.pointers()); torch.cuda.synchronize()
run_ms = e.last_kernel_ms()
lib = _ffi.hip_lib()
cells = vol.counters.numel()
cap = cells // 16
pairs = torch.empty((cap, 2), dtype=torch.int32, device="cuda:0")
n_dev = torch.zeros(1, dtype=torch.int64, device="cuda:0")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
stream = torch.cuda.current_stream().cuda_stream

def compact_others(rank):
    n_dev.zero_()
    for owner in range(world):
        if owner != rank:
            for b, e_ in vol._segments(*vol.frame_range(owner, world)):
                assert lib.r3d_volume_compact(0, vol.counters.data_ptr(), b, e_, pairs.data_ptr(), cap, n_dev.data_ptr(), stream) == 0

compact_others(0); torch.cuda.synchronize()      # warm
ev[0].record(); compact_others(0); ev[1].record(); torch.cuda.synchronize()
sent = int(n_dev.item())
assert sent <= cap, (sent, cap)
scanned = cells - sum(b_e[1] - b_e[0] for b_e in vol._segments(*vol.frame_range(0, world)))
compact_ms = ev[0].elapsed_time(ev[1])
# the add: `sent` pairs (what N - 1 peers with the same occupancy would send) into another grid of the same size
target = torch.zeros_like(vol.counters)
flags = torch.zeros(2, dtype=torch.int64, device="cuda:0")
lib.r3d_volume_scatter_add(0, target.data_ptr(), cells, pairs.data_ptr(), sent, flags.data_ptr(), stream); torch.cuda.synchronize()
ev[2].record()
lib.r3d_volume_scatter_add(0, target.data_ptr(), cells, pairs.data_ptr(), sent, flags.data_ptr(), stream)
ev[3].record(); torch.cuda.synchronize()
add_ms = ev[2].elapsed_time(ev[3])
nonzero = int((vol.counters != 0).sum().item())
events = vol.total()
print(f"config 5, rank of {world}: {n} histories in {run_ms:.1f} ms, {events} events binned ({events / n:.2f} per history), "
      f"{nonzero} of {cells} cells non-zero ({100.0 * nonzero / cells:.2f} %)")
print(f"  sparse by frame: {sent} pairs = {8 * sent / 1e9:.3f} GB sent per rank (dense by frame: {4 * scanned / 1e9:.2f} GB, "
      f"all-reduce: {2 * 4 * scanned / 1e9:.2f} GB)")
print(f"  r3d_volume_compact over {4 * scanned / 1e9:.2f} GB: {compact_ms:.2f} ms = {4 * scanned / compact_ms / 1e6:.0f} GB/s read; "
      f"r3d_volume_scatter_add of {sent} pairs: {add_ms:.2f} ms = {sent / add_ms / 1e6:.2f} G pairs/s")
vol.detach(); e.close()
