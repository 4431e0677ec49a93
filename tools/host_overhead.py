#!/usr/bin/env python3
"""Host (Python + launch) time per train_step vs GPU time: is the step launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd.deepfm import DeepFMLayer
dev = "cuda"
B, S, Dn, D, NT = 65536, 26, 13, 16, 1_000_000
so = torch.arange(S, dtype=torch.int64, device=dev) * NT
sharded = len(sys.argv) > 1 and sys.argv[1] == "sharded"
if sharded:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev, 0))
    from paddlerec_amd.sharded import ShardedDeepFMLayer
    m = ShardedDeepFMLayer(NT * S, D, Dn, S, [400, 400, 400], device=dev, slot_offset=so)
else:
    m = DeepFMLayer(NT * S, D, Dn, S, [400, 400, 400], device=dev, slot_offset=so)
g = torch.Generator(device=dev).manual_seed(1)
bat = []
for _ in range(4):
    ids = torch.randint(1, NT, (B, S), device=dev, generator=g)
    bat.append((ids, torch.rand(B, Dn, device=dev, generator=g), (torch.rand(B, 1, device=dev, generator=g) < 0.25).long()))
def step(i):
    ids, de, la = bat[i % 4]
    if sharded:
        return m.train_step(ids, de, la, next_sparse_inputs=bat[(i + 1) % 4][0])
    return m.train_step(ids, de, la)
for i in range(5): step(i)
torch.cuda.synchronize()
host = 0.0
t0 = time.perf_counter()
for i in range(20):
    a = time.perf_counter(); step(i); host += time.perf_counter() - a
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("%s: host issue %.2f ms/step, wall %.2f ms/step" % ("sharded" if sharded else "single", host / 20 * 1e3, tot / 20 * 1e3))
