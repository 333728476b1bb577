"""Where does the time of the staging ring go?  300 chunks of 1.6 MB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mlmc_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
n = 200_000
chunks = [np.random.default_rng(i).standard_normal(n) for i in range(8)]
stream = torch.cuda.Stream(device=dev)
pinned = [torch.empty(n, dtype=torch.float64, pin_memory=True) for _ in range(3)]
events = [torch.cuda.Event() for _ in range(3)]
acc = dict(sync=0.0, host=0.0, alloc=0.0, h2d=0.0, wait=0.0)
outs = []
torch.cuda.synchronize()
T0 = time.perf_counter()
for k in range(300):
    s = k % 3
    t0 = time.perf_counter(); events[s].synchronize(); t1 = time.perf_counter()
    pinned[s].copy_(torch.from_numpy(chunks[k % 8])); t2 = time.perf_counter()
    out = torch.empty(n, dtype=torch.float64, device=dev); t3 = time.perf_counter()
    with torch.cuda.stream(stream):
        out.copy_(pinned[s], non_blocking=True)
        events[s].record(stream)
    t4 = time.perf_counter()
    _lib.check(_lib.lib().mlmc_wait_event(events[s].cuda_event)); t5 = time.perf_counter()
    outs.append(out)
    acc["sync"] += t1 - t0; acc["host"] += t2 - t1; acc["alloc"] += t3 - t2; acc["h2d"] += t4 - t3; acc["wait"] += t5 - t4
_lib.check(_lib.lib().mlmc_synchronize())
T1 = time.perf_counter()
print("ring: total %.1f ms" % (1e3 * (T1 - T0)), {k: round(1e3 * v, 2) for k, v in acc.items()})
# plain synchronous pageable copies
outs = []
torch.cuda.synchronize()
T0 = time.perf_counter()
for k in range(300):
    outs.append(torch.from_numpy(chunks[k % 8]).to(dev))
torch.cuda.synchronize()
print("pageable .to(): total %.1f ms" % (1e3 * (time.perf_counter() - T0)))
# pinned + non_blocking on the current stream
outs = []
torch.cuda.synchronize()
T0 = time.perf_counter()
for k in range(300):
    s = k % 3
    events[s].synchronize()
    pinned[s].copy_(torch.from_numpy(chunks[k % 8]))
    out = torch.empty(n, dtype=torch.float64, device=dev)
    out.copy_(pinned[s], non_blocking=True)
    events[s].record()
    outs.append(out)
torch.cuda.synchronize()
print("pinned, current stream: total %.1f ms" % (1e3 * (time.perf_counter() - T0)))
# host copy alone
T0 = time.perf_counter()
for k in range(300):
    pinned[k % 3].copy_(torch.from_numpy(chunks[k % 8]))
print("host copy into pinned alone: %.1f ms" % (1e3 * (time.perf_counter() - T0)))
tmp = torch.empty(n, dtype=torch.float64)
T0 = time.perf_counter()
for k in range(300):
    tmp.copy_(torch.from_numpy(chunks[k % 8]))
print("host copy into pageable alone: %.1f ms" % (1e3 * (time.perf_counter() - T0)))
T0 = time.perf_counter()
for k in range(300):
    np.copyto(pinned[k % 3].numpy(), chunks[k % 8])
print("np.copyto into pinned: %.1f ms" % (1e3 * (time.perf_counter() - T0)))
