"""Where the time of the streaming feed goes (run on the GPU box): the legs of quantity_estimate._LevelStreamer alone."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

n_l, chunk, L = 10_000_000, 100_000, 3
levels = [np.random.default_rng(l).standard_normal((n_l, 2, 1)) for l in range(L)]
total = sum(a.nbytes for a in levels)
specs = [(l, s) for l in range(L) for s in range(0, n_l, chunk)]


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    print("%-58s %7.2f ms  %6.1f GB/s" % (label, 1e3 * best, total / best / 1e9), flush=True)


def par(work, n_threads):
    it = iter(specs)
    lock = threading.Lock()

    def run():
        while True:
            with lock:
                sp = next(it, None)
            if sp is None:
                return
            work(sp)
    ts = [threading.Thread(target=run) for _ in range(n_threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]


pinned = torch.empty(total // 8, dtype=torch.float64, pin_memory=True)
pv = pinned.numpy()
dev = torch.empty(total // 8, dtype=torch.float64, device="cuda")
off = {sp: (sp[0] * n_l + sp[1]) * 2 for sp in specs}

for thr in (1, 2, 4, 8):
    timed("fresh copy of every chunk (the stand-in read), %d thr" % thr, lambda: par(lambda sp: np.array(levels[sp[0]][sp[1]:sp[1] + chunk], copy=True), thr))
for thr in (1, 2, 4, 8):
    timed("chunk view -> pinned staging (read-into), %d thr" % thr,
          lambda: par(lambda sp: np.copyto(pv[off[sp]:off[sp] + 2 * chunk], levels[sp[0]][sp[1]:sp[1] + chunk].reshape(-1)), thr))
for thr in (1, 2, 4, 8):
    timed("fresh copy + copy into pinned staging, %d thr" % thr,
          lambda: par(lambda sp: np.copyto(pv[off[sp]:off[sp] + 2 * chunk], np.array(levels[sp[0]][sp[1]:sp[1] + chunk], copy=True).reshape(-1)), thr))


def dma(block_mb):
    nb = block_mb * 2 ** 20 // 8
    for p in range(0, total // 8, nb):
        dev[p:p + nb].copy_(pinned[p:p + nb], non_blocking=True)
    torch.cuda.synchronize()


for mb in (2, 32, 160):
    timed("DMA pinned -> HBM in blocks of %d MB" % mb, lambda: dma(mb))
timed("pageable torch .to(device) per level", lambda: [torch.from_numpy(a.reshape(-1)).to("cuda") for a in levels] and torch.cuda.synchronize())
