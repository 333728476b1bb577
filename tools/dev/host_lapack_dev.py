"""Dev: host LAPACK calls of the PDF chain on the GPU box's CPU (thread-pool effects on tiny matrices)."""
import time, os, numpy as np, scipy.linalg
rng = np.random.default_rng(0)
for R in (26, 50, 64):
    a = rng.normal(size=(R, R)); h = a @ a.T + R * np.eye(R)
    for name, fn in (("eigvalsh", lambda: np.linalg.eigvalsh(h)), ("eigh", lambda: np.linalg.eigh(h)), ("rq", lambda: scipy.linalg.rq(a))):
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
        ts = np.array(ts)
        print(f"R {R} {name:9s} min {ts.min():.3f} median {np.median(ts):.3f} max {ts.max():.3f} ms")
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    from threadpoolctl import threadpool_info
    for i in threadpool_info(): print(i.get("internal_api"), i.get("num_threads"), i.get("threading_layer"))
except Exception as e:
    print("threadpoolctl:", e)
