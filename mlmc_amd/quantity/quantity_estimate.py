"""MLMC mean / variance estimation of quantities (reference interface: mlmc/quantity/quantity_estimate.py:6-156).

`estimate_mean`, `moments`, `moment`, `covariance`, `mask_nan_samples`, `cache_clear` keep the reference's names,
arguments, return type (`QuantityMean`) and error behaviour.  What differs is where the work happens: for every
storage chunk the raw fine / coarse arrays [M, n] go to the MI355X once; the moment functions, the NaN /
out-of-domain mask, the level differences and the sums of quantity_estimate.py:43-65 are fused into the HIP
accumulation kernels (mlmc_amd/csrc/moments.hip, cov.hip).  Nothing of shape [M*R, n, 2] or [M*R*R, n, 2] is
ever materialised.
"""
import collections
import os
import threading

import numpy as np

from . import lowering
from . import quantity as qmod
from . import quantity_types as qt
from .. import engine
from .. import linearize


def mask_nan_samples(chunk):
    """Drop samples with a NaN in fine or coarse (reference: quantity_estimate.py:6-14).  Host helper kept for API
    compatibility; the estimators apply the same rule on the device."""
    mask = np.any(np.isnan(chunk), axis=0).any(axis=1)
    return chunk[..., ~mask, :], np.count_nonzero(mask)


def cache_clear():
    qmod.cache_clear()


class _MomentsNode(qmod.Quantity):
    """Quantity node 'moments of x'; evaluated lazily on the device when somebody asks for its samples, fused into
    the accumulation kernel when it is the root of an estimate."""

    def __init__(self, quantity, moments_fn, at_bottom, qtype):
        self._moments_fn = moments_fn
        self._at_bottom = at_bottom
        super().__init__(quantity_type=qtype, input_quantities=[quantity], operation=self._eval)

    def _eval(self, x):
        mom = self._moments_fn.eval_all(x)                      # [M, n, 2, R] (device evaluated)
        mom = mom.transpose((0, 3, 1, 2)) if self._at_bottom else mom.transpose((3, 0, 1, 2))
        return mom.reshape((int(np.prod(mom.shape[:-2])), mom.shape[-2], mom.shape[-1]))


class _CovarianceNode(qmod.Quantity):
    def __init__(self, quantity, moments_fn, at_bottom, qtype):
        self._moments_fn = moments_fn
        self._at_bottom = at_bottom
        super().__init__(quantity_type=qtype, input_quantities=[quantity], operation=self._eval)

    def _eval(self, x):
        # The reference materialises [M*R*R, n, 2] doubles here (quantity_estimate.py:131-147).  On this path the
        # outer products only ever exist inside the MFMA covariance kernel; there is no host fallback.
        raise NotImplementedError("samples of a covariance quantity are not materialised; use estimate_mean(covariance(...))")


def moment(quantity, moments_fn, i=0):
    """Quantity of the i-th moment function of `quantity` (reference: quantity_estimate.py:83-93)."""
    return qmod.Quantity(quantity_type=quantity.qtype, input_quantities=[quantity],
                         operation=lambda x: moments_fn.eval_single_moment(i, value=x))


def moments(quantity, moments_fn, mom_at_bottom=True):
    """Quantity of all moment functions (reference: quantity_estimate.py:96-119)."""
    if mom_at_bottom:
        qtype = quantity.qtype.replace_scalar(qt.ArrayType(shape=(moments_fn.size,), qtype=qt.ScalarType()))
    else:
        qtype = qt.ArrayType(shape=(moments_fn.size,), qtype=quantity.qtype)
    return _MomentsNode(quantity, moments_fn, mom_at_bottom, qtype)


def covariance(quantity, moments_fn, cov_at_bottom=True):
    """Quantity of the moment outer products (reference: quantity_estimate.py:122-156)."""
    shape = (moments_fn.size, moments_fn.size)
    if cov_at_bottom:
        qtype = quantity.qtype.replace_scalar(qt.ArrayType(shape=shape, qtype=qt.ScalarType()))
    else:
        qtype = qt.ArrayType(shape=shape, qtype=quantity.qtype)
    return _CovarianceNode(quantity, moments_fn, cov_at_bottom, qtype)


class _DeviceChunkCache:
    """HBM-resident copies of the raw sample chunks an estimate has consumed (LRU, bounded).

    An MLMC post-processing session runs several passes over the same samples (moments, covariance, level variances,
    the two passes of construct_density); the first pass uploads each chunk once, later passes -- and later estimates of
    the same quantity -- read it from HBM, where a whole multi-level estimate is one kernel launch.  288 GB of HBM3E
    hold 1.8e10 sample pairs; the default budget is MLMC_HIP_DEVICE_CACHE_GB = 64.
    Keyed by (source quantity, level, chunk id, chunk slice, stamp of the level = samples collected + the storage's
    modification count where it keeps one, _level_stamps): storages are append-only, so a level that has grown misses and
    is uploaded again.  `device_cache_clear()` drops everything."""

    def __init__(self):
        self._items = collections.OrderedDict()
        self._bytes = 0
        self.uploads = 0
        self.hits = 0
        # Every method is a multi-step update of (_items, _bytes): `get` alone is lookup + move_to_end, and a clear() of another
        # thread between the two raises KeyError in the looking thread (the round-2 "hang" of
        # test_concurrent_estimates_through_the_python_api).  Callers hold _estimate_lock as well; the
        # cache does not rely on that.
        self._lock = threading.RLock()

    @staticmethod
    def budget():
        return int(float(os.environ.get("MLMC_HIP_DEVICE_CACHE_GB", "64")) * 2 ** 30)

    @staticmethod
    def _owner_ref(owner):
        """The cache must not keep a storage (and its host arrays) alive: entries hold a WEAK reference to their owner and
        count only while it lives -- a dead owner's id may be recycled, a live one's cannot.  (Objects that cannot be weakly
        referenced are held strongly, as before.)"""
        if owner is None:
            return None
        import weakref
        try:
            return weakref.ref(owner)
        except TypeError:
            return lambda owner=owner: owner

    @staticmethod
    def _alive(item):
        return item[3] is None or item[3]() is not None

    def get(self, key):
        with self._lock:
            item = self._items.get(key)
            if item is not None and not self._alive(item):
                self.drop(key)                                   # its storage is gone: the rows are garbage now
                item = None
            if item is not None:
                self._items.move_to_end(key)
                self.hits += 1
            return item

    def __contains__(self, key):
        with self._lock:
            item = self._items.get(key)
            return item is not None and self._alive(item)

    def put(self, key, fine, coarse, owner=None):
        nbytes = fine.nbytes + (0 if coarse is None else coarse.nbytes)
        if nbytes > self.budget():
            return None
        item = (_upload(fine, count=False), None if coarse is None else _upload(coarse, count=False), nbytes, self._owner_ref(owner))
        with self._lock:
            self._insert(key, item, nbytes)
            self.uploads += 1
        return item

    def put_tensors(self, key, fine, coarse, owner=None):
        """Cache tensors that already live on the device (results of a lowered quantity, stored rows)."""
        nbytes = fine.numel() * 8 + (0 if coarse is None else coarse.numel() * 8)
        item = (fine, coarse, nbytes, self._owner_ref(owner))
        if nbytes > self.budget():
            return item
        with self._lock:
            self._insert(key, item, nbytes)
        return item

    def _insert(self, key, item, nbytes):
        self.drop(key)                                   # a replaced entry gives its bytes back first
        self._inserts = getattr(self, "_inserts", 0) + 1
        if self._inserts % 64 == 0 or self._bytes + nbytes > self.budget():
            self.drop_dead()                             # rows of storages that no longer exist
        while self._bytes + nbytes > self.budget() and self._items:
            _, (_, _, old, _) = self._items.popitem(last=False)
            self._bytes -= old
        self._items[key] = item
        self._bytes += nbytes

    def drop(self, key):
        with self._lock:
            item = self._items.pop(key, None)
            if item is not None:
                self._bytes -= item[2]

    def drop_dead(self):
        with self._lock:
            for key in [k for k, item in self._items.items() if not self._alive(item)]:
                self.drop(key)

    def drop_owner(self, owner):
        with self._lock:
            for key in [k for k, item in self._items.items() if item[3] is not None and item[3]() is owner]:
                self.drop(key)

    def clear(self):
        with self._lock:
            self._items.clear()
            self._bytes = 0


def _lib_device():
    from .. import _lib
    _lib.lib()
    return _lib._bound_device


_device_cache = _DeviceChunkCache()
# One estimate at a time: one GPU stream, one sample cache, one memo of chunk evaluations shared by all quantities.
_estimate_lock = threading.RLock()


_cache_generation = 0


def device_cache_generation():
    """Counts device_cache_clear() calls: results derived from resident samples (Estimate's kept covariance sums) are
    valid for one generation."""
    return _cache_generation


def device_cache_clear():
    """Drop the HBM-resident sample chunks (call after modifying stored samples in place).  Waits for a running estimate of
    another thread: its kernels may still read the resident rows."""
    global _cache_generation
    with _estimate_lock:
        _cache_generation += 1
        _device_cache.clear()
        _block_meta.clear()


def device_cache_drop_owner(storage):
    """Drop the resident chunks that came from one storage (it changed: grew, was refilled)."""
    with _estimate_lock:
        _device_cache.drop_owner(storage)
        _block_meta.drop_owner(storage)


class _LevelStreamer:
    """Streaming feed of a level that the storage delivers in many chunks (SampleStorageHDF reads one
    `collected_values[chunk_slice]` per chunk and reopens the file every time, mlmc/tool/hdf5.py:353-376,
    sample_storage_hdf.py:169-184): the chunks -- [n][2][M] record arrays, the layout of the reference's Memory storage and of
    HDF5 `collected_values` -- are read by several helper threads straight into PINNED staging blocks of
    MLMC_HIP_STREAM_BLOCK_MB (default 32), a block goes to HBM as ONE asynchronous DMA at link speed while the helpers fill
    the next blocks, and the whole level lands in ONE device tensor in the storage's own layout: the quantity tree then runs
    as one k_expr launch per level (strided LOAD: fine / coarse de-interleaving and the row gather on the device), every
    other quantity of the analysis finds the level resident.  Reads (h5py, NumPy copies) and the copies into the staging
    blocks release the GIL, so the helpers really run side by side; per chunk the main thread does nothing.
    MLMC_HIP_STREAM_THREADS (default 4: more helpers contend for the page-fault lock and the GIL and lose) helpers; MLMC_HIP_STREAM_UPLOAD=0 switches the feed off (chunk by chunk, synchronously).

    (Round 2 had a read-ahead thread that handed single chunks to the main thread -- one pageable copy, one k_expr launch and
    ~100 us of Python per 1.6 MB chunk: 16 GB/s -- and an opt-in ring of chunk-sized pinned buffers whose per-chunk
    asynchronous copy + event cost as much host time as the synchronous copy: 0.9 GB/s.  Both are gone.)"""
    N_BUF = 3

    def __init__(self):
        self._pinned = [None] * self.N_BUF
        self._stream = None
        self.blocks = 0            # staging blocks sent (tests)
        self.levels = 0

    @staticmethod
    def enabled():
        return os.environ.get("MLMC_HIP_STREAM_UPLOAD", "1") != "0"

    @staticmethod
    def block_doubles():
        return max(int(float(os.environ.get("MLMC_HIP_STREAM_BLOCK_MB", "32")) * 2 ** 20) // 8, 1)

    @staticmethod
    def n_threads():
        return max(int(os.environ.get("MLMC_HIP_STREAM_THREADS", "4")), 1)

    def _buffer(self, slot, doubles):
        import torch
        buf = self._pinned[slot]
        if buf is None or buf.numel() < doubles:
            buf = self._pinned[slot] = torch.empty(doubles, dtype=torch.float64, pin_memory=True)
        return buf

    def stream(self, leaf, levels, n_rows_read, budget_bytes):
        """levels: [(level id, [chunk specs of the level])].  Generator of (level id, block | None, first chunk | None) in the
        order given: block = (flat device tensor, sample stride, side stride, n, width) once the level has landed; None when
        the level does not qualify (unknown chunk sizes, not record arrays, too large) -- the caller then takes its chunks one
        by one and must not read the first one again (it is handed back).  The levels are ONE pipeline: while the last block of
        a level is on its way (and the caller queues the level's kernels) the helpers already fill the next level's blocks."""
        import torch
        plans = []                      # per qualifying level: dict(level, specs, sizes, sn, sw, m_total, width, raw0, out, blocks...)
        early = []
        for level_id, specs in levels:
            sizes = []
            for cs in specs:
                sl = cs.chunk_slice
                if sl is None or sl.start is None or sl.stop is None or sl.step not in (None, 1):
                    sizes = None
                    break
                sizes.append(sl.stop - sl.start)
            if sizes is None:
                early.append((level_id, None, None))
                continue
            raw0 = leaf.samples(specs[0])
            # (a one-row storage whose chunk already is an [n][2] row counts as a record array with M = 1 here)
            layout = _block_layout(raw0, n_rows_read, True) if raw0.ndim == 3 and raw0.shape[1] == sizes[0] else None
            n_total = sum(sizes)
            if layout is None or n_total * layout[1] * 8 > budget_bytes or max(sizes) * layout[1] > 16 * self.block_doubles():
                early.append((level_id, None, raw0))
                continue
            plans.append(dict(level=level_id, specs=specs, sizes=sizes, sn=layout[1], sw=layout[2], m_total=raw0.shape[0],
                              width=raw0.shape[2], raw0=raw0, n_total=n_total))
        for item in early:
            yield item
        if not plans:
            return
        dev = torch.device("cuda", _lib_device())
        cap = max([self.block_doubles()] + [max(p["sizes"]) * p["sn"] for p in plans])
        # global lists: tasks (one per chunk, in order) and staging blocks (consecutive chunks of ONE level)
        tasks, blocks = [], []          # task: (plan index, chunk index, block index, offset in block); block: [plan index, length, tasks left, offset in level]
        for pi, p in enumerate(plans):
            p["out"] = torch.empty(p["n_total"] * p["sn"], dtype=torch.float64, device=dev)
            cur_len, pos = 0, 0
            blocks.append([pi, 0, 0, 0])
            for ci, n_c in enumerate(p["sizes"]):
                need = n_c * p["sn"]
                if cur_len and cur_len + need > cap:
                    pos += cur_len
                    blocks.append([pi, 0, 0, pos])
                    cur_len = 0
                tasks.append((pi, ci, len(blocks) - 1, cur_len))
                cur_len += need
                blocks[-1][1] = cur_len
                blocks[-1][2] += 1
            p["last_block"] = len(blocks) - 1
        views = [self._buffer(k, cap).numpy() for k in range(min(self.N_BUF, len(blocks)))]
        # storages that can write a chunk's [n][2][M] records straight into a caller buffer (sample_storage.Memory
        # .sample_records_into; an HDF5 storage: Dataset.read_direct) skip the intermediate array: one host copy, no fresh pages
        into = getattr(getattr(leaf, "_storage", None), "sample_records_into", None)
        if os.environ.get("MLMC_HIP_STREAM_READ_INTO", "1") == "0":
            into = None
        for p in plans:
            p["into"] = into is not None and p["sn"] == 2 * p["m_total"] and (p["width"] == 1 or p["sw"] == p["m_total"])
        cond = threading.Condition()
        state = {"next": 0, "released": 0, "error": None}
        ready = [threading.Event() for _ in blocks]

        def worker():
            while True:
                with cond:
                    if state["error"] is not None or state["next"] >= len(tasks):
                        return
                    pi, ci, b, off = tasks[state["next"]]
                    state["next"] += 1
                    while b >= state["released"] + self.N_BUF and state["error"] is None:   # its slot still holds an older block
                        cond.wait(0.05)
                    if state["error"] is not None:
                        return
                try:
                    p = plans[pi]
                    if p["into"] and ci > 0:                     # (chunk 0 was read as an array to learn the layout)
                        n_c = p["sizes"][ci]
                        into(p["specs"][ci], views[b % self.N_BUF][off:off + n_c * p["sn"]].reshape(n_c, 2, p["m_total"]))
                        with cond:
                            blocks[b][2] -= 1
                            done = blocks[b][2] == 0
                        if done:
                            ready[b].set()
                        continue
                    raw = p["raw0"] if ci == 0 else leaf.samples(p["specs"][ci])
                    lay = _block_layout(raw, n_rows_read, True) if raw.ndim == 3 else None
                    if raw.shape != (p["m_total"], p["sizes"][ci], p["width"]) or lay is None or lay[1:] != (p["sn"], p["sw"]):
                        raise ValueError("chunk {} of level {} does not continue the record layout of the level's first chunk"
                                         .format(p["specs"][ci].chunk_id, p["specs"][ci].level_id))
                    span = (p["sizes"][ci] - 1) * p["sn"] + (p["width"] - 1) * p["sw"] + p["m_total"]
                    flat = np.lib.stride_tricks.as_strided(raw, shape=(span,), strides=(8,))
                    np.copyto(views[b % self.N_BUF][off:off + span], flat)
                except BaseException as e:                       # noqa: BLE001 - handed to the consumer
                    with cond:
                        state["error"] = e
                        cond.notify_all()
                    for ev in ready:
                        ev.set()
                    return
                with cond:
                    blocks[b][2] -= 1
                    done = blocks[b][2] == 0
                if done:
                    ready[b].set()

        threads = [threading.Thread(target=worker, name="mlmc-level-reader", daemon=True)
                   for _ in range(min(self.n_threads(), len(tasks)))]
        for t in threads:
            t.start()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        try:
            for b, (pi, length, _, pos) in enumerate(blocks):
                ready[b].wait()
                if state["error"] is not None:
                    raise state["error"]
                p = plans[pi]
                with torch.cuda.stream(self._stream):
                    p["out"][pos:pos + length].copy_(self._pinned[b % self.N_BUF][:length], non_blocking=True)
                self._stream.synchronize()                       # the helpers fill the other slots meanwhile
                with cond:
                    state["released"] = b + 1
                    cond.notify_all()
                self.blocks += 1
                if b == p["last_block"]:
                    self.levels += 1
                    yield p["level"], (p["out"], p["sn"], p["sw"], p["n_total"], p["width"]), None
        except BaseException:
            with cond:
                if state["error"] is None:
                    state["error"] = RuntimeError("level stream aborted")
                cond.notify_all()
            raise
        finally:
            for t in threads:
                t.join()


_streamer = _LevelStreamer()


def _upload(host_array, count=True):
    """Host array (any shape, float64, C-contiguous) -> device tensor of the same shape, as one synchronous pageable copy (a
    whole level in one chunk goes up at link speed that way; many-chunk levels take the _LevelStreamer)."""
    import torch
    a = np.ascontiguousarray(host_array, dtype=np.float64)
    if count:
        _device_cache.uploads += 1
    dev = torch.device("cuda", _lib_device())
    t = torch.from_numpy(a).to(dev)
    torch.cuda.current_stream(dev).synchronize()                 # the library reads it on its own stream
    return t


class _ChunkPrefetcher:
    """Read-ahead for the chunks the _LevelStreamer does not take (levels whose chunks are not record arrays -- their rows go
    up one by one --, partly resident levels, chunk specs without slices): a helper thread reads the chunks the estimate is
    going to miss, in order, up to `depth` ahead of the consumer, while the main thread uploads the previous chunk and queues
    its kernels; per chunk the feed costs max(read, upload) instead of their sum.  MLMC_HIP_STREAM_UPLOAD=0 switches it off."""

    def __init__(self, leaf, specs, depth=2):
        import queue
        self._leaf = leaf
        self._specs = list(specs)
        self._queue = queue.Queue(maxsize=depth)
        self._stop = False
        self._thread = threading.Thread(target=self._run, name="mlmc-chunk-reader", daemon=True)
        self._thread.start()

    @staticmethod
    def enabled():
        return os.environ.get("MLMC_HIP_STREAM_UPLOAD", "1") != "0"

    def _run(self):
        try:
            for spec in self._specs:
                if self._stop:
                    return
                self._queue.put((spec, self._leaf.samples(spec), None))
        except BaseException as e:                                # noqa: BLE001 - handed to the consumer
            self._queue.put((None, None, e))

    def get(self, spec):
        """The chunk of `spec`; specs must be asked for in the order they were given."""
        got, raw, err = self._queue.get()
        if err is not None:
            raise err
        assert got is spec or (got.level_id, got.chunk_id) == (spec.level_id, spec.chunk_id)
        return raw

    def close(self):
        self._stop = True
        try:
            while True:
                self._queue.get_nowait()
        except Exception:
            pass


def _split_fine_coarse(chunk, level_id):
    """[M, n, 2|1] -> contiguous fine[M, n], coarse[M, n] | None (level 0 carries no coarse samples)."""
    fine = np.ascontiguousarray(chunk[:, :, 0], dtype=np.float64)
    if level_id == 0 or chunk.shape[-1] == 1:
        return fine, None
    return fine, np.ascontiguousarray(chunk[:, :, 1], dtype=np.float64)


class _AccumulatorPool:
    """Finished LevelAccumulators, reused by the next estimate with the same (moment functions, levels, mode,
    components): creating one allocates device state, scratch and a pinned mirror (~0.3 ms), an estimate over resident
    samples takes about as long."""

    def __init__(self, capacity=8):
        self._items = collections.OrderedDict()
        self._capacity = capacity

    def take(self, fn, n_levels, mode, n_comp, mean_only=False):
        key = (id(fn), n_levels, mode, n_comp, bool(mean_only))
        item = self._items.pop(key, None)
        if item is not None:
            acc = item[0]
            acc.reset()
            return acc
        return engine.LevelAccumulator(fn, n_levels, mode, n_comp=n_comp, mean_only=mean_only)

    def give(self, fn, n_levels, mode, n_comp, acc, mean_only=False):
        self._items[(id(fn), n_levels, mode, n_comp, bool(mean_only))] = (acc, fn)     # fn kept alive: its id stays unique
        while len(self._items) > self._capacity:
            _, (old, _) = self._items.popitem(last=False)
            old.close()

    def clear(self):
        for acc, _ in self._items.values():
            acc.close()
        self._items.clear()


_acc_pool = _AccumulatorPool()


def _device_tree_enabled():
    return os.environ.get("MLMC_HIP_DEVICE_TREE", "1") != "0"


def _owner_of(plan):
    """The object whose lifetime bounds the resident rows / blocks of a lowered tree: the storage behind the leaf, or the
    leaf itself when it has none.  ONE rule for the upload paths, the residency check of the prefetcher and the cache keys."""
    storage = getattr(plan.leaf, "_storage", None)
    return storage if storage is not None else plan.leaf


def _stored_row_on_device(plan, chunk_spec, chunk_key, stored_row, use_cache, raw=None):
    """One stored row of one chunk as a device tensor in the storage's own layout: interleaved (fine, coarse) pairs
    [n, 2], or [n, 1] at level 0.  Uploaded once per (storage, chunk, row) and shared by every quantity that reads it."""
    import torch
    storage = getattr(plan.leaf, "_storage", None)
    owner = _owner_of(plan)
    key = ("row", id(owner)) + chunk_key + (stored_row,)
    item = _device_cache.get(key) if use_cache else None
    if item is not None:
        return item[0]
    if hasattr(storage, "device_row"):                            # samples that already live in HBM (sim/synth_device.py)
        t = storage.device_row(chunk_spec, stored_row)
    else:
        if raw is None:
            raw = plan.leaf.samples(chunk_spec)                   # [M_stored, n, 2|1]
        t = _upload(raw[stored_row])
    if use_cache:
        _device_cache.put_tensors(key, t, None, owner=owner)
    return t


def _block_layout(raw, n_rows_read, ready_rows_too=False):
    """Can the chunk view raw [M_stored, n, 2|1] go to the device as one flat copy of the storage's [n][sides][M] records?
    -> (span in doubles from the first to the last value, sample stride, side stride) or None (rows are uploaded one by
    one: they already are contiguous [n][2] / [n] rows, the tree reads less than 1/8 of a wide record, or the view is not
    a record array).  Pure layout arithmetic on shape and strides."""
    m_total, n, width = raw.shape
    if n == 0 or raw.dtype != np.float64 or any(st % 8 for st in raw.strides):
        return None
    sm, sn, sw = (st // 8 for st in raw.strides)
    if m_total == 1:
        sm = 1                                                   # the stride of a length-1 axis carries no meaning
    if width == 1:
        sw = 0
    rows_are_ready = (sn == width and (width == 1 or sw == 1))   # raw[m] already is an [n][2] / [n] row
    if (rows_are_ready and not ready_rows_too) or (m_total > 1 and n_rows_read * 8 < m_total):
        return None
    if sm != 1 or sn < m_total * width or (width == 2 and sw < m_total):
        return None                                              # not an [n][sides][M] record array
    span = (n - 1) * sn + (width - 1) * sw + m_total              # doubles between the first and the last value
    if span > 3 * raw.size:
        return None
    return span, sn, sw


class _BlockMeta:
    """(block key [, rows the tree reads]) -> (sample stride, side stride, n, width) of a resident block, or None when the
    chunk is uploaded row by row: a later estimate decides without reading the chunk from the storage again.  The keys
    carry id(owner); every entry also holds a weak reference to that owner and counts only while it is the same live
    object -- a new storage that happens to get the id of a freed one inherits nothing."""
    _MISSING = object()

    def __init__(self):
        self._d = {}

    @staticmethod
    def _ref(owner):
        import weakref
        try:
            return weakref.ref(owner)
        except TypeError:                     # not weak-referenceable: keep it alive, its id then stays unique
            return lambda owner=owner: owner

    def get(self, key, owner, default=_MISSING):
        entry = self._d.get(key)
        if entry is None or entry[0]() is not owner:
            if entry is not None:
                del self._d[key]
            return default
        return entry[1]

    def has(self, key, owner):
        return self.get(key, owner) is not self._MISSING

    def put(self, key, owner, value):
        if len(self._d) > 65536:
            self._d.clear()
        self._d[key] = (self._ref(owner), value)

    def drop_owner(self, owner):
        oid = id(owner)
        for key in [k for k in self._d if k[1] == oid]:
            del self._d[key]

    def clear(self):
        self._d.clear()

    def __len__(self):
        return len(self._d)


_block_meta = _BlockMeta()


def _stored_block_on_device(plan, chunk_spec, chunk_key, use_cache, raw=None):
    """The whole stored chunk as ONE device buffer in the storage's own [n][2][M] layout (reference Memory storage and
    HDF5 `collected_values`, sample_storage.py:169-184), when the host view allows it: -> (flat device tensor,
    sample stride, side stride, n, width) or None.  One contiguous PCIe copy replaces one strided host gather per
    stored row (a row of an [n][2][M] array touches a separate cache line per value once M >= 8); k_expr de-interleaves
    while loading (strided LOAD).  Taken when the tree reads at least 1/8 of the stored rows -- and for M = 1 at level 0,
    where the storage keeps an unused coarse column ([n][2], the fine values at stride 2): copying 2x the bytes at link
    speed beats a strided host gather of half of them (2.8 ms vs 8 ms + 1.4 ms for 10^7 samples)."""
    import torch
    storage = getattr(plan.leaf, "_storage", None)
    if hasattr(storage, "device_row") or os.environ.get("MLMC_HIP_BLOCK_UPLOAD", "1") == "0":
        return None
    owner = _owner_of(plan)
    key = ("block", id(owner)) + chunk_key
    item = _device_cache.get(key) if use_cache else None
    if use_cache and (_block_meta.has(key + ("any",), owner) or _block_meta.has(key + (len(plan.in_rows),), owner)):
        return None                                               # known: the chunk goes up row by row (for any / this many rows)
    if item is not None and _block_meta.has(key, owner):
        return (item[0],) + _block_meta.get(key, owner)           # a resident block serves every tree
    if raw is None:
        raw = plan.leaf.samples(chunk_spec)                       # [M_stored, n, 2|1] view of the storage / fresh read
    layout = _block_layout(raw, len(plan.in_rows))
    if layout is None or layout[0] * 8 > _DeviceChunkCache.budget() // 4:
        if use_cache:                                             # not a record array at all, or too few of its rows are read
            intrinsic = layout is not None or _block_layout(raw, raw.shape[0]) is None
            _block_meta.put(key + (("any",) if intrinsic else (len(plan.in_rows),)), owner, None)
        return None
    span, sn, sw = layout
    m_total, n, width = raw.shape
    if item is None:
        flat = np.lib.stride_tricks.as_strided(raw, shape=(span,), strides=(8,))
        t = _upload(flat)
        if use_cache:
            _device_cache.put_tensors(key, t, None, owner=owner)
            _block_meta.put(key, owner, (sn, sw, n, width))
    else:
        t = item[0]
    return t, sn, sw, n, width


def _level_blocks_on_device(plan, owner, levels, n_collected, use_cache, first_raw):
    """The stored samples of whole levels as one device tensor each, in the storage's [n][2][M] layout, for levels the storage
    hands out in several chunks: resident from an earlier estimate of any quantity over the same storage, or streamed now
    (_LevelStreamer, all levels as one pipeline).  levels: [(level id, [chunk specs])].  Generator of
    (level id, (flat tensor, sample stride, side stride, n, width)); levels that do not qualify are skipped (they are taken
    chunk by chunk; a first chunk that was read on the way is left in `first_raw` for that path)."""
    to_stream = []
    for level_id, level_specs in levels:
        stamp = None if n_collected is None else n_collected[level_id]
        key = ("lvlblock", id(owner), level_id, stamp)
        if use_cache:
            item = _device_cache.get(key)
            if item is not None and _block_meta.has(key, owner):
                yield level_id, (item[0],) + _block_meta.get(key, owner)
                continue
            if _block_meta.has(key + ("no", len(plan.in_rows)), owner):
                continue                                          # known: this level does not stream for a tree of that many rows
        if len(level_specs) < 2 or not _LevelStreamer.enabled() or os.environ.get("MLMC_HIP_BLOCK_UPLOAD", "1") == "0":
            continue
        if use_cache:                                             # a level that is partly resident keeps its chunk-wise path
            partly = False
            for cs in level_specs:
                ck = _chunk_key(None, cs, n_collected)[1:]
                if (("block", id(owner)) + ck) in _device_cache or any((("row", id(owner)) + ck + (r,)) in _device_cache for r in plan.in_rows):
                    partly = True
                    break
            if partly:
                continue
        to_stream.append((level_id, level_specs))
    if not to_stream:
        return
    first_spec = {level_id: specs[0] for level_id, specs in to_stream}
    for level_id, got, raw0 in _streamer.stream(plan.leaf, to_stream, len(plan.in_rows), _DeviceChunkCache.budget() // 4):
        stamp = None if n_collected is None else n_collected[level_id]
        key = ("lvlblock", id(owner), level_id, stamp)
        if got is None:
            if raw0 is not None:
                first_raw[id(first_spec[level_id])] = raw0
            if use_cache:
                _block_meta.put(key + ("no", len(plan.in_rows)), owner, None)
            continue
        _device_cache.uploads += 1
        if use_cache:
            _device_cache.put_tensors(key, got[0], None, owner=owner)
            _block_meta.put(key, owner, got[1:])
        yield level_id, got


def _evaluate_on_device(plan, chunk_spec, chunk_key, use_cache, raw=None):
    """Result rows of a lowered quantity for one chunk: fine [M, n'], coarse [M, n'] | None (torch CUDA tensors).
    raw: the chunk as the storage returned it, when a prefetcher has read it already."""
    blk = _stored_block_on_device(plan, chunk_spec, chunk_key, use_cache, raw)
    if blk is not None:
        t, sn, sw, n, width = blk
        # without the cache nothing keeps the uploaded block alive past this call: wait for the kernel that reads it
        fine, coarse, _ = plan.evaluate([t[r:] for r in plan.in_rows], has_coarse=(width == 2), n=n, sample_stride=sn,
                                        side_stride=max(sw, 1), sync=not use_cache)
        return fine, coarse
    storage = getattr(plan.leaf, "_storage", None)
    if raw is None and not hasattr(storage, "device_row"):
        owner = _owner_of(plan)
        if not use_cache or any(("row", id(owner)) + chunk_key + (r,) not in _device_cache for r in plan.in_rows):
            raw = plan.leaf.samples(chunk_spec)                  # ONE read of the chunk for all its rows
    rows = [_stored_row_on_device(plan, chunk_spec, chunk_key, r, use_cache, raw) for r in plan.in_rows]
    n, width = rows[0].shape
    if n == 0:
        return None
    if width == 1 and plan.is_row_copy:
        return rows[0].view(1, n), None         # a level-0 row already is the contiguous fine array: no kernel, no copy
    fine, coarse, _ = plan.evaluate(rows, has_coarse=(width == 2), n=n, sync=not use_cache)
    return fine, coarse


def _cache_ident(source, plan):
    """Identity of a quantity's rows in the device cache: (storage, lowered program) for lowered trees -- equivalent
    trees (rebuilt quantity objects, a second make_root_quantity over the same storage) share entries -- else the
    quantity object itself."""
    if plan is not None:
        return (id(_owner_of(plan)), plan.signature)
    return id(source)


def _level_stamps(storage_q):
    """Per level (samples collected, storage version): what the resident rows of a level are valid for.  Storages of the
    reference are append-only, so the count alone identifies a level's contents; mlmc_amd's own Memory storage also counts
    its modifications per level (`_level_versions`), which covers a level that was rebuilt with the same number of samples.  Arrays of a
    foreign storage edited in place are invisible to both: call device_cache_clear() after such an edit."""
    versions = getattr(getattr(storage_q, "_storage", None), "_level_versions", None) or {}
    return tuple((int(c), versions.get(level)) for level, c in enumerate(storage_q.n_collected()))


def _chunk_key(ident, chunk_spec, n_collected):
    sl = chunk_spec.chunk_slice
    return (ident, chunk_spec.level_id, chunk_spec.chunk_id, None if sl is None else (sl.start, sl.stop),
            None if n_collected is None else n_collected[int(chunk_spec.level_id)])


def _merge_level(pairs):
    """The resident chunks of one level as one tensor pair (one device copy), or None when they are not all resident /
    not all of one kind / too large to hold twice for a moment."""
    import torch
    pairs = [p for p in pairs if p is not None and p[0].shape[-1] > 0]
    if len(pairs) < 2 or not all(isinstance(p[0], torch.Tensor) for p in pairs):
        return None
    if len({p[1] is None for p in pairs}) != 1:
        return None
    nbytes = sum(p[0].numel() * 8 * (1 if p[1] is None else 2) for p in pairs)
    if 2 * nbytes > _DeviceChunkCache.budget():
        return None
    from .. import _lib
    _lib.check(_lib.lib().mlmc_synchronize())                     # the chunks may still be being written by k_expr
    fine = torch.cat([p[0] for p in pairs], dim=1).contiguous()
    coarse = None if pairs[0][1] is None else torch.cat([p[1] for p in pairs], dim=1).contiguous()
    torch.cuda.current_stream(fine.device).synchronize()          # the library reads them on its own stream
    return fine, coarse


def _chunk_for_device(source, plan, chunk_spec, n_collected, use_cache, raw=None):
    """Sample rows of `source` for one storage chunk, ready for the accumulators: (fine [M, n], coarse [M, n] | None) as
    torch CUDA tensors (resident: served from / added to the HBM cache) or, when the chunk does not fit the cache budget
    and the tree is evaluated on the host, as NumPy arrays that go through the staging buffer of the C ABI."""
    owner = _owner_of(plan) if plan is not None else source
    ident = _cache_ident(source, plan)
    key = _chunk_key(ident, chunk_spec, n_collected)
    item = _device_cache.get(key) if use_cache else None
    if item is not None:
        return item[0], item[1]
    if plan is not None:
        got = _evaluate_on_device(plan, chunk_spec, key[1:], use_cache, raw)
        if got is None:
            import torch
            return torch.empty((plan.n_out, 0), dtype=torch.float64), None
        if use_cache:
            _device_cache.put_tensors(key, got[0], got[1], owner=owner)
        return got
    raw = source.samples(chunk_spec)                             # [M, n, 2|1], host evaluation of the tree
    if raw.shape[1] == 0:
        return np.empty((raw.shape[0], 0)), None
    fine, coarse = _split_fine_coarse(raw, chunk_spec.level_id)
    item = _device_cache.put(key, fine, coarse, owner=source) if use_cache else None
    if item is None:
        return fine, coarse                                      # not cached: staged through the C ABI
    return item[0], item[1]


def fine_samples_for_device(quantity, chunk_spec):
    """Fine values of one chunk of `quantity` where the device wants them: a torch CUDA tensor when the tree is lowered
    (stored rows resident or generated in HBM, nothing crosses PCIe), else the host array of the host-evaluated tree.
    Used by Estimate.estimate_domain (percentiles of the fine samples)."""
    with _estimate_lock:
        return _fine_samples_for_device(quantity, chunk_spec)


def _fine_samples_for_device(quantity, chunk_spec):
    plan = lowering.plan_for(quantity) if _device_tree_enabled() else None
    if plan is None:
        return np.squeeze(quantity.samples(chunk_spec)[..., 0])
    storage_q = quantity.get_quantity_storage()
    try:
        n_collected = _level_stamps(storage_q)
    except Exception:
        n_collected = None
    use_cache = _DeviceChunkCache.budget() > 0 and n_collected is not None
    fine, _ = _chunk_for_device(quantity, plan, chunk_spec, n_collected, use_cache)
    return fine


def _subsample_on_device(pair, params):
    """Quantity.pick_samples (reference quantity.py:308-325) on the device: the chunk's share of the k-of-n sub-sample is
    drawn on the host (hypergeometric count, as in the reference), the `size` columns -- uniform with replacement,
    RNG.choice(chunk, size, axis=1) -- are gathered by mlmc_subsample_gather with a counter-based generator."""
    import torch
    from .. import _lib
    fine, coarse = pair
    n = fine.shape[-1]
    # scipy.stats.hypergeom(M = n_total, n = k, N = chunk).rvs() of the reference, drawn with the NumPy generator directly
    # (building a frozen scipy distribution costs ~0.3 ms per chunk, more than the estimate itself)
    size = int(qmod.RNG.hypergeometric(params._orig_k, params._orig_n - params._orig_k, min(n, params._orig_n)))
    seed = int(qmod.RNG.integers(0, 2 ** 63 - 1))
    if not isinstance(fine, torch.Tensor):                       # host chunk (over the cache budget): upload for the gather
        dev = torch.device("cuda", _lib_device())
        fine = torch.from_numpy(np.ascontiguousarray(fine)).to(dev)
        coarse = None if coarse is None else torch.from_numpy(np.ascontiguousarray(coarse)).to(dev)
        torch.cuda.current_stream(dev).synchronize()
    m = fine.shape[0]
    out_f = torch.empty((m, size), dtype=torch.float64, device=fine.device)
    out_c = None if coarse is None else torch.empty((m, size), dtype=torch.float64, device=fine.device)
    if size > 0:
        _lib.check(_lib.lib().mlmc_subsample_gather(fine.data_ptr(), None if coarse is None else coarse.data_ptr(), m, n, size,
                                                    seed, out_f.data_ptr(), None if out_c is None else out_c.data_ptr()))
    return out_f, out_c


def _linearized_basis(fn):
    """The larger member of fn's family whose level SUMS give the level sums of fn's moment covariance (linearize.py):
    Legendre / monomial / Fourier moments, `estimate_mean(covariance(q, fn), variance=False)`.  None: the covariance mean
    stays on the matrix cores (splines, transformed moments, sizes the family does not reach; MLMC_HIP_LINEARIZE=0)."""
    if os.environ.get("MLMC_HIP_LINEARIZE", "1") == "0":
        return None
    k_ext = linearize.extended_size(fn)
    if k_ext is None or k_ext > 512:
        return None
    ext = fn.__dict__.get("_lin_ext")
    if ext is None or ext.size != k_ext:
        ext = fn.change_size(k_ext)
        fn.__dict__["_lin_ext"] = ext          # one device handle and one pooled accumulator per moments object
    return ext


def _sums_from_linearized_memo(fn, key, owner):
    """Level sums of TransformedMoments(base, T) from the kept sums of a linearised covariance-mean estimate of `base` over the
    same samples (same quantity rows, same level stamps, same cache generation): the transformed moments are linear in the
    base moments, sum_n (T d_n) = T sum_n d_n, and the keep / drop decision is the base transform's.  This is the second
    pass of Estimate.construct_density (estimator.py:304-331) without a second pass.  -> (n, n_rm, sums [L, n_comp * Rt]) | None"""
    from ..moments import TransformedMoments
    if not isinstance(fn, TransformedMoments) or key is None:
        return None
    memo = fn._base.__dict__.get("_lin_memo")
    if memo is None or memo["key"] != key or fn._base_matrix.shape[1] != memo["R"] or memo["owner"]() is not owner:
        return None                       # (the key carries id(owner): the weak reference rules out a recycled id)
    L, n_comp, K, R = memo["sums"].shape[0], memo["n_comp"], memo["K"], memo["R"]
    base = memo["sums"].reshape(L * n_comp, K)[:, :R]
    return memo["n"].copy(), memo["n_rm"].copy(), (base @ fn._base_matrix.T).reshape(L, -1), n_comp


def estimate_mean(quantity, group=None, variance=True):
    """MLMC mean estimator (reference: quantity_estimate.py:22-80).  Thread-safe: estimates of several host threads are
    serialised (one GPU stream, one sample cache, the memo of chunk evaluations shared by all quantities).

    :param quantity: Quantity
    :param group: optional torch.distributed process group; when a group (or the default group) with more than one
                  rank is initialised every rank passes ITS shard of the samples and the level sums are all-reduced.
    :param variance: False when only `.mean` / `.l_means` will be read (Estimate.construct_density): device passes that
                  exist for the variances alone are skipped and the corresponding variances come back as NaN.
    :return: QuantityMean
    """
    with _estimate_lock:
        return _estimate_mean(quantity, group, variance)


def _estimate_mean(quantity, group, variance):
    cache_clear()
    quantity_vec_size = quantity.size()
    storage_q = quantity.get_quantity_storage()
    level_ids = storage_q.level_ids()
    n_levels = int(np.max(level_ids)) + 1

    if isinstance(quantity, (_MomentsNode, _CovarianceNode)):
        source = quantity._input_quantities[0]
        fn = quantity._moments_fn
        mode = engine.LevelAccumulator.MOMENTS if isinstance(quantity, _MomentsNode) else engine.LevelAccumulator.COV
        rows_per_comp = fn.size if mode == engine.LevelAccumulator.MOMENTS else fn.size * fn.size
    else:
        source, fn, mode, rows_per_comp = quantity, None, engine.LevelAccumulator.MOMENTS, 1
    rows_out = rows_per_comp                                      # rows per component the caller sees
    # Mean of the moment covariance without its variance (Estimate.construct_density): for Legendre / monomial / Fourier
    # moments the R x R level sums are a fixed linear map of the level sums of ~2 R moments of the same family
    # (linearize.py) -- one pass of the moments kernel instead of the matrix-core pass.
    lin_fn = None
    if mode == engine.LevelAccumulator.COV and not variance:
        ext = _linearized_basis(fn)
        if ext is not None:
            lin_fn, fn, mode, rows_per_comp = fn, ext, engine.LevelAccumulator.MOMENTS, ext.size

    # bootstrap sub-sample of a quantity (Quantity.subsample): the chunks of the underlying quantity stay resident in HBM,
    # every estimate draws its random columns on the device (mlmc_subsample_gather)
    subsample_params = None
    sym = source.__dict__.get("_sym")
    if sym is not None and sym[0] == "subsample" and _device_tree_enabled():
        subsample_params = sym[1]
        source = source._input_quantities[0]

    # a tree of per-sample nodes runs as one device program over the stored rows (quantity/lowering.py)
    plan = lowering.plan_for(source) if _device_tree_enabled() else None
    acc = None
    n_comp = None
    use_cache = _DeviceChunkCache.budget() > 0 and not getattr(source, "_volatile", False)
    try:
        n_collected = _level_stamps(storage_q)
    except Exception:
        n_collected = None
        use_cache = False
    # A level that arrives in several resident chunks (HDF5 files deliver levels that way) is concatenated into ONE tensor
    # per level before it is pushed -- once; the level-wide entry replaces the chunk entries in the cache -- so every
    # estimate, the first one included, is one push per level and gives bit-identical sums.
    consolidate = use_cache and subsample_params is None
    ident = _cache_ident(source, plan)
    owner = _owner_of(plan) if plan is not None else source

    def push_pair(level_id, pair):
        nonlocal acc, n_comp
        if pair is None:
            return
        if acc is None:
            n_comp = pair[0].shape[0]
            if pair[0].shape[-1] > 0:
                assert n_comp * rows_out == quantity_vec_size
            acc = _acc_pool.take(fn, n_levels, mode, n_comp, mean_only=not variance)
        if pair[0].shape[-1] == 0:                               # empty chunk / every sample deselected
            return
        fine, coarse = pair
        if n_comp == 1:
            fine, coarse = fine[0], (None if coarse is None else coarse[0])
        # Resident chunks are gathered and handed over in ONE call of the C ABI at the end (mlmc_accum_estimate: reset +
        # pushes + finalize); as soon as a host chunk shows up -- it is staged through a reused device buffer and has to be
        # pushed at once -- everything goes through push() in arrival order.
        resident = (not pushed_directly and getattr(fine, "is_cuda", False) and fine.is_contiguous()
                    and (coarse is None or (getattr(coarse, "is_cuda", False) and coarse.is_contiguous())))
        if resident:
            gathered.append((level_id, fine, coarse))
            return
        flush_gathered()
        acc.push(level_id, fine, coarse)

    gathered = []
    pushed_directly = False

    def flush_gathered():
        nonlocal pushed_directly
        pushed_directly = True
        for item in gathered:
            acc.push(*item)
        gathered.clear()

    def flush_level(level_id, pairs, keys):
        if not pairs:
            return
        merged = _merge_level(pairs) if (consolidate and len(pairs) > 1) else None
        if merged is None:
            for pair in pairs:
                push_pair(level_id, pair)
            return
        _device_cache.put_tensors((ident, level_id, "level", n_collected[level_id]), merged[0], merged[1], owner=owner)
        for key in keys:
            _device_cache.drop(key)
        push_pair(level_id, merged)

    memo_key = None
    if n_collected is not None and subsample_params is None and group is None and not engine._dist_group_active(group):
        memo_key = (ident, n_collected, _cache_generation)
    if not variance and mode == engine.LevelAccumulator.MOMENTS and lin_fn is None and fn is not None:
        short = _sums_from_linearized_memo(fn, memo_key, owner)
        if short is not None:
            n_samples, n_rm_samples, sums, n_comp = short
            return _finish_estimate(quantity, fn, n_levels, n_comp, rows_out, n_samples, n_rm_samples, sums,
                                    np.full_like(sums, np.nan))

    level_done = set()
    if use_cache:
        for level_id in sorted({int(l) for l in level_ids}):
            item = _device_cache.get((ident, level_id, "level", n_collected[level_id]))
            if item is not None:
                level_done.add(level_id)
                pair = (item[0], item[1])
                if subsample_params is not None and pair[0].shape[-1] > 0:   # the whole level is one resident chunk
                    pair = _subsample_on_device(pair, subsample_params[level_id])
                push_pair(level_id, pair)
    # (a storage that cuts its levels into hundreds of chunks: when every level is served level-wide from HBM the chunk
    # specs need not even be generated)
    all_levels = {int(l) for l in level_ids}
    specs = [] if level_done >= all_levels else [cs for cs in storage_q.chunks() if int(cs.level_id) not in level_done]
    host_storage = plan is not None and not hasattr(getattr(plan.leaf, "_storage", None), "device_row")
    # Levels of a lowered tree that the storage hands out in several chunks of [n][2][M] records: the whole level becomes
    # ONE device tensor in the storage's layout -- resident from an earlier estimate of any quantity over this storage, or
    # streamed now through pinned staging blocks (_LevelStreamer) -- and the tree runs as ONE k_expr launch over it.
    first_raw = {}
    if host_storage:
        by_level = collections.OrderedDict()
        for cs in specs:
            by_level.setdefault(int(cs.level_id), []).append(cs)
        candidates = [(level_id, level_specs) for level_id, level_specs in by_level.items()
                      if not (use_cache and any(_chunk_key(ident, cs, n_collected) in _device_cache for cs in level_specs))]
        for level_id, blk in _level_blocks_on_device(plan, owner, candidates, n_collected, use_cache, first_raw):
            t, sn, sw, n, width = blk
            fine, coarse, _ = plan.evaluate([t[r:] for r in plan.in_rows], has_coarse=(width == 2), n=n, sample_stride=sn,
                                            side_stride=max(sw, 1), sync=not use_cache)
            if use_cache:
                _device_cache.put_tensors((ident, level_id, "level", n_collected[level_id]), fine, coarse, owner=owner)
            pair = (fine, coarse)
            if subsample_params is not None and n > 0:
                pair = _subsample_on_device(pair, subsample_params[level_id])
            push_pair(level_id, pair)
            level_done.add(level_id)
        specs = [cs for cs in specs if int(cs.level_id) not in level_done]
    # remaining chunks of a lowered tree that have to come from the host storage (neither their result rows nor their stored
    # block are resident): read ahead by a helper thread while this thread uploads and launches (_ChunkPrefetcher)
    to_read = []
    if host_storage and _ChunkPrefetcher.enabled():
        have = _device_cache
        for cs in specs:
            key = _chunk_key(ident, cs, n_collected)
            resident = use_cache and (key in have or (("block", id(owner)) + key[1:]) in have
                                      or all((("row", id(owner)) + key[1:] + (r,)) in have for r in plan.in_rows))
            if not resident and id(cs) not in first_raw:
                to_read.append(cs)
    prefetch = _ChunkPrefetcher(plan.leaf, to_read) if len(to_read) > 1 else None
    waiting = {id(cs) for cs in to_read} if prefetch is not None else set()
    current, pairs, keys = None, [], []
    try:
        for chunk_spec in specs:
            level_id = int(chunk_spec.level_id)
            if level_id != current:
                flush_level(current, pairs, keys)
                current, pairs, keys = level_id, [], []
            raw = prefetch.get(chunk_spec) if id(chunk_spec) in waiting else first_raw.pop(id(chunk_spec), None)
            pair = _chunk_for_device(source, plan, chunk_spec, n_collected, use_cache, raw)     # (fine [M, n], coarse | None)
            if pair is not None and subsample_params is not None and pair[0].shape[-1] > 0:
                pair = _subsample_on_device(pair, subsample_params[level_id])
            pairs.append(pair)
            keys.append(_chunk_key(ident, chunk_spec, n_collected))
        flush_level(current, pairs, keys)
    finally:
        if prefetch is not None:
            prefetch.close()
    if acc is None:
        raise Exception("All samples were masked")
    if gathered and not pushed_directly:
        n_samples, n_rm_samples, sums, sums_sq = acc.estimate(gathered, group=group)
    else:
        flush_gathered()
        n_samples, n_rm_samples, sums, sums_sq = acc.finalize(group=group)
    _acc_pool.give(fn, n_levels, mode, n_comp, acc, mean_only=not variance)
    if int(np.sum(n_samples)) == 0:
        raise Exception("All samples were masked")
    if lin_fn is not None:
        if memo_key is not None:
            lin_fn.__dict__["_lin_memo"] = dict(key=memo_key, n=n_samples.copy(), n_rm=n_rm_samples.copy(), sums=sums.copy(),
                                                n_comp=n_comp, K=fn.size, R=lin_fn.size, owner=_BlockMeta._ref(owner))
        sums = linearize.covariance_sums_from_moment_sums(lin_fn, sums, n_comp)
        sums_sq = np.full_like(sums, np.nan)                       # mean only: the variances were not asked for
        fn = lin_fn
    return _finish_estimate(quantity, fn, n_levels, n_comp, rows_out, n_samples, n_rm_samples, sums, sums_sq)


def _finish_estimate(quantity, fn, n_levels, n_comp, rows_per_comp, n_samples, n_rm_samples, sums, sums_sq):
    if fn is not None and not quantity._at_bottom and n_comp > 1:
        # device rows are (component, moment...); 'on the surface' wants (moment..., component)
        sums = sums.reshape(n_levels, n_comp, rows_per_comp).transpose(0, 2, 1).reshape(n_levels, -1)
        sums_sq = sums_sq.reshape(n_levels, n_comp, rows_per_comp).transpose(0, 2, 1).reshape(n_levels, -1)
    l_means, l_vars = engine.level_stats(n_samples, sums, sums_sq)
    return qmod.QuantityMean(quantity.qtype, l_means=l_means, l_vars=l_vars, n_samples=[int(v) for v in n_samples],
                             n_rm_samples=[int(v) for v in n_rm_samples])
