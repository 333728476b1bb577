"""CPU tests of the resident-sample cache under threads (no GPU: the cached "tensors" are stand-ins with numel()).

Round 2 recorded a "worker thread hangs" failure of test_concurrent_estimates_through_the_python_api.
The run's timing shows that no thread was stuck on the GPU: of the three 300-s joins exactly one timed out, the one of the
cache-clearing helper, which loops until the analysis thread sets `stop` at its end -- the analysis thread had ended
with an exception instead.  The exception that the code of that day could raise: `_DeviceChunkCache.get` was a lookup
followed by `move_to_end`, `device_cache_clear()` took no lock, and a clear between the two steps gives KeyError.  The
first test below forces exactly that interleaving with a second thread; the second one hammers the cache."""
import collections
import threading

import pytest


class _FakeTensor:
    def __init__(self, n):
        self._n = n

    def numel(self):
        return self._n


class _HookedDict(collections.OrderedDict):
    """OrderedDict whose get() pauses after the lookup (the window of the race) until `hook` returns."""
    hook = None

    def get(self, key, default=None):
        item = super().get(key, default)
        if item is not None and _HookedDict.hook is not None:
            _HookedDict.hook()
        return item


def test_clear_between_lookup_and_move_to_end_cannot_interleave():
    from mlmc_amd.quantity import quantity_estimate as qe
    cache = qe._DeviceChunkCache()
    cache._items = _HookedDict()
    cache.put_tensors(("k", 1), _FakeTensor(10), None, owner=None)
    go, cleared = threading.Event(), threading.Event()

    def clearer():
        go.wait(5)
        cache.clear()               # must block while `get` is between its two steps
        cleared.set()

    t = threading.Thread(target=clearer, daemon=True)
    t.start()
    state = {}

    def hook():
        _HookedDict.hook = None
        go.set()
        state["cleared_inside_get"] = cleared.wait(0.3)     # an unlocked clear() finishes within microseconds

    _HookedDict.hook = hook
    try:
        item = cache.get(("k", 1))      # the code of round 2 raised KeyError here (move_to_end of a vanished key)
    finally:
        _HookedDict.hook = None
    t.join(5)
    assert item is not None and state["cleared_inside_get"] is False
    assert cleared.is_set() and cache.get(("k", 1)) is None and cache._bytes == 0


def test_cache_hammered_by_three_threads_keeps_its_books():
    from mlmc_amd.quantity import quantity_estimate as qe
    cache = qe._DeviceChunkCache()
    owners = [object(), object()]
    stop = threading.Event()
    errors = []

    def user(seed):
        try:
            for i in range(4000):
                key = ("k", (seed * 7919 + i) % 37)
                if cache.get(key) is None:
                    cache.put_tensors(key, _FakeTensor(100 + i % 5), _FakeTensor(3) if i % 2 else None, owner=owners[i % 2])
                if i % 11 == 0:
                    cache.drop(key)
                assert (key in cache) in (True, False)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)
        finally:
            stop.set()

    def clearer():
        try:
            while not stop.wait(0.0005):
                cache.clear()
                cache.drop_owner(owners[0])
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=user, args=(1,), daemon=True), threading.Thread(target=user, args=(2,), daemon=True),
               threading.Thread(target=clearer, daemon=True)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    assert cache._bytes == sum(item[2] for item in cache._items.values())


def test_block_meta_forgets_dead_owners():
    """A decision remembered for one storage must not be inherited by a new storage that gets the freed object's id."""
    from mlmc_amd.quantity import quantity_estimate as qe

    class Storage:
        pass

    meta = qe._BlockMeta()
    a = Storage()
    key = ("block", id(a), 0, 0, None, 10)
    meta.put(key, a, None)
    meta.put(key + ("any",), a, None)
    assert meta.has(key, a) and meta.get(key, a) is None
    b = Storage()
    assert not meta.has(key, b)                      # same key, another live object: no inheritance (and the entry is dropped)
    meta.put(key, a, (2, 1, 10, 2))
    assert meta.get(key, a) == (2, 1, 10, 2)
    meta.drop_owner(a)
    assert not meta.has(key, a) and not meta.has(key + ("any",), a) and len(meta) == 0
    meta.put(key, a, (2, 1, 10, 2))
    del a                                            # the owner dies: the weak reference goes dead with it
    assert meta.get(key, b, "gone") == "gone"


def test_cache_entries_die_with_their_owner():
    """The resident-sample cache holds its owners (storages, source quantities) weakly: rows of a storage that no longer
    exists are invisible at once and are purged with the next insertions -- the cache must not keep a storage's host arrays
    alive, and a recycled id() must not find another object's rows."""
    import gc
    from mlmc_amd.quantity import quantity_estimate as qe

    class Storage:
        pass

    cache = qe._DeviceChunkCache()
    a, b = Storage(), Storage()
    ka, kb = ("row", id(a), 0), ("row", id(b), 0)
    cache.put_tensors(ka, _FakeTensor(1000), None, owner=a)
    cache.put_tensors(kb, _FakeTensor(500), _FakeTensor(500), owner=b)
    assert ka in cache and kb in cache and cache._bytes == 8000 + 8000
    del a
    gc.collect()
    assert ka not in cache and cache.get(ka) is None and kb in cache         # invisible at once, dropped by the failed lookup
    assert cache._bytes == 8000
    cache.drop_owner(b)
    assert kb not in cache and cache._bytes == 0
    # purge on insertion: dead entries give their bytes back without anybody looking them up
    owners = [Storage() for _ in range(70)]
    for i, o in enumerate(owners):
        cache.put_tensors(("row", id(o), i), _FakeTensor(10), None, owner=o)
    del owners, o
    gc.collect()
    keeper = Storage()
    for i in range(64):
        cache.put_tensors(("keep", id(keeper), i), _FakeTensor(1), None, owner=keeper)
    assert all(k[0] == "keep" for k in cache._items) and cache._bytes == 64 * 8
