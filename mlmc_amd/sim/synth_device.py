"""SynthSimulation samples generated in HBM.

The reference produces its synthetic test samples one by one on the host: `Sampler` names a sample "L{level:02d}_S{i:07d}"
(mlmc/sampler.py:120), the pool derives the seed from the md5 of that name (mlmc/sampling_pool.py:75-84) and
`SynthSimulation.calculate` draws two normals from `RandomState(seed)` and applies x + h sqrt(1e-4 + |x|)
(mlmc/sim/synth_simulation.py:37-46,75-131).  `mlmc_synth_generate` (mlmc_amd/csrc/synth.hip) runs that chain per
sample on the device, so a storage of 10^8 samples per level exists only as the rows an analysis actually reads.

`SynthDeviceStorage` presents those samples through the SampleStorage interface (mlmc/sample_storage.py:9-132); rows
are generated on first use and served to the device estimators without touching the host.
"""
import ctypes as C

import numpy as np

from .. import _lib
from ..quantity.quantity_spec import ChunkSpec, QuantitySpec
from ..sample_storage import SampleStorage

N_ROWS = 24          # result_format(): 2 quantities x 3 times x 2 locations x shape (2, 1)


def result_format():
    """SynthSimulation.result_format (synth_simulation.py:136-145)."""
    return [QuantitySpec(name="length", unit="m", shape=(2, 1), times=[1, 2, 3], locations=['10', '20']),
            QuantitySpec(name="width", unit="mm", shape=(2, 1), times=[1, 2, 3], locations=['30', '40'])]


def n_ops_estimate(step, complexity=2):
    """SynthSimulation.n_ops_estimate (synth_simulation.py:133-134)."""
    return (1 / step) ** complexity * np.log(max(1 / step, 2.0))


def sample_seeds(level_id, first_sample, n):
    """SamplingPool.compute_seed of the sample ids of a level, evaluated on the device (uint32 array)."""
    out = np.empty(int(n), dtype=np.uint32)
    _lib.check(_lib.lib().mlmc_synth_seeds(int(level_id), int(first_sample), int(n), out.ctypes.data_as(C.c_void_p)))
    return out


def generate_rows(level_id, first_sample, n, fine_step, coarse_step, rows, device=None, loc=0.0, scale=1.0, out=None):
    """Stored rows `rows` (indices 0..23) of samples first_sample .. first_sample + n - 1 of one level, as torch CUDA
    tensors in the storage layout: [n, 2] (fine, coarse) when coarse_step != 0, else [n, 1].
    out: existing contiguous tensors [n, width] to fill (complete in memory on torch's stream) instead of new ones."""
    import torch
    lib = _lib.lib()
    dev = torch.device("cuda", _lib._bound_device if device is None else device)
    width = 2 if coarse_step != 0 else 1
    if out is None:
        out = [torch.empty((int(n), width), dtype=torch.float64, device=dev) for _ in rows]
        torch.cuda.current_stream(dev).synchronize()
    assert all(t.is_contiguous() and tuple(t.shape) == (int(n), width) for t in out)
    if int(n) == 0:
        return out
    row_ids = (C.c_int32 * len(rows))(*[int(r) for r in rows])
    ptrs = (C.c_void_p * len(rows))(*[t.data_ptr() for t in out])
    _lib.check(lib.mlmc_synth_generate(int(level_id), int(first_sample), int(n), float(fine_step), float(coarse_step),
                                       float(loc), float(scale), len(rows), row_ids, ptrs))
    return out


class SynthDeviceStorage(SampleStorage):
    """Read-only storage of SynthSimulation samples that live (only) in HBM."""

    def __init__(self, level_steps, n_samples, chunk_size=None, complexity=2, loc=0.0, scale=1.0, shard=None):
        """level_steps: fine simulation step of every level (e.g. estimator.determine_level_parameters);
        n_samples: collected samples per level; chunk_size: samples per chunk (None: one chunk per level);
        loc, scale: the distribution scipy.stats.norm(loc, scale) of SynthSimulation's config;
        shard: (rank, world_size) -- this process owns the contiguous slice engine.shard_bounds of every level
        (multi-GPU estimates: one process per GPU, the level sums are all-reduced by estimate_mean)."""
        self._loc, self._scale = float(loc), float(scale)
        self._steps = [float(np.ravel(s)[0]) for s in level_steps]
        self._shard = shard
        self._rows = {}       # (level, stored row) -> [tensor [capacity, width], generated samples, first sample index]
        self._set_counts(n_samples)
        assert len(self._steps) == len(self._n)
        self._chunk_size = chunk_size
        self._n_ops = [n_ops_estimate(h, complexity) for h in self._steps]

    def _set_counts(self, n_samples):
        from ..engine import shard_bounds
        bounds = [shard_bounds(int(v), *self._shard) if self._shard is not None else (0, int(v)) for v in n_samples]
        self._first = [lo for lo, _ in bounds]             # sample index of the first owned sample of every level
        self._n = [hi - lo for lo, hi in bounds]

    def set_n_samples(self, n_samples):
        """Grow the levels to n_samples collected samples (job-wide counts; a shard re-derives its slice): the samples
        are a function of (level, index), so new ones only extend every level's sequence (sampler.DeviceSampler).
        Rows of the former extent that estimates left in the HBM cache are released."""
        assert len(n_samples) == len(self._steps)
        old = (list(self._first), list(self._n))
        self._set_counts(n_samples)
        if old != (self._first, self._n):
            from ..quantity import quantity_estimate
            quantity_estimate.device_cache_drop_owner(self)

    # ---- the part of the interface the estimators use ---------------------------------------------------------
    def get_level_ids(self):
        return list(range(len(self._n)))

    def get_n_levels(self):
        return len(self._n)

    def get_n_collected(self):
        return list(self._n)

    def get_level_parameters(self):
        return [[h] for h in self._steps]

    def get_n_ops(self):
        return list(self._n_ops)

    def load_result_format(self):
        return result_format()

    def n_finished(self):
        return np.array(self._n, dtype=float)

    def _level_chunks(self, level_id, n_samples=None):
        total = self._n[level_id] if n_samples is None else min(self._n[level_id], n_samples)
        size = total if (self._chunk_size is None or total == 0) else self._chunk_size
        if total == 0:
            yield ChunkSpec(chunk_id=0, chunk_slice=slice(0, 0, 1), level_id=level_id)
            return
        for cid, start in enumerate(range(0, total, size)):
            yield ChunkSpec(chunk_id=cid, chunk_slice=slice(start, min(start + size, total), 1), level_id=level_id)

    def _steps_of(self, level_id):
        return self._steps[level_id], (self._steps[level_id - 1] if level_id > 0 else 0.0)

    def _resident_bytes(self):
        return sum(ent[0].numel() * 8 for ent in self._rows.values())

    def _level_row(self, level, stored_row):
        """The stored row of a whole level, resident in HBM: generated once, extended in place when the level grows
        (samples are a function of (level, index): an adaptive loop only ever generates the new ones)."""
        import torch
        n, first = self._n[level], self._first[level]
        h_f, h_c = self._steps_of(level)
        ent = self._rows.get((level, stored_row))
        if ent is not None and (ent[2] != first or ent[1] > n):      # a shard whose slice moved: start over
            ent = None
        if ent is None:
            ent = [generate_rows(level, first, n, h_f, h_c, [stored_row], loc=self._loc, scale=self._scale)[0], n, first]
            self._rows[(level, stored_row)] = ent
            return ent[0]
        t, have = ent[0], ent[1]
        if have < n:
            if t.shape[0] < n:                                       # grow geometrically, keep what exists
                _lib.check(_lib.lib().mlmc_synchronize())            # the old rows may still be being generated / read
                grown = torch.empty((max(n, int(1.5 * t.shape[0])), t.shape[1]), dtype=torch.float64, device=t.device)
                grown[:have].copy_(t[:have])
                torch.cuda.current_stream(t.device).synchronize()    # the library writes / reads it on its own stream
                ent[0] = t = grown
            generate_rows(level, first + have, n - have, h_f, h_c, [stored_row], loc=self._loc, scale=self._scale,
                          out=[t[have:n]])
            ent[1] = n
        return t[:n]

    def device_row(self, chunk_spec, stored_row):
        """One stored row of a chunk in HBM: torch CUDA tensor [n, 2] (level 0: [n, 1]).  Rows of a level stay resident
        and are extended when the level grows, as long as the storage holds less than `resident_gb` (default: the device
        cache budget MLMC_HIP_DEVICE_CACHE_GB); beyond that every request generates its chunk afresh."""
        level = int(chunk_spec.level_id)
        sl = chunk_spec.chunk_slice if chunk_spec.chunk_slice is not None else slice(0, self._n[level], 1)
        from ..quantity.quantity_estimate import _DeviceChunkCache
        width = 2 if level > 0 else 1
        if (level, stored_row) in self._rows or self._resident_bytes() + self._n[level] * width * 8 <= _DeviceChunkCache.budget():
            return self._level_row(level, stored_row)[sl.start:sl.stop]
        h_f, h_c = self._steps_of(level)
        return generate_rows(level, self._first[level] + sl.start, sl.stop - sl.start, h_f, h_c, [stored_row], loc=self._loc,
                             scale=self._scale)[0]

    def release_rows(self):
        """Free the resident rows (they are regenerated on the next read)."""
        self._rows.clear()

    def sample_pairs_level(self, chunk_spec):
        """Host copy [24, n, 2|1] of a chunk (small chunks / tests; the estimators use device_row)."""
        level = int(chunk_spec.level_id)
        sl = chunk_spec.chunk_slice if chunk_spec.chunk_slice is not None else slice(0, self._n[level], 1)
        h_f, h_c = self._steps_of(level)
        rows = generate_rows(level, self._first[level] + sl.start, sl.stop - sl.start, h_f, h_c, list(range(N_ROWS)), loc=self._loc,
                             scale=self._scale)
        _lib.check(_lib.lib().mlmc_synchronize())
        return np.stack([t.cpu().numpy() for t in rows])

    def sample_pairs(self):
        return [self.sample_pairs_level(ChunkSpec(level_id=l)) for l in self.get_level_ids()]

    # ---- write side of the interface: the samples are a pure function of (level, index) -------------------------
    def _read_only(self, *args, **kwargs):
        raise NotImplementedError("SynthDeviceStorage is generated, not written")

    save_samples = save_result_format = save_global_data = save_scheduled_samples = save_n_ops = _read_only

    def load_scheduled_samples(self):
        return {}

    def unfinished_ids(self):
        return []
