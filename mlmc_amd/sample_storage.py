"""Sample storage interface and the in-memory storage (reference interface: mlmc/sample_storage.py:9-338).

Data supplier of the hot path: the estimators only need `chunks()`, `sample_pairs_level()`,
`get_level_ids()`, `get_n_levels()`, `get_n_collected()`, `get_level_parameters()` and `get_n_ops()`; any object
with these methods -- including the reference's own `Memory` / `SampleStorageHDF` instances -- can be passed to
`make_root_quantity` and `Estimate`.
"""
import itertools
from abc import ABCMeta, abstractmethod

import numpy as np

from .quantity.quantity_spec import ChunkSpec


class SampleStorage(metaclass=ABCMeta):
    @abstractmethod
    def save_samples(self, successful_samples, failed_samples):
        """Store results of finished samples."""

    @abstractmethod
    def save_result_format(self, res_spec):
        """Store the result format (list of QuantitySpec)."""

    @abstractmethod
    def load_result_format(self):
        """Return the result format."""

    @abstractmethod
    def save_global_data(self, result_format, level_parameters=None):
        """Store result format and level parameters."""

    @abstractmethod
    def save_scheduled_samples(self, level_id, samples):
        """Remember ids of scheduled samples."""

    @abstractmethod
    def load_scheduled_samples(self):
        """Dict[level_id, list of sample ids]."""

    @abstractmethod
    def sample_pairs(self):
        """List over levels of arrays [M, N, 2]."""

    def chunks(self, level_id=None, n_samples=None):
        """Generator of ChunkSpec over the requested (or all) levels."""
        assert isinstance(n_samples, (type(None), int)), "n_samples param must be int"
        level_ids = self.get_level_ids() if level_id is None else [level_id]
        return itertools.chain(*[self._level_chunks(lid, n_samples) for lid in level_ids])

    @abstractmethod
    def _level_chunks(self, level_id, n_samples=None):
        """Generator of ChunkSpec of one level."""

    @abstractmethod
    def n_finished(self):
        """Finished samples per level."""

    @abstractmethod
    def save_n_ops(self, n_ops):
        """Store cost estimates per level."""

    @abstractmethod
    def get_n_ops(self):
        """Cost per sample on every level."""

    @abstractmethod
    def unfinished_ids(self):
        """Ids of scheduled but unfinished samples."""

    @abstractmethod
    def get_level_ids(self):
        """Level ids."""

    @abstractmethod
    def get_n_levels(self):
        """Number of levels."""

    @abstractmethod
    def get_level_parameters(self):
        """Level parameters (simulation steps)."""

    @abstractmethod
    def get_n_collected(self):
        """Collected samples per level."""


class Memory(SampleStorage):
    """All samples in host RAM, one chunk per level (reference: sample_storage.py:135-338).

    Besides the reference's `save_samples` protocol (lists of (sample_id, (fine, coarse)) per level) whole levels can be
    handed over as arrays with `set_level_samples`, which is how large synthetic sets are loaded without Python loops.
    """

    def __init__(self, chunk_size=None, copy_chunks=False):
        """chunk_size: samples per chunk (None = one chunk per level, the reference's Memory).  copy_chunks: every
        `sample_pairs_level` call hands out a freshly allocated [n, 2, M] array, as a file-backed storage does
        (SampleStorageHDF reads `collected_values[chunk_slice]` from the file per chunk, mlmc/tool/hdf5.py:365-376):
        the stand-in for HDF5 storages in tests and in bench.py's `h2d_inclusive` block (h5py is not part of this image)."""
        self._copy_chunks = bool(copy_chunks)
        self._failed = {}
        self._results = {}
        self._successful_sample_ids = {}
        self._scheduled = {}
        self._result_specification = []
        self._n_ops = {}
        self._n_finished = {}
        self._level_parameters = []
        self._chunk_size = chunk_size     # None = one chunk per level (reference behaviour)

    def save_samples(self, successful_samples, failed_samples):
        self._save_successful(successful_samples)
        self._save_failed(failed_samples)

    def save_global_data(self, result_format, level_parameters=None):
        self.save_result_format(result_format)
        self._level_parameters = level_parameters

    def set_level_samples(self, level_id, fine, coarse=None, sample_ids=None):
        """Append a block of samples: fine / coarse arrays [N] or [N, M] (coarse ignored / zero at level 0)."""
        fine = np.asarray(fine, dtype=np.float64)
        fine = fine.reshape(fine.shape[0], int(np.prod(fine.shape[1:])))     # also for a block without samples
        coarse = np.zeros_like(fine) if coarse is None else np.asarray(coarse, dtype=np.float64).reshape(fine.shape)
        block = np.stack([fine, coarse], axis=1)                # [N, 2, M]
        self._append(level_id, block, sample_ids if sample_ids is not None else
                     ["L{:02d}_S{:07d}".format(level_id, i) for i in range(self._n_finished.get(level_id, 0),
                                                                          self._n_finished.get(level_id, 0) + len(block))])

    def _append(self, level_id, block, sample_ids):
        # per-level modification count: stamp of the HBM-resident copies of the level (quantity_estimate._level_stamps)
        self._level_versions = getattr(self, "_level_versions", {})
        self._level_versions[int(level_id)] = self._level_versions.get(int(level_id), 0) + 1
        self._successful_sample_ids.setdefault(level_id, []).extend(sample_ids)
        self._n_finished[level_id] = self._n_finished.get(level_id, 0) + block.shape[0]
        if level_id in self._results:
            self._results[level_id] = np.concatenate((self._results[level_id], block), axis=0)
        else:
            self._results[level_id] = block

    def _save_successful(self, samples):
        """samples: Dict[level_id, List[(sample_id, (fine_result, coarse_result))]]"""
        for level_id, res in samples.items():
            if len(res) == 0:
                continue
            ids = [r[0] for r in res]
            block = np.array([[np.ravel(r[1][0]), np.ravel(r[1][1])] for r in res], dtype=np.float64)   # [N, 2, M]
            self._append(level_id, block, ids)

    def _save_failed(self, samples):
        for level_id, res in samples.items():
            self._failed.setdefault(level_id, []).extend(res)
            if level_id not in self._n_finished:
                self._n_finished[level_id] = 0
            else:
                self._n_finished[level_id] += len(res)

    def save_result_format(self, res_spec):
        self._result_specification = res_spec

    def n_finished(self):
        out = np.zeros(max(self._n_finished.keys()) + 1)
        for level_id, n in self._n_finished.items():
            out[level_id] = n
        return out

    def load_result_format(self):
        return self._result_specification

    def save_scheduled_samples(self, level_id, samples):
        self._scheduled.setdefault(level_id, []).extend(samples)

    def load_scheduled_samples(self):
        return self._scheduled

    def sample_pairs(self):
        return [self.sample_pairs_level(ChunkSpec(level_id=level_id)) for level_id in sorted(self.get_level_ids())]

    def _level_chunks(self, level_id, n_samples=None):
        total = len(self._results[level_id][:n_samples])
        if self._chunk_size is None or total == 0:
            yield ChunkSpec(chunk_id=0, chunk_slice=slice(0, total, 1), level_id=level_id)
            return
        for cid, start in enumerate(range(0, total, self._chunk_size)):
            yield ChunkSpec(chunk_id=cid, chunk_slice=slice(start, min(start + self._chunk_size, total), 1), level_id=level_id)

    def sample_pairs_level(self, chunk_spec):
        """-> ndarray [M, chunk size, 2]; level 0 has no coarse samples: [M, chunk size, 1]."""
        results = self._results[int(chunk_spec.level_id)]
        chunk = results if chunk_spec.chunk_slice is None else results[chunk_spec.chunk_slice]
        if self._copy_chunks:
            chunk = np.array(chunk, copy=True)
        if chunk.ndim != 3:
            chunk = chunk.reshape(chunk.shape[0], chunk.shape[1], -1)
        if chunk_spec.level_id == 0:
            chunk = chunk[:, :1, :]
        return chunk.transpose((2, 0, 1))

    def sample_records_into(self, chunk_spec, out):
        """Optional fast path of the storage interface (not part of the reference's SampleStorage): write the chunk's records
        -- the storage layout [n][2][M], float64, what SampleStorageHDF keeps in `collected_values` (mlmc/tool/hdf5.py:14-46)
        -- straight into `out`, a C-contiguous float64 array [n, 2, M] supplied by the caller.  The streaming feed
        (quantity_estimate._LevelStreamer) hands in a window of a PINNED staging block, so a chunk costs one copy instead of
        two and no freshly faulted pages; an HDF5-backed storage implements this with `Dataset.read_direct(out, np.s_[a:b])`."""
        results = self._results[int(chunk_spec.level_id)]
        chunk = results if chunk_spec.chunk_slice is None else results[chunk_spec.chunk_slice]
        np.copyto(out, chunk.reshape(out.shape))

    def save_n_ops(self, n_ops):
        """n_ops: iterable of (level, (time, number of valid samples))"""
        for level, (time, n_samples) in n_ops:
            self._n_ops.setdefault(level, 0)
            if n_samples != 0:
                self._n_ops[level] += time / n_samples

    def get_n_ops(self):
        return [self._n_ops[level] for level in sorted(self._n_ops.keys())]

    def unfinished_ids(self):
        return []

    def get_level_ids(self):
        return list(self._results.keys())

    def get_n_collected(self):
        n_collected = [0] * len(self._results)
        for level_id in self.get_level_ids():
            n_collected[int(level_id)] = len(self._results[int(level_id)])
        return n_collected

    def get_n_levels(self):
        return len(self._results)

    def get_level_parameters(self):
        return self._level_parameters


class DeviceMemory(SampleStorage):
    """Samples that already live in HBM (torch CUDA tensors), presented through the SampleStorage interface: a level is a
    tensor `[M, n, 2]` (fine, coarse pairs of M stored rows; level 0: `[M, n, 1]`), handed over with `set_level_samples`.
    The estimators read the rows where they are (`device_row`: the same hook `sim.synth_device.SynthDeviceStorage` offers) --
    nothing crosses PCIe.  Not part of the reference (its storages are host-side); for producers that write their samples
    on the GPU, and for `bench.py`'s `pdf_solve` block, which times `Estimate.construct_density` on resident samples."""

    def __init__(self):
        self._levels = {}
        self._level_parameters = []
        self._result_specification = []
        self._n_ops = {}
        self._level_versions = {}

    def save_global_data(self, result_format, level_parameters=None):
        self._result_specification = result_format
        self._level_parameters = level_parameters

    def set_level_samples(self, level_id, pairs):
        """pairs: torch CUDA tensor [M, n, 2] (level 0: [M, n, 1] or [M, n, 2] with an unused coarse column), float64."""
        assert pairs.is_cuda and pairs.dim() == 3 and pairs.dtype.is_floating_point
        if int(level_id) == 0 and pairs.shape[2] == 2:
            pairs = pairs[:, :, :1]
        self._levels[int(level_id)] = pairs.contiguous()
        self._level_versions[int(level_id)] = self._level_versions.get(int(level_id), 0) + 1

    def device_row(self, chunk_spec, stored_row):
        t = self._levels[int(chunk_spec.level_id)][int(stored_row)]
        sl = chunk_spec.chunk_slice
        return t if sl is None else t[sl.start:sl.stop]

    def sample_pairs_level(self, chunk_spec):
        """Host copy [M, n, 2|1] of a chunk (tests, host-evaluated trees)."""
        t = self._levels[int(chunk_spec.level_id)]
        sl = chunk_spec.chunk_slice
        return (t if sl is None else t[:, sl.start:sl.stop]).cpu().numpy()

    def sample_pairs(self):
        return [self.sample_pairs_level(ChunkSpec(level_id=l)) for l in sorted(self._levels)]

    def _level_chunks(self, level_id, n_samples=None):
        total = self._levels[int(level_id)].shape[1]
        if n_samples is not None:
            total = min(total, n_samples)
        yield ChunkSpec(chunk_id=0, chunk_slice=slice(0, total, 1), level_id=level_id)

    def load_result_format(self):
        return self._result_specification

    def save_result_format(self, res_spec):
        self._result_specification = res_spec

    def save_n_ops(self, n_ops):
        for level, (time, n_samples) in n_ops:
            self._n_ops.setdefault(level, 0)
            if n_samples != 0:
                self._n_ops[level] += time / n_samples

    def get_n_ops(self):
        return [self._n_ops[level] for level in sorted(self._n_ops.keys())]

    def n_finished(self):
        return np.array([self._levels[l].shape[1] for l in sorted(self._levels)], dtype=float)

    def get_level_ids(self):
        return sorted(self._levels.keys())

    def get_n_levels(self):
        return len(self._levels)

    def get_n_collected(self):
        return [int(self._levels[l].shape[1]) for l in sorted(self._levels)]

    def get_level_parameters(self):
        return self._level_parameters

    def unfinished_ids(self):
        return []

    def _not_stored(self, *args, **kwargs):
        raise NotImplementedError("DeviceMemory takes whole levels: set_level_samples")

    save_samples = save_scheduled_samples = _not_stored

    def load_scheduled_samples(self):
        return {}
