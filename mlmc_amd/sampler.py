"""Adaptive sampling loop over device-generated samples.

`DeviceSampler` has the scheduling interface of the reference's `Sampler` (mlmc/sampler.py:9-290) for the one
simulation the device can run itself, `SynthSimulation` (sim/synth_device.py): levels are grown towards the sample
counts `estimate_n_samples_for_target_variance` asks for, the new samples "L{level:02d}_S{index:07d}" are generated in
HBM when an estimate first reads them, and the loop of the reference's test/test_run.py:93-103

    variances, n_ops = estimator.estimate_diff_vars_regression(sampler._n_scheduled_samples)
    n_estimated = estimate_n_samples_for_target_variance(target_var, variances, n_ops, n_levels=sampler.n_levels)
    while not sampler.process_adding_samples(n_estimated, sleep, add_coef): ...

runs without a sample ever crossing PCIe.  The bookkeeping follows the reference with its `OneProcessPool`
(mlmc/sampling_pool.py:203-290): a scheduled sample is computed at once but reaches the storage only with the next
`ask_sampling_pool_for_samples`, so the storage may lag one round behind the scheduled counts -- the trajectory of
scheduled / finished counts is the reference's (tests/golden/G9_sampler_loop.json).
"""
import time

import numpy as np


class DeviceSampler:
    ADDING_SAMPLES_TIMEOUT = 1e-15

    def __init__(self, sample_storage, level_parameters=None):
        """sample_storage: a storage whose levels grow on request -- `set_n_samples(counts)`, `n_finished()`,
        `get_level_parameters()` (sim.synth_device.SynthDeviceStorage); level_parameters: checked against the storage's
        when given (the reference passes them to the simulation factory, here the storage already holds the steps)."""
        self.sample_storage = sample_storage
        own = [list(np.ravel(p)) for p in sample_storage.get_level_parameters()]
        if level_parameters is not None:
            assert [list(np.ravel(p)) for p in level_parameters] == own, "level_parameters differ from the storage's"
        self._n_levels = len(own)
        self._n_target_samples = np.zeros(self._n_levels)
        self._n_scheduled_samples = np.array(sample_storage.n_finished(), dtype=float)   # samples created so far
        self._n_pending = np.zeros(self._n_levels)           # computed by the "pool", not yet handed to the storage

    @property
    def n_levels(self):
        return self._n_levels

    @property
    def n_finished_samples(self):
        return self.sample_storage.n_finished()

    def sample_range(self, n0, nL):
        """Geometric sequence of n_levels counts from n0 down to nL (sampler.py:82-90)."""
        return np.round(np.exp2(np.linspace(np.log2(n0), np.log2(nL), self.n_levels))).astype(int)

    def set_initial_n_samples(self, n_samples=None):
        """Target counts per level: explicit list, (n0, nL) -> sample_range, (n0,) -> (n0, 10) (sampler.py:92-112)."""
        n_samples = np.atleast_1d([100, 10] if n_samples is None else n_samples)
        if len(n_samples) == 1:
            n_samples = np.array([n_samples[0], 10])
        if len(n_samples) == 2:
            n_samples = self.sample_range(*n_samples)
        self._n_target_samples = n_samples

    def schedule_samples(self, timeout=None):
        """Create the samples missing to the targets (sampler.py:122-150): their ids continue every level's sequence."""
        self.ask_sampling_pool_for_samples(timeout=timeout)
        plan = np.asarray(self._n_target_samples, dtype=float) - self._n_scheduled_samples
        for level_id, n_new in enumerate(plan):
            n_new = max(int(n_new), 0)
            self._n_scheduled_samples[level_id] += n_new
            self._n_pending[level_id] += n_new

    def ask_sampling_pool_for_samples(self, sleep=0, timeout=None):
        """Hand finished samples to the storage (sampler.py:160-183); -> number of running simulations (0; 1 when the
        call was told not to wait, timeout <= 0, as in the reference)."""
        if timeout is not None and timeout <= 0:
            return 1
        if np.any(self._n_pending > 0):
            self._n_pending[:] = 0
            self.sample_storage.set_n_samples([int(v) for v in self._n_scheduled_samples])
        if sleep:
            time.sleep(sleep)
        return 0

    def process_adding_samples(self, n_estimated, sleep=0, add_coeff=0.1, timeout=ADDING_SAMPLES_TIMEOUT):
        """One round of the adaptive loop (sampler.py:195-229): move every level a fraction `add_coeff` of the way from
        its scheduled count to the estimate -- or all the way once that fraction of the estimate covers the rest.
        -> True when no level is left below its estimate."""
        self.ask_sampling_pool_for_samples(timeout=timeout)
        n_estimated = np.asarray(n_estimated)
        have = self.l_scheduled_samples()
        missing = n_estimated - have
        step = np.where(n_estimated * add_coeff > missing, n_estimated, have + missing * add_coeff)
        n_scheduled = np.ceil(np.where(n_estimated < have, have, step))
        behind = np.where(np.greater(n_estimated, n_scheduled))[0]
        self.set_scheduled_and_wait(n_scheduled, behind, sleep, timeout=timeout)
        return bool(np.all(n_estimated[behind] == n_scheduled[behind]))

    def set_scheduled_and_wait(self, n_scheduled, greater_items, sleep, fin_sample_coef=0.5, timeout=1e-7):
        """Schedule up to n_scheduled and collect until `fin_sample_coef` of them are in the storage on every level of
        `greater_items` (sampler.py:231-252)."""
        self.set_level_target_n_samples(n_scheduled)
        self.schedule_samples(timeout=timeout)
        n_finished = self.n_finished_samples
        while np.any(n_finished[greater_items] < fin_sample_coef * n_scheduled[greater_items]):
            time.sleep(sleep)
            self.ask_sampling_pool_for_samples(timeout=timeout)
            n_finished = self.n_finished_samples

    def set_level_target_n_samples(self, n_samples):
        """Raise (never lower) the targets (sampler.py:254-261)."""
        self._n_target_samples = np.asarray(self._n_target_samples, dtype=float)
        for level, n in enumerate(n_samples):
            self._n_target_samples[level] = max(self._n_target_samples[level], n)

    def l_scheduled_samples(self):
        return self._n_scheduled_samples

    def renew_failed_samples(self):
        """Generated samples do not fail (SynthSimulation with nan_fraction = 0)."""
