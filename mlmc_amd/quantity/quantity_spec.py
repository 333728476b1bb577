"""Result-format and chunk descriptors (reference interface: mlmc/quantity/quantity_spec.py:6-28)."""
import numpy as np


class QuantitySpec:
    """One named quantity of a simulation result: `shape` values per (time, location)."""

    __slots__ = ("name", "unit", "shape", "times", "locations")
    __hash__ = None

    def __init__(self, name, unit, shape, times, locations):
        self.name, self.unit = name, unit
        self.shape = shape                 # (rows, columns) of the value at one time and location
        self.times = times                 # list of floats
        self.locations = locations         # list of names, or of (x, y, z) points

    def _same_axes(self, other):
        return np.array_equal(self.shape, other.shape) and np.array_equal(self.times, other.times)

    def __eq__(self, other):
        """Equal when name, unit, shape and times agree and every location of self is known to other."""
        if (self.name, self.unit) != (other.name, other.unit) or not self._same_axes(other):
            return False
        return set(self.locations) <= set(other.locations)

    def __repr__(self):
        return "QuantitySpec(name={!r}, unit={!r}, shape={!r}, times={!r}, locations={!r})".format(
            self.name, self.unit, self.shape, self.times, self.locations)


class ChunkSpec:
    """Which samples a storage is asked for: a chunk of a level (all three fields may be left open)."""

    __slots__ = ("chunk_id", "chunk_slice", "level_id")

    def __init__(self, chunk_id=None, chunk_slice=None, level_id=None):
        self.chunk_id, self.chunk_slice, self.level_id = chunk_id, chunk_slice, level_id

    def _key(self):
        return (self.chunk_id, self.chunk_slice, self.level_id)

    def __eq__(self, other):
        return isinstance(other, ChunkSpec) and self._key() == other._key()

    def __repr__(self):
        return "ChunkSpec(chunk_id={!r}, chunk_slice={!r}, level_id={!r})".format(*self._key())
