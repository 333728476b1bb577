"""Result-format and chunk descriptors (reference interface: mlmc/quantity/quantity_spec.py:6-28)."""
from dataclasses import dataclass
from typing import List, Tuple, Union

import numpy as np


@dataclass(eq=False)
class QuantitySpec:
    name: str
    unit: str
    shape: Tuple[int, int]
    times: List[float]
    locations: Union[List[str], List[Tuple[float, float, float]]]

    def __eq__(self, other):
        return (self.name, self.unit) == (other.name, other.unit) \
            and np.array_equal(self.shape, other.shape) \
            and np.array_equal(self.times, other.times) \
            and not (set(self.locations) - set(other.locations))

    __hash__ = None


@dataclass
class ChunkSpec:
    chunk_id: int = None
    chunk_slice: slice = None
    level_id: int = None
