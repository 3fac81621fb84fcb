"""Observation / action spaces.  Uses gymnasium.spaces when it is importable (the
reference's env and RLlib expect those classes); otherwise a minimal stand-in
with the attributes the reference touches (.shape, .dtype, .low, .high,
.sample(), .contains(), Dict item access / .spaces)."""
from __future__ import annotations

import numpy as np

try:                                            # pragma: no cover - depends on the install
    from gymnasium import spaces as _gs
    Box = _gs.Box
    Dict = _gs.Dict
    HAVE_GYMNASIUM = True
except Exception:                               # gymnasium is absent in the build container
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict:
        def __init__(self, spaces=None, **kw):
            self.spaces = dict(spaces or {})
            self.spaces.update(kw)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

        def items(self):
            return self.spaces.items()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

        def __repr__(self):
            return f"Dict({self.spaces})"
