"""Test-only stand-in for the `gymnasium` package (NOT installed in the build container).

Purpose: let `tests/golden/make_golden.py` import the *real* reference environment
(`/root/reference/src/environment/uav_env.py`, which does `import gymnasium as gym` and
`from gymnasium import spaces`, uav_env.py:6-7) in the build container so golden vectors can be
captured from it.  This is the build's own code: it only provides the three names the reference
touches (`Env`, `spaces.Discrete`, `spaces.Box`).  It is never imported by the product package and
is never needed on the GPU box (the fixtures travel, the reference does not).

`Env.reset(seed=...)` mirrors gymnasium 1.x seeding semantics that the reference relies on at
uav_env.py:406 (`super().reset(seed=seed)`): a non-None seed re-creates
`self.np_random = numpy.random.Generator(PCG64(SeedSequence(seed)))`; `seed=None` keeps the
current generator (creating an unseeded one on first use).
"""
import numpy as np

from . import spaces  # noqa: F401

__version__ = "0.0-stub"


class Env:
    metadata = {}
    render_mode = None
    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    def close(self):
        pass
