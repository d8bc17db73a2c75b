"""Observation / action space objects.  Real `gymnasium.spaces` when gymnasium is importable (so
SB3 accepts them unchanged), otherwise minimal look-alikes with the attributes callers read
(.n, .shape, .dtype, .low, .high, .sample(), .contains())."""
import numpy as np

try:  # pragma: no cover - depends on the box
    from gymnasium import spaces as _gs
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium is not installed in the build container
    _gs = None
    HAVE_GYMNASIUM = False


class _Discrete:
    def __init__(self, n, seed=None):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return int(self._rng.integers(self.n))

    def contains(self, x):
        try:
            return 0 <= int(x) < self.n
        except Exception:
            return False

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return f"Discrete({self.n})"


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


def Discrete(n):
    return _gs.Discrete(n) if HAVE_GYMNASIUM else _Discrete(n)


def Box(low, high, shape=None, dtype=np.float32):
    if HAVE_GYMNASIUM:
        return _gs.Box(low=low, high=high, shape=shape, dtype=dtype)
    return _Box(low, high, shape, dtype)
