"""Minimal `gymnasium.spaces` stand-in: only the two constructors the reference env calls
(uav_env.py:347 `spaces.Discrete(5)`, uav_env.py:354 `spaces.Box(low, high, dtype)`)."""
import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def contains(self, x):
        return 0 <= int(x) < self.n


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        self.low, self.high = low, high
        self.shape = low.shape
        self.dtype = np.dtype(dtype)
