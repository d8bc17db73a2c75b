"""Device-side frame stack with SB3 `VecFrameStack` semantics (agents/dqn/dqn.py:1278 wraps the env in
VecFrameStack(n_stack=4); UAVAttentionExtractor expects N_STACK frames of 153 floats, dqn.py:567-571).

The stacked observation never leaves the GPU: one small HIP kernel (uavenv_frame_stack) shifts each
environment's row by one frame in place and appends the new observation.
"""
import ctypes as C

import torch

from . import _native as N


class FrameStack:
    def __init__(self, num_envs, obs_dim, n_stack, device):
        self.E, self.D, self.k = int(num_envs), int(obs_dim), int(n_stack)
        if self.k * self.D > 2560:
            raise ValueError("n_stack * obs_dim must be <= 2560")
        self.device = torch.device(device)
        self.stacked = torch.zeros(self.E, self.k * self.D, dtype=torch.float32, device=self.device)
        self.terminal_stacked = torch.zeros_like(self.stacked)
        self.L = N.lib()

    @staticmethod
    def _p(t):
        return C.c_void_p(0 if t is None else t.data_ptr())

    def reset(self, obs):
        """SB3 StackedObservations.reset: zeros, newest frame = obs."""
        self.stacked.zero_()
        self.stacked[:, -self.D:] = obs
        return self.stacked

    def step(self, obs, done=None, terminal_obs=None):
        """obs float32 cuda [E, D]; done uint8 cuda [E] (envs that were auto-reset in this step);
        terminal_obs float32 cuda [E, D] (rows valid where done).  Returns the stacked tensor [E, k*D];
        `self.terminal_stacked` rows are valid where done."""
        for t, shape, dt in ((obs, (self.E, self.D), torch.float32), (done, (self.E,), torch.uint8),
                             (terminal_obs, (self.E, self.D), torch.float32)):
            if t is not None:
                assert t.is_cuda and t.is_contiguous() and tuple(t.shape) == shape and t.dtype == dt, (t.shape, t.dtype)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        N.check(self.L.uavenv_frame_stack(self._p(self.stacked), self._p(obs), self._p(done), self._p(terminal_obs),
                                          self._p(self.terminal_stacked if terminal_obs is not None else None),
                                          self.E, self.k, self.D, stream))
        return self.stacked
