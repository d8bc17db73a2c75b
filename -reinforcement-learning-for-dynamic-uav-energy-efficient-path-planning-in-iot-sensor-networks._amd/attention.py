"""Fused inference forward of the reference's UAVAttentionExtractor (agents/dqn/dqn.py:548-650) on the
frame-stacked observations of this environment: one HIP launch instead of ~15 eager PyTorch launches
(csrc/uavenv_attention.hip).  Training still uses the PyTorch module; this is the acting path."""
import ctypes as C

import torch

from . import _native as N

# parameter names of the reference module (dqn.py:583-602) and of examples/train_dqn.py:AttentionFeatures
_NAMES = {
    "uav_w": ("uav_encoder.0.weight", "uav.0.weight"), "uav_b": ("uav_encoder.0.bias", "uav.0.bias"),
    "ln1_g": ("uav_encoder.1.weight", "uav.1.weight"), "ln1_b": ("uav_encoder.1.bias", "uav.1.bias"),
    "sens_w": ("sensor_proj.weight", "sensor.weight"), "sens_b": ("sensor_proj.bias", "sensor.bias"),
    "in_w": ("cross_attn.in_proj_weight", "attn.in_proj_weight"), "in_b": ("cross_attn.in_proj_bias", "attn.in_proj_bias"),
    "out_w": ("cross_attn.out_proj.weight", "attn.out_proj.weight"), "out_b": ("cross_attn.out_proj.bias", "attn.out_proj.bias"),
    "ln2_g": ("attn_norm.weight", "norm.weight"), "ln2_b": ("attn_norm.bias", "norm.bias"),
    "fuse_w": ("fusion.0.weight", "fuse.0.weight"), "fuse_b": ("fusion.0.bias", "fuse.0.bias"),
}


def pack_attention_weights(state_dict, n_stack, device, out=None):
    """Flatten the extractor's parameters into the block csrc/uavenv_attention.hip expects (weights transposed
    so that lane j of a wavefront reads consecutive addresses)."""
    def g(key):
        for name in _NAMES[key]:
            if name in state_dict:
                return state_dict[name].detach().to(device=device, dtype=torch.float32)
        raise KeyError(f"none of {_NAMES[key]} in state_dict")
    uav_w = g("uav_w")
    assert tuple(uav_w.shape) == (64, 3 * n_stack), (uav_w.shape, n_stack)
    in_w, in_b = g("in_w"), g("in_b")
    wq, wk, wv = in_w[:64], in_w[64:128], in_w[128:]
    sens_w = g("sens_w")
    parts = [uav_w.t(), g("uav_b"), g("ln1_g"), g("ln1_b"), sens_w[:, 0], sens_w[:, 1], sens_w[:, 2], g("sens_b"),
             wq.t(), in_b[:64], wk, in_b[64:128], wv.t(), in_b[128:], g("out_w").t(), g("out_b"), g("ln2_g"), g("ln2_b"),
             g("fuse_w").t(), g("fuse_b")]
    pieces = [p.contiguous().reshape(-1) for p in parts]
    if out is not None:                                   # refresh an existing block in place (same address: graph replays see it)
        assert out.numel() == sum(p.numel() for p in pieces)
        torch.cat(pieces, out=out)
        return out
    flat = torch.cat(pieces).contiguous()
    assert flat.numel() == N.lib().uavenv_attention_weight_floats(n_stack)
    return flat


class FusedAttentionFeatures:
    def __init__(self, module_or_state_dict, n_stack, device):
        sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
        self.n_stack, self.device = int(n_stack), torch.device(device)
        self.weights = pack_attention_weights(sd, self.n_stack, self.device)
        self.L = N.lib()

    def refresh(self, module_or_state_dict):
        """Re-pack the (updated) parameters into the same weight block: ~25 small device copies, no host round trip."""
        sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
        pack_attention_weights(sd, self.n_stack, self.device, out=self.weights)

    def __call__(self, obs):
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.n_stack * 153
        out = torch.empty(obs.shape[0], 128, dtype=torch.float32, device=obs.device)
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        N.check(self.L.uavenv_attention_features(C.c_void_p(obs.data_ptr()), C.c_void_p(self.weights.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), obs.shape[0], self.n_stack, stream))
        return out
