"""Fused inference forward of the reference's UAVAttentionExtractor (agents/dqn/dqn.py:548-650) on the
frame-stacked observations of this environment: one HIP launch instead of ~15 eager PyTorch launches
(csrc/uavenv_attention.hip: 16 samples per workgroup, the shared-weight projections on the f32 MFMA).  Training still uses the PyTorch module; this is the acting path."""
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


def _frag(wt):
    """[K, N] (the TRANSPOSED weight: K inputs, N outputs) -> MFMA fragment order [N / 16 tiles][64 lanes][K / 4]: lane
    l = (k-group g = l >> 4, column c = l & 15) of tile nt holds wt[g * K/4 + t][16 * nt + c] for t = 0 .. K/4 - 1
    (csrc/uavenv_attention.hip: mfma_tile)."""
    K, n = wt.shape
    return wt.reshape(4, K // 4, n // 16, 16).permute(2, 0, 3, 1).reshape(-1)


def pack_attention_weights(state_dict, n_stack, device, out=None):
    """Flatten the extractor's parameters into the block csrc/uavenv_attention.hip expects (`Offsets` there): the six
    shared-weight projections in MFMA fragment order, the sensor projection as channel-pair rows, biases / LayerNorm
    parameters as they are."""
    def g(key):
        for name in _NAMES[key]:
            if name in state_dict:
                return state_dict[name].detach().to(device=device, dtype=torch.float32)
        raise KeyError(f"none of {_NAMES[key]} in state_dict")
    uav_w = g("uav_w")
    assert tuple(uav_w.shape) == (64, 3 * n_stack), (uav_w.shape, n_stack)
    k0 = 16 * ((3 * n_stack + 15) // 16)                  # the encoder's K, zero padded to whole k-steps per lane group
    uav_t = torch.zeros(k0, 64, dtype=torch.float32, device=device)
    uav_t[:3 * n_stack] = uav_w.t()
    in_w, in_b = g("in_w"), g("in_b")
    wq, wk, wv = in_w[:64], in_w[64:128], in_w[128:]
    sens_w, sens_b = g("sens_w"), g("sens_b")
    # [channel pair 32][sw0 pair, sw1 pair, sw2 pair, bias pair]
    swp = torch.stack([sens_w[:, 0], sens_w[:, 1], sens_w[:, 2], sens_b], 0).reshape(4, 32, 2).permute(1, 0, 2)
    fuse_t = g("fuse_w").t()
    parts = [_frag(uav_t), g("uav_b"), g("ln1_g"), g("ln1_b"), swp,
             _frag(wq.t()), in_b[:64],
             torch.cat([_frag(wk[16 * h:16 * h + 16]) for h in range(4)]),       # per head: q_h -> the 64 embedding channels
             _frag(wv.t()), in_b[128:], _frag(g("out_w").t()), g("out_b"), g("ln2_g"), g("ln2_b"),
             _frag(fuse_t[:64]), _frag(fuse_t[64:]), g("fuse_b")]
    pieces = [p.contiguous().reshape(-1) for p in parts]
    if out is not None:                                   # refresh an existing block in place (same address: graph replays see it)
        assert out.numel() == sum(p.numel() for p in pieces)
        torch.cat(pieces, out=out)
        return out
    flat = torch.cat(pieces).contiguous()
    assert flat.numel() == N.lib().uavenv_attention_weight_floats(n_stack), (flat.numel(), N.lib().uavenv_attention_weight_floats(n_stack))
    return flat


class FusedAttentionFeatures:
    def __init__(self, module_or_state_dict, n_stack, device):
        sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
        self.n_stack, self.device = int(n_stack), torch.device(device)
        self.weights = pack_attention_weights(sd, self.n_stack, self.device)
        self.L = N.lib()
        self._gather = None

    def _build_gather(self, module):
        """The packing as ONE gather: the block is a fixed permutation (plus zero padding) of the module's parameters, so it is
        read off once by packing tensors that hold their own positions in a flat copy of the parameters (position + 1; the
        padding's zeros then point at slot 0, which holds 0.0)."""
        sd = module.state_dict()
        keys = [name for key in _NAMES for name in _NAMES[key] if name in sd]
        tensors = [dict(module.named_parameters())[k] for k in keys]
        sizes = [t.numel() for t in tensors]
        assert sum(sizes) + 1 < 2 ** 24                        # (positions travel through float32 exactly)
        fake, off = {}, 1
        for k, t in zip(keys, tensors):
            fake[k] = torch.arange(off, off + t.numel(), dtype=torch.float32, device=self.device).view_as(t)
            off += t.numel()
        idx = pack_attention_weights(fake, self.n_stack, self.device).round().to(torch.int64)
        flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self._gather = (tensors, flat, idx, id(module))

    def refresh(self, module_or_state_dict):
        """Re-pack the (updated) parameters into the same weight block (same address: graph replays see it).  From a module: two
        launches -- the parameters concatenated into a flat buffer, one gather through the packing's index map -- where packing
        piece by piece is ~25 small copies; from a state dict: piece by piece."""
        if hasattr(module_or_state_dict, "named_parameters"):
            if self._gather is None or self._gather[3] != id(module_or_state_dict):
                self._build_gather(module_or_state_dict)
            tensors, flat, idx, _ = self._gather
            torch.cat([t.detach().reshape(-1) for t in tensors], out=flat[1:])
            torch.index_select(flat, 0, idx, out=self.weights)
            return
        pack_attention_weights(module_or_state_dict, self.n_stack, self.device, out=self.weights)

    def __call__(self, obs):
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.n_stack * 153
        out = torch.empty(obs.shape[0], 128, dtype=torch.float32, device=obs.device)
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        N.check(self.L.uavenv_attention_features(C.c_void_p(obs.data_ptr()), C.c_void_p(self.weights.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), obs.shape[0], self.n_stack, stream))
        return out
