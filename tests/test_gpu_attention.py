"""GPU: the fused UAVAttentionExtractor forward (csrc/uavenv_attention.hip) against the PyTorch fp32 module
with the reference's architecture (agents/dqn/dqn.py:548-650), on real frame-stacked observations of the
environment (ghost slots and out-of-range sensors exercise the key mask).  Floating-point kernel: fp32 sums
in a different order than torch -> tolerance 2e-5 absolute on features of magnitude O(1)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module(n_stack):
    import torch
    from uavenv_amd.learner import AttentionFeatures
    torch.manual_seed(0)
    m = AttentionFeatures(n_stack).cuda().eval()
    with torch.no_grad():                       # non-trivial LayerNorm / bias parameters
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m


@pytest.mark.parametrize("n_stack,n", [(10, 20), (4, 50), (1, 10)])
def test_fused_attention_matches_torch_module(n_stack, n):
    import torch
    import uavenv_amd as U
    E = 300
    env = U.BatchedUAVEnv(E, num_sensors=n, pad_sensors=50, grid_size=(300, 300), seed=1, duty_cycle=50.0)
    fs = U.FrameStack(E, env.obs_dim, n_stack, env.device)
    stacked = fs.reset(env.reset())
    for _ in range(n_stack + 3):
        o, r, d = env.step_random()
        stacked = fs.step(o, d, None)
    m = _module(n_stack)
    fused = U.FusedAttentionFeatures(m, n_stack, env.device)
    x = stacked.clone()
    x[0] = 0.0                                   # a sample whose tokens are ALL masked (the unmask-everything branch)
    x[1, -150:] = 0.0
    x[1, -150 + 2] = 0.4                         # exactly one visible token
    with torch.no_grad():
        want = m(x)
    got = fused(x)
    err = (got - want).abs().max().item()
    assert torch.isfinite(got).all() and err <= 2e-5, err
    frac_masked = ((x[:, -150:].view(E, 50, 3)[:, :, 2] < 1e-6).float().mean().item())
    assert 0.2 < frac_masked < 1.0               # the mask is really exercised
    env.close()


@pytest.mark.parametrize("n_stack", [10, 4])
def test_weight_refresh_through_the_index_map_equals_the_piecewise_packing(n_stack):
    """FusedAttentionFeatures.refresh(module) -- the parameters concatenated + ONE gather through an index map built from the
    packing itself -- gives the block pack_attention_weights builds piece by piece, bit for bit, after the parameters changed
    in place (an optimiser step) and for another module."""
    import torch
    import uavenv_amd as U
    from uavenv_amd.attention import pack_attention_weights
    m = _module(n_stack)
    fused = U.FusedAttentionFeatures(m, n_stack, "cuda")
    addr = fused.weights.data_ptr()
    for _ in range(2):
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.1 * torch.randn_like(p))
        fused.refresh(m)
        assert fused.weights.data_ptr() == addr                           # same block: captured launches keep reading it
        assert torch.equal(fused.weights, pack_attention_weights(m.state_dict(), n_stack, "cuda"))
    other = _module(n_stack)
    with torch.no_grad():
        for p in other.parameters():
            p.mul_(1.5)
    fused.refresh(other)                                                  # another module: the map is rebuilt for its tensors
    assert torch.equal(fused.weights, pack_attention_weights(other.state_dict(), n_stack, "cuda"))
    fused.refresh(m.state_dict())                                         # a state dict: the piecewise path
    assert torch.equal(fused.weights, pack_attention_weights(m.state_dict(), n_stack, "cuda"))


@pytest.mark.parametrize("n_stack", [10, 4])
def test_folded_training_forward_equals_the_module_form_values_and_gradients(n_stack):
    """AttentionFeatures.forward (key projection folded into the query, value projection after the weighted mean, batched sensor
    projection) against forward_module (nn.Linear + nn.MultiheadAttention, the reference's form) on real observations: features and
    the gradients of every parameter under the same upstream gradient.  (in_proj_bias: the key bias' gradient is zero in theory and
    cancellation noise in the module form -- compared on the query and value thirds.)"""
    import torch
    import uavenv_amd as U
    E = 256
    env = U.BatchedUAVEnv(E, num_sensors=20, pad_sensors=50, grid_size=(300, 300), seed=2, duty_cycle=50.0)
    fs = U.FrameStack(E, env.obs_dim, n_stack, env.device)
    stacked = fs.reset(env.reset())
    for _ in range(n_stack + 2):
        o, r, d = env.step_random()
        stacked = fs.step(o, d, None)
    x = stacked.clone()
    x[0] = 0.0                                   # all tokens masked -> the unmask-everything branch
    m = _module(n_stack).train()
    up = torch.randn(E, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outs, grads = [], []
    for f in (m.forward, m.forward_module):
        m.zero_grad(set_to_none=True)
        y = f(x)
        (y * up).sum().backward()
        outs.append(y.detach().clone())
        grads.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
    assert torch.allclose(outs[0], outs[1], rtol=1e-4, atol=2e-5), float((outs[0] - outs[1]).abs().max())
    for name in grads[0]:
        a, b = grads[0][name], grads[1][name]
        if name == "attn.in_proj_bias":
            a, b = torch.cat([a[:64], a[128:]]), torch.cat([b[:64], b[128:]])
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, (name, float((a - b).abs().max()), scale)
    env.close()


@pytest.mark.parametrize("B,H,T", [(256, 4, 50), (5, 4, 50), (64, 8, 64), (33, 1, 7), (16, 2, 33)])
def test_attention_core_kernels_equal_the_pytorch_ops_forward_and_backward(B, H, T):
    """uavenv_attn_core_forward / _backward (scores, masked softmax, weighted mean of the tokens for one query per sample, one launch
    each way) against bmm + masked_fill + softmax + bmm and autograd through them: mix, and the gradients w.r.t. the folded query
    and the tokens under the same upstream gradient.  Floating-point kernels: fp32 sums in another order -> 1e-5 on O(1) values."""
    import torch
    from uavenv_amd.learner import _AttentionCore
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T)
    qk = torch.randn(B, H, 64, device="cuda", generator=g, requires_grad=True)
    kv = torch.randn(B, T, 64, device="cuda", generator=g, requires_grad=True)
    mask = torch.rand(B, T, device="cuda", generator=g) < 0.4
    mask[:, 0] = False                                        # never a whole row
    mask[0, 1:] = True                                        # one sample with a single visible token
    up = torch.randn(B, H, 64, device="cuda", generator=g)
    scores = torch.bmm(qk, kv.transpose(1, 2)).masked_fill(mask.unsqueeze(1), float("-inf"))
    want = torch.bmm(torch.softmax(scores, -1), kv)
    (want * up).sum().backward()
    gq, gk = qk.grad.clone(), kv.grad.clone()
    qk.grad = None; kv.grad = None
    got = _AttentionCore.apply(qk, kv, mask)
    (got * up).sum().backward()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), float((got - want).abs().max())
    assert torch.allclose(qk.grad, gq, rtol=1e-4, atol=2e-5), float((qk.grad - gq).abs().max())
    assert torch.allclose(kv.grad, gk, rtol=1e-4, atol=2e-5), float((kv.grad - gk).abs().max())
    assert bool((kv.grad[mask] == 0).all())                   # ignored tokens receive no gradient


@pytest.mark.parametrize("form", ["folded_fused_core", "folded_torch_core"])
def test_extractor_gradients_out_of_graph_replays_equal_eager_autograd_on_new_inputs(form):
    """The training step replays the extractor's forward + backward as a HIP graph.  Every parameter's gradient out of a replay --
    on inputs that CHANGED since the capture -- must be what eager autograd computes for them.  (The gradient of a bias handed to
    torch.baddbmm, a [B, T, E] -> [E] reduction, comes back stale from replays on this PyTorch / ROCm build while a first replay on
    unchanged inputs looks right: AttentionFeatures.forward carries the bias as a fourth always-one input instead.  The reference's form,
    forward_module -- nn.Linear / nn.MultiheadAttention over the [B x 50] tokens -- has the same stale gradients for every token-level
    bias (sensor.bias, the value third of attn.in_proj_bias: 12 800-row reductions); the learner replays only `forward`.)"""
    import torch
    m = _module(10).train()
    m.fused_core = form == "folded_fused_core"
    fwd = m.forward
    params = list(m.parameters())
    names = [n for n, _ in m.named_parameters()]
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(256, 1530, device="cuda", generator=gen)
    up = torch.randn(256, 128, device="cuda", generator=gen)
    flat = torch.zeros(sum(p.numel() for p in params), device="cuda")

    def step():
        for p in params:
            p.grad = None
        (fwd(x) * up).sum().backward()
        torch.cat([p.grad.reshape(-1) for p in params], out=flat)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for change in (lambda: x.mul_(0.5), lambda: x.add_(0.1), lambda: up.neg_()):
        change()
        for p in params:
            p.grad = None
        (fwd(x) * up).sum().backward()
        want = torch.cat([p.grad.reshape(-1) for p in params])
        g.replay()
        torch.cuda.synchronize()
        off = 0
        for name, p in zip(names, params):
            k = p.numel()
            a, b = flat[off:off + k], want[off:off + k]
            off += k
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-3), (form, name, float((a - b).abs().max()), float(b.abs().max()))
