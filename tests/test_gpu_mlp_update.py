"""GPU: the hand-written DQN update of the MLP policy (csrc/uavenv_learner.hip, uavenv_amd/mlp_update.py) against PyTorch --
the small-batch MFMA GEMM with its operand transforms against torch.matmul, and whole updates (forward of both networks,
smooth-L1 TD loss, backward, clip_grad_norm_, Adam) against torch autograd + torch.optim.Adam on the same batches.
Floating-point kernels: fp32 sums in another order than the library's (K split over the wavefronts of a workgroup) ->
tolerances 1e-4 relative / 1e-5 absolute on O(1) values."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mods():
    import torch
    import uavenv_amd as U
    from uavenv_amd import _native as N
    from uavenv_amd import learner as LR
    from uavenv_amd.mlp_update import FusedMLPUpdate
    return torch, U, N, LR, FusedMLPUpdate


def _product(N, A, B, Cm, M, Nn, K, a_sm, a_sk, b_sk, b_sn, flags=0, bias=None, mask=None, row_sum=None, sumsq=None):
    dp = lambda t: None if t is None else t.data_ptr()
    return N.UavGemm(A=dp(A), B=dp(B), C=dp(Cm), bias=dp(bias), a_mask=dp(mask), row_sum=dp(row_sum), M=M, N=Nn, K=K, flags=flags,
                     a_sm=a_sm, a_sk=a_sk, b_sk=b_sk, b_sn=b_sn, ldc=Cm.stride(0), sumsq=dp(sumsq))


def _gemm(torch, N, A, B, Cm, M, Nn, K, a_sm, a_sk, b_sk, b_sn, flags=0, bias=None, mask=None, row_sum=None, second=None, sumsq=None):
    g = _product(N, A, B, Cm, M, Nn, K, a_sm, a_sk, b_sk, b_sn, flags, bias, mask, row_sum, sumsq)
    rc = N.lib().uavenv_gemm_f32(C.byref(g), None if second is None else C.byref(second), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


@pytest.mark.parametrize("M,Nn,K", [(256, 512, 612), (256, 5, 256), (64, 70, 33), (16, 64, 16), (5, 256, 256), (512, 612, 256), (256, 256, 5),
                                    (304, 960, 100)])    # 19 x 15 = 285 tiles: two tiles per workgroup, the last one half idle
def test_small_batch_gemm_all_three_layouts_match_torch(M, Nn, K):
    torch, U, N, LR, F = _mods()
    g = torch.Generator(device="cuda").manual_seed(M * 1000 + Nn)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    # forward layout: A [M x K] and W [N x K] contiguous along k, bias, relu on A
    A, W, bias = r(M, K), r(Nn, K), r(Nn)
    Cm = torch.full((M, Nn), 7.0, device="cuda")             # (every product overwrites its whole output)
    _gemm(torch, N, A, W, Cm, M, Nn, K, K, 1, 1, K, flags=N.GEMM_BIAS | N.GEMM_A_RELU, bias=bias)
    want = torch.relu(A).double() @ W.double().t() + bias.double()
    assert torch.allclose(Cm.double(), want, rtol=1e-4, atol=1e-4 * K ** 0.5), float((Cm.double() - want).abs().max())
    # input-gradient layout: A [M x K] contiguous along k, B [K x N] contiguous along n, mask on A
    Bm, Z = r(K, Nn), r(M, K)
    _gemm(torch, N, A, Bm, Cm, M, Nn, K, K, 1, Nn, 1, flags=N.GEMM_A_MASK, mask=Z)
    want = (A * (Z > 0)).double() @ Bm.double()
    assert torch.allclose(Cm.double(), want, rtol=1e-4, atol=1e-4 * K ** 0.5)
    # weight-gradient layout: A(m, k) = S[k][m] (transposed read), B [K x N] contiguous along n with relu, row sums of A
    # ... and the sums of squares of everything written, one per (16-row tile, 32-column group), every entry exactly once
    S, rs = r(K, M), torch.full((M,), 3.0, device="cuda")
    nsq = N.gemm_sumsq_count(M, Nn)
    sq = torch.full((nsq + 3,), -1.0, device="cuda")
    _gemm(torch, N, S, Bm, Cm, M, Nn, K, 1, M, Nn, 1, flags=N.GEMM_B_RELU | N.GEMM_ROWSUM | N.GEMM_SUMSQ, row_sum=rs, sumsq=sq)
    want = S.double().t() @ torch.relu(Bm).double()
    assert torch.allclose(Cm.double(), want, rtol=1e-4, atol=1e-4 * K ** 0.5)
    assert torch.allclose(rs.double(), S.double().sum(0), rtol=1e-4, atol=1e-4 * K ** 0.5)
    assert bool((sq[nsq:] == -1).all()) and bool((sq[:nsq] >= 0).all())
    groups = (Nn + 31) // 32
    pad = torch.zeros(-(-M // 16) * 16, groups * 32, dtype=torch.float64, device="cuda")
    pad[:M, :Nn] = Cm.double() ** 2
    want_sq = pad.view(-1, 16, groups, 32).sum((1, 3))
    rpad = torch.zeros(-(-M // 16) * 16, dtype=torch.float64, device="cuda")
    rpad[:M] = rs.double() ** 2
    want_sq[:, 0] += rpad.view(-1, 16).sum(1)                # the row sums (a bias gradient) count in their tile's first group
    assert torch.allclose(sq[:nsq].double().view(-1, groups), want_sq, rtol=1e-4, atol=1e-6), float((sq[:nsq].double().view(-1, groups) - want_sq).abs().max())
    # a strided output (a sub-block of a wider matrix): nothing outside it is touched
    wide = torch.ones(M, Nn + 7, device="cuda")
    sub = wide[:, 3:3 + Nn]
    _gemm(torch, N, A, W, sub, M, Nn, K, K, 1, 1, K)
    assert torch.allclose(sub.double(), A.double() @ W.double().t(), rtol=1e-4, atol=1e-4 * K ** 0.5)
    assert bool((wide[:, :3] == 1).all()) and bool((wide[:, 3 + Nn:] == 1).all())
    # two independent products of different layouts in ONE launch (a weight gradient and an input gradient, as the update pairs them)
    C1, C2 = torch.zeros(M, Nn, device="cuda"), torch.zeros(M, K, device="cuda")
    W2 = r(Nn, K)
    second = _product(N, Cm, W2, C2, M, K, Nn, Nn, 1, K, 1)                  # C2[m][k] = sum_n Cm[m][n] W2[n][k]
    _gemm(torch, N, S, Bm, C1, M, Nn, K, 1, M, Nn, 1, second=second)
    assert torch.allclose(C1.double(), S.double().t() @ Bm.double(), rtol=1e-4, atol=1e-4 * K ** 0.5)
    assert torch.allclose(C2.double(), Cm.double() @ W2.double(), rtol=1e-4, atol=2e-4 * (Nn * K) ** 0.5)


def test_gemm_rejects_bad_arguments():
    torch, U, N, LR, F = _mods()
    a = torch.zeros(16, 16, device="cuda")
    L = N.lib()
    ok = _product(N, a, a, a, 16, 16, 16, 16, 1, 1, 16)
    assert L.uavenv_gemm_f32(C.byref(_product(N, a, a, a, 16, 16, 16, 16, 2, 1, 16)), None, None) == N.E_INVALID      # no unit stride in A
    assert L.uavenv_gemm_f32(C.byref(_product(N, a, a, a, 16, 16, 16, 16, 1, 1, 16, flags=N.GEMM_BIAS)), None, None) == N.E_INVALID   # bias flag, no bias
    assert L.uavenv_gemm_f32(C.byref(_product(N, None, a, a, 16, 16, 16, 16, 1, 1, 16)), None, None) == N.E_INVALID
    assert L.uavenv_gemm_f32(C.byref(ok), C.byref(_product(N, a, None, a, 16, 16, 16, 16, 1, 1, 16)), None) == N.E_INVALID    # a bad second product
    assert L.uavenv_gemm_f32(None, None, None) == N.E_INVALID


def _batch(torch, B, K0, seed, n_invalid=9):
    g = torch.Generator(device="cuda").manual_seed(seed)
    b = dict(obs=torch.rand(B, K0, device="cuda", generator=g), next_obs=torch.rand(B, K0, device="cuda", generator=g),
             action=torch.randint(0, 5, (B,), device="cuda", generator=g), reward=torch.randn(B, device="cuda", generator=g) * 3,
             valid=torch.ones(B, dtype=torch.bool, device="cuda"))
    b["valid"][:n_invalid] = False
    return b


@pytest.mark.parametrize("arch,B", [((512, 512, 256), 256), ((64, 32), 128), ((48,), 64)])
def test_fused_update_matches_autograd_and_adam(arch, B):
    """Five updates on five batches, the target network synchronised in between: losses, gradients (first update), parameters
    and Adam moments (all updates) against torch.  The reference's shape: 612 -> 512 -> 512 -> 256 -> 5, batch 256."""
    import copy
    torch, U, N, LR, F = _mods()
    torch.manual_seed(3)
    q = LR.QNetwork(153, 4, arch).cuda()
    qt = copy.deepcopy(q).requires_grad_(False)
    with torch.no_grad():
        for p_ in qt.parameters():
            p_.add_(0.01 * torch.randn_like(p_))            # a target network that differs from the online one
    q_ref, qt_ref = copy.deepcopy(q), copy.deepcopy(qt)
    gamma, max_norm, scale, lr = 0.97, 0.02, 0.5, 3e-3      # a clip threshold that bites (the first gradient norm is ~0.08)
    # (eps 1e-5 instead of Adam's default 1e-8 on BOTH sides: with the default, elements whose gradient is of eps' size turn a
    #  1e-9 difference in g -- summation order -- into a visible fraction of lr, and the comparison measures that, not the kernels;
    #  the default eps runs in tests/test_gpu_learner.py's hand-arithmetic and graph-replay tests)
    eps = 1e-5
    upd = F(q, qt, B, gamma, max_norm, reward_scale=scale, lr=lr, eps=eps)
    opt = torch.optim.Adam(q_ref.parameters(), lr=lr, eps=eps)
    for it in range(5):
        batch = _batch(torch, B, 612, 100 + it)
        loss_ref = LR.td_loss(q_ref, qt_ref, batch, gamma, scale)
        opt.zero_grad()
        loss_ref.backward()
        gref = [p_.grad.detach().clone() for p_ in q_ref.parameters()]
        torch.nn.utils.clip_grad_norm_(q_ref.parameters(), max_norm)
        opt.step()
        upd.backward(batch)
        if it == 0:
            total = torch.sqrt(sum((g_ ** 2).sum() for g_ in gref))
            assert float(total) > max_norm                  # (so the clipping is exercised)
            mine = [t for pair in zip(upd.gw, upd.gb) for t in pair]
            for a, b_ in zip(mine, gref):
                assert torch.allclose(a, b_, rtol=1e-4, atol=1e-6), float((a - b_).abs().max())
        norm2 = float((upd.grad.double() ** 2).sum())
        upd.apply(grads_changed=(it == 1))                   # (it 1: the norm recomputed from the flat gradients, as after an all-reduce)
        assert float(upd.scalars[N.UPD_NORM2]) == pytest.approx(norm2, rel=1e-5)       # (else: from the products' partial sums)
        assert float(upd.loss) == pytest.approx(float(loss_ref.detach()), rel=1e-5)
        for a, b_ in zip(q.parameters(), q_ref.parameters()):
            diff = (a.detach() - b_.detach()).abs()
            assert float(diff.max()) <= 2e-3 * lr * (it + 1) and float(diff.mean()) <= 1e-7, (it, float(diff.max()), float(diff.mean()))
        if it == 2:
            upd.sync_target()
            qt_ref.load_state_dict(q_ref.state_dict())
            for a, b_ in zip(qt.parameters(), q.parameters()):
                assert torch.equal(a, b_)
    assert upd.step_count == 5
    st = opt.state_dict()["state"]
    m_ref = torch.cat([st[i]["exp_avg"].reshape(-1) for i in range(len(st))])
    v_ref = torch.cat([st[i]["exp_avg_sq"].reshape(-1) for i in range(len(st))])
    assert torch.allclose(upd.exp_avg, m_ref, rtol=1e-3, atol=1e-7) and torch.allclose(upd.exp_avg_sq, v_ref, rtol=1e-3, atol=1e-10)
    # acting through the torch module sees the updated parameters (they are views of the flat buffer)
    x = torch.rand(7, 612, device="cuda")
    with torch.no_grad():
        assert torch.allclose(q(x), q_ref(x), rtol=1e-3, atol=1e-4)
