"""One DQN gradient step of the reference's MLP policy as 11 HIP launches (csrc/uavenv_learner.hip) instead of ~60 PyTorch
ones: what stable-baselines3's `DQN.train()` does per gradient step for `MlpPolicy` with `net_arch=[512, 512, 256]`
(agents/dqn/dqn.py:1077-1099) -- forward of the online and the target network, smooth-L1 TD loss, backward, clip_grad_norm_,
Adam -- on one sampled batch.  torch is plumbing here (device buffers, the modules whose parameters become views of one flat
buffer so that acting keeps using them); every product, the loss, the clipping and the optimiser step run in the library.

Why: the update is 1.45 GFLOP (9 us at the f32 MFMA rate) that PyTorch spends 337 us on -- a dozen library GEMMs whose
256-row outputs are 16 macro tiles for 256 CUs, and ~45 small launches around them.  At the reference's update ratio (one
update per 16 transitions) the update is the whole cost of training.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _native as N


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


class FusedMLPUpdate:
    """Owns the flat parameter / gradient / Adam-state buffers of `q` (and `q_target`) and runs updates on sampled batches.

    q, q_target: QNetwork (learner.py): `head` = Linear, ReLU, ..., Linear.  The heads' parameters are re-pointed to views of
    `self.flat` / `self.flat_target` (same values), so `q(x)` for acting sees every update.  With the Flatten extractor that is the
    whole network; with a features extractor in front (the attention extractor) the caller runs the extractor, hands its output in
    as `obs` / `next_obs`, continues the backward pass from `self.dx0` with autograd (`input_grad=True`) and lets this object
    optimise the extractor's parameters as well (`extra_params`): learner.py's "hybrid" update.
    """

    def __init__(self, q, q_target, batch_size, gamma, max_grad_norm, reward_scale=1.0, betas=(0.9, 0.999), eps=1e-8, lr=1e-3,
                 input_grad=False, extra_params=(), extra_target_params=()):
        self.L = N.lib()
        self.q, self.q_target = q, q_target
        # input_grad: also produce `self.dx0`, the gradient w.r.t. the first layer's input -- a features extractor in front of
        # the layers (the attention extractor) continues the backward pass from it with autograd
        self.input_grad = bool(input_grad)
        # extra_params: parameters OUTSIDE the layers (that extractor's) which this object optimises too -- they become views of
        # the same flat buffers; after autograd has produced their gradients, `collect_extra_grads` concatenates them into the flat
        # gradient buffer (one launch), so that ONE clip + Adam launch covers the whole network
        self.extra = list(extra_params)
        # extra_target_params: the target network's copies of them, in the same order: they join `flat_target`, so that the hard
        # target update stays ONE contiguous copy
        self.extra_t = list(extra_target_params)
        assert not self.extra_t or [p.shape for p in self.extra_t] == [p.shape for p in self.extra]
        self.layers = [m for m in q.head if isinstance(m, nn.Linear)]
        self.layers_t = [m for m in q_target.head if isinstance(m, nn.Linear)]
        assert len(self.layers) >= 1 and all(isinstance(m, (nn.Linear, nn.ReLU)) for m in q.head)
        self.dev = self.layers[0].weight.device
        assert self.dev.type == "cuda", "the fused update is a HIP path: there is no CPU form"
        self.B = int(batch_size)
        assert self.B % 16 == 0 and self.B <= 1024, "batch_size must be a multiple of 16 (MFMA rows) and <= 1024"
        self.gamma, self.max_norm, self.reward_scale = float(gamma), float(max_grad_norm), float(reward_scale)
        self.beta1, self.beta2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        # ---- one flat buffer per (parameters, target parameters, gradients, Adam moments); the modules' tensors become views
        sizes = []
        for m in self.layers:
            sizes += [m.weight.numel(), m.bias.numel()]
        self.n_head = sum(sizes)
        self.n_params = self.n_head + sum(p.numel() for p in self.extra)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.flat = torch.empty(self.n_params, **f32)
        self.flat_target = torch.empty(self.n_params if self.extra_t else self.n_head, **f32)
        self.exp_avg = torch.zeros(self.n_params, **f32)
        self.exp_avg_sq = torch.zeros(self.n_params, **f32)
        self.w, self.b, self.wt, self.bt, self.gw, self.gb = [], [], [], [], [], []
        # gradients | activations in ONE allocation (every product overwrites its whole output: nothing has to be zeroed)
        K0 = self.layers[0].in_features
        outs = [m.out_features for m in self.layers]
        act = self.B * sum(outs)
        # (the activations start on a 256-byte boundary: the products read rows four floats at a time only from 16-byte aligned
        # addresses, and the parameter count of the reference network is odd)
        act0 = -(-self.n_params // 64) * 64
        self.work = torch.zeros(act0 + 2 * act + self.B * sum(outs[:-1]) + self.B * outs[-1], **f32)
        # the gradient norm's partial sums: every weight-gradient product leaves its own (bias gradient included)
        shapes = [(m.out_features, m.in_features) for m in self.layers]
        self.sq_off = [0]
        for M_, N_ in shapes:
            self.sq_off.append(self.sq_off[-1] + N.gemm_sumsq_count(M_, N_))
        self.norm_workspace = torch.zeros(max(N.UPD_WORKSPACE, self.sq_off[-1] + 1), **f32)     # (+ 1: apply's extra_norm2)
        z = self.work
        self.grad = z[:self.n_params]
        off = 0
        for m, mt in zip(self.layers, self.layers_t):
            for name in ("weight", "bias"):
                p, pt = getattr(m, name), getattr(mt, name)
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                self.flat_target[off:off + n].copy_(pt.detach().reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                pt.data = self.flat_target[off:off + n].view_as(pt)
                (self.w if name == "weight" else self.b).append(p.data)
                (self.wt if name == "weight" else self.bt).append(pt.data)
                (self.gw if name == "weight" else self.gb).append(self.grad[off:off + n].view_as(p))
                off += n
        self.extra_grad = self.grad[off:]
        for i, p in enumerate(self.extra):
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            if self.extra_t:
                pt = self.extra_t[i]
                self.flat_target[off:off + n].copy_(pt.detach().reshape(-1))
                pt.data = self.flat_target[off:off + n].view_as(pt)
            off += n
        cur = act0
        self.z, self.zt, self.da = [], [], []
        for lst in (self.z, self.zt):                      # pre-activations of the online / the target network
            for o in outs:
                lst.append(z[cur:cur + self.B * o].view(self.B, o)); cur += self.B * o
        for o in outs[:-1]:                                # gradients w.r.t. the hidden activations
            self.da.append(z[cur:cur + self.B * o].view(self.B, o)); cur += self.B * o
        self.dq = z[cur:cur + self.B * outs[-1]].view(self.B, outs[-1]); cur += self.B * outs[-1]
        self.scalars = torch.zeros(N.UPD_COUNT, **f32)     # loss | norm^2 | step | bias corrections | lr   (step persists: not zeroed)
        self.K0, self.outs = K0, outs
        self.dx0 = torch.zeros(self.B, K0, **f32) if self.input_grad else None
        self.scalars[N.UPD_LR] = float(lr)

    # ---- helpers -----------------------------------------------------------------------------------------------
    @staticmethod
    def _product(A, Bm, Cm, M, Nn, K, a_sm, a_sk, b_sk, b_sn, flags=0, bias=None, mask=None, row_sum=None, sumsq=None):
        dp = lambda t: None if t is None else t.data_ptr()
        return N.UavGemm(A=dp(A), B=dp(Bm), C=dp(Cm), bias=dp(bias), a_mask=dp(mask), row_sum=dp(row_sum), M=M, N=Nn, K=K, flags=flags,
                         a_sm=a_sm, a_sk=a_sk, b_sk=b_sk, b_sn=b_sn, ldc=Cm.stride(0), sumsq=dp(sumsq))

    def _launch(self, first, second, stream):
        rc = self.L.uavenv_gemm_f32(C.byref(first), None if second is None else C.byref(second), stream)
        if rc:
            raise RuntimeError(f"uavenv_gemm_f32 failed ({rc})")

    def set_lr(self, lr):
        self.scalars[N.UPD_LR] = float(lr)

    def sync_target(self):
        """SB3's hard target update (tau = 1) of everything this object holds of the target network: one contiguous copy."""
        self.flat_target.copy_(self.flat[:self.flat_target.numel()])

    def collect_extra_grads(self):
        """The extra parameters' gradients (fresh tensors from autograd: set `.grad = None` before the backward pass, so that it
        hands them over instead of adding them element by element into existing ones) -> the flat gradient buffer, one launch."""
        torch.cat([p.grad.reshape(-1) for p in self.extra], out=self.extra_grad)

    @property
    def loss(self):
        return self.scalars[N.UPD_LOSS]

    @property
    def step_count(self):
        return int(self.scalars[N.UPD_STEP].item())

    def _forward_layer(self, l, x, ws, bs, zs):
        """Y[b][n] = bias[n] + sum_k act(X[b][k]) W[n][k] of layer l: both operands contiguous along k"""
        inp = x if l == 0 else zs[l - 1]
        K = self.K0 if l == 0 else self.outs[l - 1]
        W = ws[l]
        return self._product(inp, W, zs[l], self.B, W.shape[0], K, inp.stride(0), 1, 1, W.stride(0),
                             flags=N.GEMM_BIAS | (N.GEMM_A_RELU if l > 0 else 0), bias=bs[l])

    # ---- the update ----------------------------------------------------------------------------------------------
    def backward(self, batch):
        """sample -> loss -> gradients (in `self.grad`).  batch: dict of the ring's sample_stacked (obs, next_obs, action,
        reward, valid).  Everything is enqueued on the current stream; nothing synchronises.  9 launches (the weight-gradient
        products also leave the gradient norm's partial sums): a layer of the online
        and of the target network share one, so do a layer's weight gradient and the gradient w.r.t. its input."""
        B = self.B
        obs, nxt = batch["obs"], batch["next_obs"]
        assert obs.shape == (B, self.K0) and obs.is_contiguous() and nxt.is_contiguous() and obs.dtype == torch.float32
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        nl = len(self.layers)
        for l in range(nl):
            self._launch(self._forward_layer(l, obs, self.w, self.b, self.z), self._forward_layer(l, nxt, self.wt, self.bt, self.zt), stream)
        valid = batch["valid"]
        valid_u8 = valid.view(torch.uint8) if valid.dtype == torch.bool else valid
        rc = self.L.uavenv_td_loss(_p(self.z[-1]), _p(self.zt[-1]), _p(batch["action"]), _p(batch["reward"]), _p(valid_u8), B, self.outs[-1],
                                   self.gamma, self.reward_scale, self.beta1, self.beta2, _p(self.dq), _p(self.scalars), stream)
        if rc:
            raise RuntimeError(f"uavenv_td_loss failed ({rc})")
        # backward through the layers, last to first; dz_l = da_l masked by (z_l > 0) on the fly (no mask at the output layer)
        dz, mask = self.dq, None
        for l in range(nl - 1, -1, -1):
            n = self.outs[l]
            x_in = obs if l == 0 else self.z[l - 1]
            K = self.K0 if l == 0 else self.outs[l - 1]
            mflag = N.GEMM_A_MASK if mask is not None else 0
            # dW[n][k] = sum_b dz[b][n] act(x[b][k]);  db[n] = sum_b dz[b][n]   (A = dz read transposed, B = x; the sum runs over b)
            dw = self._product(dz, x_in, self.gw[l], n, K, B, 1, dz.stride(0), x_in.stride(0), 1,
                               flags=mflag | N.GEMM_ROWSUM | N.GEMM_SUMSQ | (N.GEMM_B_RELU if l > 0 else 0), mask=mask, row_sum=self.gb[l],
                               sumsq=self.norm_workspace[self.sq_off[l]:])
            dx = None
            if l > 0 or self.input_grad:
                # da[b][k] = sum_n dz[b][n] W[n][k]
                dx = self._product(dz, self.w[l], self.da[l - 1] if l > 0 else self.dx0, B, K, n, dz.stride(0), 1, self.w[l].stride(0), 1,
                                   flags=mflag, mask=mask)
            self._launch(dw, dx, stream)
            if l > 0:
                dz, mask = self.da[l - 1], self.z[l - 1]

    def apply(self, grads_changed=False, extra_norm2=None):
        """clip_grad_norm_ + Adam over the flat buffers, with the gradients as they stand in `self.grad`.  The squared norm comes
        from the partial sums `backward` left -- unless `grads_changed` (several ranks: `self.grad` has been all-reduced since):
        then one more launch recomputes it from `self.grad`.  extra_norm2 (a device scalar): the squared gradient norm of
        parameters that are NOT in the flat buffer but are clipped together with it (a features extractor trained by autograd);
        scalars[UPD_NORM2] is the total, from which the caller scales those gradients by the same coefficient."""
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        n_partials = 0 if grads_changed else self.sq_off[-1]
        if self.extra and not grads_changed:                  # the extra parameters' share of the squared norm: one more partial sum
            assert extra_norm2 is None
            extra_norm2 = torch.dot(self.extra_grad, self.extra_grad)
        if extra_norm2 is not None:
            assert not grads_changed
            self.norm_workspace[n_partials:n_partials + 1].copy_(extra_norm2.reshape(1))
            n_partials += 1
        rc = self.L.uavenv_clip_adam(_p(self.flat), _p(self.grad), _p(self.exp_avg), _p(self.exp_avg_sq), self.n_params, _p(self.scalars),
                                     _p(self.norm_workspace), n_partials, self.max_norm, self.beta1,
                                     self.beta2, self.eps, stream)
        if rc:
            raise RuntimeError(f"uavenv_clip_adam failed ({rc})")

    def clip_coefficient(self):
        """min(1, max_norm / (norm + 1e-6)) of the last `apply` (torch.nn.utils.clip_grad_norm_'s factor), as a device scalar."""
        return torch.clamp(self.max_norm / (self.scalars[N.UPD_NORM2].sqrt() + 1e-6), max=1.0)

    def update(self, batch):
        self.backward(batch)
        self.apply()
        return self.loss
