"""Device-resident DQN learner over the batched environment (the caller on both sides of the hot path
in the trainer loop).  It stands where the reference runs

    DQN("MlpPolicy", VecFrameStack(DummyVecEnv([Monitor(DomainRandEnv(..))] * 4), n_stack=4), **HYPERPARAMS).learn(3_000_000)

(agents/dqn/dqn.py:1077-1099 HYPERPARAMS / TRAINING_CONFIG, :1276-1288, :1324) and follows stable-baselines3 2.7's DQN
semantics -- restated from its public documentation, SB3 is not installed in the build image, so this row is "parity
unpinned" by a run of SB3 itself; its arithmetic is pinned by tests/test_gpu_learner.py against a hand-written torch
reference:

  * one rollout = `train_freq` VECTOR steps (each adds num_envs transitions), then `gradient_steps` updates;
  * `learning_starts`, `exploration_fraction`, the learning-rate schedule and `total_timesteps` count TRANSITIONS
    (num_timesteps += num_envs per vector step);
  * the target network is copied every max(target_update_interval // num_envs, 1) vector steps (tau = 1);
  * loss = smooth-L1(Q(s, a), r + gamma * (1 - done) * max_a' Q_target(s', a')); an episode end by truncation is NOT a
    `done` for the target (SB3 handle_timeout_termination; `terminated` is always False in this environment,
    uav_env.py:471), so the target bootstraps from the TERMINAL observation -- which the step kernel has put into the
    replay ring (replay.py);
  * Adam, gradient clipping at max_grad_norm = 10;
  * the learning rate is `schedule(progress_remaining)`; HYPERPARAMS passes `lambda progress: 3e-4 * max(0.1, 1.0 -
    progress * 0.8)` and SB3 calls it with progress_REMAINING (1 at the start, 0 at the end), so the rate the reference
    trains with RISES from 6e-5 to 3e-4 (dqn.py:1081's comment says the opposite; the code is what runs);
  * epsilon-greedy: epsilon falls linearly from 1.0 to `exploration_final_eps` over `exploration_fraction` of the run.
    SB3's predict() flips ONE coin for the whole vector of environments; with thousands of environments that is an
    artefact, so every environment flips its own (shared_exploration_coin=True restores SB3's behaviour).

Launch-bound loops are replayed as HIP graphs (`use_graphs`): one vector step of acting -- Q-network forward,
epsilon-greedy choice, environment step into the ring slot, frame stack, target copy -- is ~15 launches of a few
microseconds of GPU work each behind ~270 us of Python, and a gradient step ~150 launches behind ~2 ms; captured once per
ring slot (the slot's pointers are baked into the launches) and once for the update, a vector step of the reference
configuration costs what its kernels cost.  The eager paths stay (CPU rings, tests of the arithmetic).

Everything stays on the GPU: the step kernel writes observations and (action, reward, done, terminal ticket) straight
into the TransitionRing, the frame stack for acting is the uavenv_frame_stack kernel, sampled batches gather their
frame stacks from the ring (each frame stored once instead of 2 x n_stack times, dqn.py:1085's budget).

Several ranks (torch.distributed initialised; BASELINE config 4 as a training run): every rank steps its own shard of
environments, the ring all-gathers the transition blocks chunk by chunk, and every rank draws batch_size / world samples of
the SHARED ring for an update, so that one update still sees `batch_size` transitions (dqn.py:1086) -- the gradients are
averaged with ONE all-reduce of a flat buffer and every rank applies the same Adam step: all replicas hold identical weights.
Graph replay coexists with the exchange: the collectives are issued from the host between replays (the chunk's all-gather
when its last slot has been stepped; the gradient all-reduce between the update's two graphs: sample -> loss -> backward ->
flatten | clip -> Adam).
"""
import copy
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from .frame_stack import FrameStack
from .replay import TransitionRing

# agents/dqn/dqn.py:1077-1099
REFERENCE_HYPERPARAMS = dict(
    learning_rate=lambda progress_remaining: 3e-4 * max(0.1, 1.0 - progress_remaining * 0.8),
    buffer_size=150_000, batch_size=256, gamma=0.99, learning_starts=25_000, exploration_fraction=0.25,
    exploration_final_eps=0.03, target_update_interval=5_000, train_freq=4, gradient_steps=1,
    net_arch=(512, 512, 256), n_stack=4, total_timesteps=3_000_000, max_grad_norm=10.0, exploration_initial_eps=1.0,
)


class _AttentionCore(torch.autograd.Function):
    """scores = qk . kv^T, masked softmax over the tokens, mix = attn . kv for one query per sample -- forward and backward in one
    launch each (csrc/uavenv_learner.hip: attn_core_*_kernel) instead of six batched 4 x 64 x 50 products + masked_fill + softmax
    and their backward (~15 launches, the GEMM library's 128 x 256 macro tiles at 20 us apiece)."""

    @staticmethod
    def forward(ctx, qk, kv, mask):
        import ctypes as C
        B, H, E = qk.shape
        T = kv.shape[1]
        qk, kv = qk.contiguous(), kv.contiguous()
        m8 = mask.contiguous().view(torch.uint8)
        mix = torch.empty(B, H, E, dtype=torch.float32, device=qk.device)
        attn = torch.empty(B, H, T, dtype=torch.float32, device=qk.device)
        N.check(N.lib().uavenv_attn_core_forward(C.c_void_p(qk.data_ptr()), C.c_void_p(kv.data_ptr()), C.c_void_p(m8.data_ptr()), B, H, T,
                                                 C.c_void_p(mix.data_ptr()), C.c_void_p(attn.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(qk.device).cuda_stream)))
        ctx.save_for_backward(qk, kv, attn)
        return mix

    @staticmethod
    def backward(ctx, dmix):
        import ctypes as C
        qk, kv, attn = ctx.saved_tensors
        B, H, E = qk.shape
        T = kv.shape[1]
        dmix = dmix.contiguous()
        dqk, dkv = torch.empty_like(qk), torch.empty_like(kv)
        N.check(N.lib().uavenv_attn_core_backward(C.c_void_p(qk.data_ptr()), C.c_void_p(kv.data_ptr()), C.c_void_p(attn.data_ptr()),
                                                  C.c_void_p(dmix.data_ptr()), B, H, T, C.c_void_p(dqk.data_ptr()), C.c_void_p(dkv.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream(qk.device).cuda_stream)))
        return dqk, dkv, None


class AttentionFeatures(nn.Module):
    """Layer for layer the architecture of the reference's UAVAttentionExtractor (dqn.py:548-650): an MLP over the UAV
    header of all stacked frames, one-query cross-attention over the 50 sensor slots of the newest frame with ghost
    and out-of-range slots masked, LayerNorm, fusion to 128 features.  (The module names are this package's; the fused
    inference kernel accepts both naming schemes, attention.py.)"""

    def __init__(self, n_stack, frame=153, slots=50, embed=64, heads=4, features=128):
        super().__init__()
        self.n_stack, self.frame, self.slots = n_stack, frame, slots
        self.uav = nn.Sequential(nn.Linear(3 * n_stack, embed), nn.LayerNorm(embed), nn.ReLU())
        self.sensor = nn.Linear(3, embed)
        self.attn = nn.MultiheadAttention(embed, heads, batch_first=True)
        self.norm = nn.LayerNorm(embed)
        self.fuse = nn.Sequential(nn.Linear(2 * embed, features), nn.ReLU())
        self.features_dim = features
        self.fused_core = True          # the scores / softmax / mix core of `forward` in the library's kernels (GPU); False: PyTorch ops

    def _inputs(self, obs):
        B = obs.shape[0]
        fr = obs.view(B, self.n_stack, self.frame)
        sens = fr[:, -1, 3:].view(B, self.slots, 3)
        mask = (sens.abs().sum(-1) < 1e-6) | (sens[:, :, 2] < 1e-6)
        mask = mask & ~mask.all(1, keepdim=True)
        return fr[:, :, :3].reshape(B, -1), sens, mask

    def forward_module(self, obs):
        """The architecture as the reference writes it (dqn.py:617-650): nn.Linear over the tokens, nn.MultiheadAttention."""
        uav, sens, mask = self._inputs(obs)
        q = self.uav(uav)
        kv = F.relu(self.sensor(sens))
        ctx, _ = self.attn(q.unsqueeze(1), kv, kv, key_padding_mask=mask)
        return self.fuse(torch.cat([q, self.norm(ctx.squeeze(1))], -1))

    def forward(self, obs):
        """The same function of the same parameters, arranged for ONE query per sample (what the training step differentiates):
        the key projection is folded into the query (q_h . (Wk_h kv_t + bk_h) = (Wk_h^T q_h) . kv_t + const -- the constant drops
        out of the softmax) and the value projection applied to the attention-weighted mean of the tokens instead of to every token
        (the weights sum to one), so no [batch x 50 tokens] GEMM is left in the forward or the backward pass; the 3 -> 64 sensor
        projection is a batched product with the weight expanded over the batch, so that its weight gradient is a batched product
        plus a sum instead of one [64 x 12 800] . [12 800 x 3] product.  With nn.MultiheadAttention those token-level products were
        150 us of every update (two weight-gradient GEMMs of 91 and 59 us: a 12 800-long reduction into a 64 x 3 / 128 x 64 result)."""
        uav, sens, mask = self._inputs(obs)
        B, H, E = obs.shape[0], self.attn.num_heads, self.attn.embed_dim
        d = E // H
        q = self.uav(uav)
        # (the bias rides along as a fourth input that is always 1: as `baddbmm(bias, ...)` its gradient is a [B, T, E] -> [E] reduction
        #  that comes back STALE from HIP-graph replays on this PyTorch / ROCm build -- every other gradient of the module replays
        #  correctly, tools/graph_grad_check.py -- while the [B, 4, E] -> [4, E] sum of the expanded weight's gradient does not)
        w1 = torch.cat([self.sensor.weight.t(), self.sensor.bias.unsqueeze(0)], 0)                              # [4, E]
        kv = F.relu(torch.bmm(F.pad(sens, (0, 1), value=1.0), w1.expand(B, 4, E)))                              # [B, T, E]
        wq, wk, wv = self.attn.in_proj_weight.view(3, H, d, E)
        bq, _, bv = self.attn.in_proj_bias.view(3, H, d)
        qh = (F.linear(q, wq.reshape(E, E), bq.reshape(E)) * d ** -0.5).view(B, H, d)
        qk = torch.einsum("bhd,hde->bhe", qh, wk)                                                              # [B, H, E]
        if obs.is_cuda and E == 64 and self.slots <= 64 and H in (1, 2, 4, 8) and self.fused_core:
            mix = _AttentionCore.apply(qk, kv, mask)                                                           # [B, H, E], one launch each way
        else:
            scores = torch.bmm(qk, kv.transpose(1, 2)).masked_fill(mask.unsqueeze(1), float("-inf"))          # [B, H, T]
            mix = torch.bmm(torch.softmax(scores, -1), kv)                                                     # [B, H, E]
        ctx = (torch.einsum("bhe,hde->bhd", mix, wv) + bv).reshape(B, E)
        ctx = F.linear(ctx, self.attn.out_proj.weight, self.attn.out_proj.bias)
        return self.fuse(torch.cat([q, self.norm(ctx)], -1))


class QNetwork(nn.Module):
    """SB3's DQN "MlpPolicy" q-net: features extractor (Flatten, or the attention extractor) + MLP `net_arch` + 5 outputs."""

    def __init__(self, obs_dim, n_stack, net_arch=(512, 512, 256), extractor="mlp", n_actions=5):
        super().__init__()
        if extractor == "attention":
            self.features = AttentionFeatures(n_stack, frame=obs_dim)
            d = self.features.features_dim
        else:
            self.features = nn.Flatten()
            d = obs_dim * n_stack
        layers = []
        for h in net_arch:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        self.head = nn.Sequential(*layers, nn.Linear(d, n_actions))

    def forward(self, x):
        return self.head(self.features(x))

    @torch.no_grad()
    def head_inference(self, h, upto_last=False):
        """`self.head(h)` for acting (no gradients): every Linear + ReLU pair as ONE library GEMM with the bias + ReLU epilogue
        (`torch._addmm_activation`: hipBLASLt's fused epilogue, bit-identical to Linear followed by ReLU) -- eager PyTorch runs the
        ReLU as a launch of its own over the [envs x width] activations: 3 x 6.5 us of a 144 us vector step at 4096 environments."""
        mods = list(self.head)[:-1] if upto_last else list(self.head)      # upto_last: stop in front of the output layer
        fused = h.is_cuda and hasattr(torch, "_addmm_activation")
        i = 0
        while i < len(mods):
            m = mods[i]
            if fused and isinstance(m, nn.Linear) and m.bias is not None and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                h = torch._addmm_activation(m.bias, h, m.weight.t())
                i += 2
            else:
                h = m(h)
                i += 1
        return h

    @torch.no_grad()
    def q_inference(self, x, features=None):
        """Q-values for acting: `features` (optional) replaces the extractor's module forward (the fused attention kernel)."""
        return self.head_inference(self.features(x) if features is None else features(x))


def tunable_cache_path(device):
    """Where the TunableOp results for `device` are kept: $UAVENV_CACHE_DIR (default ~/.cache/uavenv_amd) /
    tunableop_<device model>_<torch version>.csv -- GEMM choices depend on the chip and the library build, not on the process."""
    root = os.environ.get("UAVENV_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "uavenv_amd")
    name = torch.cuda.get_device_name(device) if torch.cuda.is_available() else "cpu"
    tag = "".join(c if c.isalnum() else "_" for c in f"{name}_{torch.__version__}")
    return os.path.join(root, f"tunableop_{tag}.csv")


def linear_epsilon(progress_remaining, initial, final, fraction):
    """SB3 get_linear_fn(initial, final, fraction)(progress_remaining)."""
    done = 1.0 - progress_remaining
    if done > fraction:
        return final
    return initial + done * (final - initial) / fraction


def td_loss(q_net, q_target, batch, gamma, reward_scale=1.0):
    """SB3 DQN.train's loss on one sampled batch: smooth-L1 between Q(s, a) and r + gamma * max_a' Q_target(s', a'),
    averaged over the transitions whose next observation is available (`valid`).  No (1 - done) factor: every episode
    end of this environment is a truncation, which SB3 does not treat as terminal for the target."""
    with torch.no_grad():
        target = reward_scale * batch["reward"] + gamma * q_target(batch["next_obs"]).max(dim=1).values
    current = q_net(batch["obs"]).gather(1, batch["action"].unsqueeze(1)).squeeze(1)
    w = batch["valid"].to(current.dtype)
    return (F.smooth_l1_loss(current, target, reduction="none") * w).sum() / w.sum().clamp(min=1.0)


class DQNLearner:
    """`DQNLearner(env, **REFERENCE_HYPERPARAMS).learn()` = the reference's `DQN(...).learn(total_timesteps)` on device."""

    def __init__(self, env, learning_rate=REFERENCE_HYPERPARAMS["learning_rate"], buffer_size=150_000, batch_size=256, gamma=0.99,
                 learning_starts=25_000, exploration_fraction=0.25, exploration_final_eps=0.03, exploration_initial_eps=1.0,
                 target_update_interval=5_000, train_freq=4, gradient_steps=1, net_arch=(512, 512, 256), n_stack=4,
                 total_timesteps=3_000_000, max_grad_norm=10.0, extractor="mlp", shared_exploration_coin=False, seed=0,
                 chunk_len=None, reward_scale=1.0, use_graphs=None, tune_gemms=None, frame_stack_cls=FrameStack,
                 updates_per_transition=None, fused_update=None, replicate_replay=True):
        """reward_scale (not an SB3 / reference option; default 1.0 = theirs): the environment's rewards reach 1e4-1e5 per
        step (+5000 per new sensor, 100 x bytes x urgency), which a smooth-L1 loss follows at one unit of gradient per
        sample -- the reference spends 750 k gradient steps on it.  Short runs (the tests) scale the reward in the loss."""
        self.env, self.dev, self.E, self.D = env, env.device, env.num_envs, env.obs_dim
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.n_envs_total = self.E * self.world                   # SB3's n_envs: transitions per vector step
        self.lr_schedule = learning_rate if callable(learning_rate) else (lambda _p, v=float(learning_rate): v)
        self.batch_size, self.gamma, self.learning_starts = int(batch_size), float(gamma), int(learning_starts)
        # every rank draws its share of the batch from the shared ring: one update = batch_size transitions in all (dqn.py:1086)
        assert self.batch_size % self.world == 0, "batch_size must be divisible by the number of ranks"
        self.local_batch = self.batch_size // self.world
        self.eps0, self.eps1, self.eps_fraction = float(exploration_initial_eps), float(exploration_final_eps), float(exploration_fraction)
        self.train_freq, self.gradient_steps = int(train_freq), int(gradient_steps)
        # SB3: gradient_steps = -1 means "as many updates as transitions collected in the rollout".  The reference (4 environments,
        # train_freq 4, gradient_steps 1) makes ONE update per 16 transitions; with thousands of environments gradient_steps = 1
        # is one update per train_freq * n_envs transitions -- 1 / 16 384 at 4096 environments: a loop that steps fast and learns
        # nothing.  `updates_per_transition` states the ratio directly (1 / 16 = the reference's) and overrides gradient_steps.
        if updates_per_transition is not None:
            self.gradient_steps = max(1, round(float(updates_per_transition) * self.train_freq * self.n_envs_total))
        elif self.gradient_steps < 0:
            self.gradient_steps = self.train_freq * self.n_envs_total
        self.updates_per_transition = self.gradient_steps / float(self.train_freq * self.n_envs_total)
        self.total_timesteps, self.max_grad_norm, self.k = int(total_timesteps), float(max_grad_norm), int(n_stack)
        self.target_every = max(int(target_update_interval) // self.n_envs_total, 1)     # vector steps (SB3 DQN._on_step)
        self.shared_coin = bool(shared_exploration_coin)
        self.reward_scale = float(reward_scale)
        torch.manual_seed(seed)                                    # same seed on every rank: identical initial weights
        self.q = QNetwork(self.D, self.k, net_arch, extractor).to(self.dev)
        self.q_target = copy.deepcopy(self.q).requires_grad_(False)
        # (capturable + a tensor learning rate: the update can be replayed as a graph; eager steps use the same optimiser)
        on_gpu = self.dev.type == "cuda"
        self.opt = torch.optim.Adam(self.q.parameters(), lr=torch.tensor(self.lr_schedule(1.0), device=self.dev) if on_gpu
                                    else self.lr_schedule(1.0), capturable=on_gpu, fused=on_gpu or None)
        # fused_update (default: on for the MLP policy on a GPU): the gradient step runs in the library's own kernels
        # (mlp_update.py / csrc/uavenv_learner.hip: small-batch MFMA GEMMs, TD loss, clip + Adam over one flat buffer, ~20
        # launches) instead of torch autograd + torch.optim.Adam (~60 launches, a dozen of them 256-row library GEMMs that use
        # 16 of 256 CUs); the modules' parameters become views of its flat buffers.  False keeps the PyTorch update.
        # With the attention extractor (one rank) the same kernels run the layers AFTER the extractor -- both networks' heads, the
        # loss, the head's backward down to the gradient w.r.t. the features, clip + Adam of the head -- and autograd continues from
        # that gradient through the extractor, into gradient views of the library's flat buffer: one clip + Adam launch for all ("hybrid").
        self.fused_update = (on_gpu and (extractor == "mlp" or (self.world == 1 and self.D == 153))) if fused_update is None else bool(fused_update)
        assert not (self.fused_update and extractor != "mlp" and (self.world > 1 or self.D != 153 or not on_gpu)), \
            "the hybrid update (attention extractor) is single-rank, on a GPU, for 153-float frames"
        self._mlp, self._hybrid = None, False
        if self.fused_update:
            from .mlp_update import FusedMLPUpdate
            self._hybrid = extractor != "mlp"
            self._mlp = FusedMLPUpdate(self.q, self.q_target, self.local_batch, self.gamma, self.max_grad_norm, self.reward_scale,
                                       lr=self.lr_schedule(1.0), input_grad=self._hybrid,
                                       extra_params=list(self.q.features.parameters()) if self._hybrid else (),
                                       extra_target_params=list(self.q_target.features.parameters()) if self._hybrid else ())
            self.opt = None                                   # (every parameter now belongs to the library's optimiser)
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs)
        self._act_graphs, self._train_graph, self._train_graph_b, self._fused = None, None, None, None
        # one flat buffer for the gradient all-reduce (world > 1)
        self._flat_grad = None
        if self.world > 1:
            self._flat_grad = self._mlp.grad if self._mlp is not None else torch.zeros(sum(p.numel() for p in self.q.parameters()), device=self.dev)
        # tune_gemms: PyTorch's TunableOp picks the GEMM kernel per shape by timing the candidates the first time a shape is seen
        # (the eager steps before the captures).  The update is a dozen float32 GEMMs of batch 256 whose default kernels leave
        # most of the 256 CUs idle (14-29 us each): 599 -> 348 us per update (1 491 -> 864 us with the attention extractor).
        # The tuning itself takes 3 s (13 s with the attention extractor), so its results are PERSISTED: one file per device
        # model and PyTorch build under the cache directory (tunable_cache_path), read back by the next process -- which then
        # tunes only shapes it has not seen -- and written again once both graphs exist.  None (default) = use the cache when
        # there is one, tune from scratch only for runs long enough to pay for it (>= 20 M timesteps); True / False force it.
        # It is a process-wide PyTorch switch; tuning is switched off again once both graphs exist (the chosen kernels stay).
        self._tune_path = tunable_cache_path(self.dev) if on_gpu else None
        if tune_gemms is None:
            tune_gemms = on_gpu and (os.path.exists(self._tune_path) or self.total_timesteps >= 20_000_000)
        self.tune_gemms = bool(tune_gemms) and on_gpu
        if self.tune_gemms:
            import torch.cuda.tunable as tunable
            tunable.enable(True)
            tunable.tuning_enable(True)
            try:
                os.makedirs(os.path.dirname(self._tune_path), exist_ok=True)
                tunable.set_filename(self._tune_path)          # read at the first GEMM when it exists (validated against the
                tunable.write_file_on_exit(False)               # library versions it was made with); written by _finish_tuning
            except Exception:
                pass
        self._sample_seed = seed * 7919 + 13 + self.rank          # every rank draws its own share of a batch
        self.gen = torch.Generator(device=self.dev).manual_seed(self._sample_seed)
        # replay: buffer_size transitions = buffer_size // n_envs vector slots (SB3 ReplayBuffer), in chunks (one terminal
        # section and, across ranks, one collective per chunk); one chunk is always being recycled, hence the extra one
        slots = max(int(buffer_size) // self.n_envs_total, self.k + 2)
        L = int(chunk_len) if chunk_len else max(1, min(64, slots // 4))
        capacity = (math.ceil(slots / L) + 1) * L
        # replicate_replay (several ranks): True = the north star's shared buffer, every rank holds every rank's transitions (one
        # all-gather per chunk: 2.57 MB per rank and step at 4096 x 50, the xGMI links bound the step rate);
        # False = every rank keeps ITS OWN transitions only and draws its batch_size / world samples from them -- with the
        # gradients averaged, an update then sees a STRATIFIED uniform sample of the union of the buffers (same expectation as
        # uniform sampling from a shared buffer, lower variance) and no transition ever crosses a link: the only collective left is
        # the gradient all-reduce.  The buffer holds buffer_size transitions in all either way.
        self.replicate_replay = bool(replicate_replay) or self.world == 1
        if self.replicate_replay:
            self.ring = TransitionRing(capacity, self.E, self.D, self.dev, world_size=self.world, rank=self.rank, chunk_len=L)
        else:
            self.ring = TransitionRing(capacity, self.E, self.D, self.dev, world_size=1, rank=0, chunk_len=L)
        self.ring.attach(env)
        self.fs = frame_stack_cls(self.E, self.D, self.k, self.dev)
        self.num_timesteps, self.n_calls, self.n_updates = 0, 0, 0
        self._warm_act, self._warm_upd = 0, 0                       # eager vector steps / updates made by THIS object (graph capture waits for 3)
        self.last_loss = None
        self._stacked = None

    # ---- schedules ------------------------------------------------------------------------------------
    def progress_remaining(self):
        return 1.0 - min(1.0, self.num_timesteps / float(self.total_timesteps))

    def exploration_rate(self):
        return linear_epsilon(self.progress_remaining(), self.eps0, self.eps1, self.eps_fraction)

    # ---- acting ---------------------------------------------------------------------------------------
    def _start(self):
        obs = self.env.reset()
        self.ring.local_obs_slot().copy_(obs)
        z = torch.zeros(self.E, device=self.dev)
        self.ring.commit(z, z, z)                                  # slot 0: the reset observation (no incoming transition)
        self._stacked = self.fs.reset(obs)

    def _select_actions(self, qvalues, eps_dev):
        """epsilon-greedy over a vector of Q-values in ONE launch (uavenv_epsilon_greedy): argmax, coin, random action; the draw
        counter lives on the device and is advanced by the kernel, so eager calls and graph replays draw the same sequence."""
        import ctypes as C
        if self.__dict__.get("_act_out") is None:
            self._act_out = torch.zeros(self.E, dtype=torch.int32, device=self.dev)
            if self.__dict__.get("_act_counter") is None:      # (a checkpoint may have put it there already)
                self._act_counter = torch.zeros(1, dtype=torch.float32, device=self.dev)
            self._alib = N.lib()
        q = qvalues.contiguous()
        rc = self._alib.uavenv_epsilon_greedy(C.c_void_p(q.data_ptr()), self.E, q.shape[1], C.c_void_p(eps_dev.data_ptr()),
                                              C.c_void_p(self._act_counter.data_ptr()), (self._sample_seed * 2654435761 + 97) & 0xFFFFFFFFFFFFFFFF,
                                              1 if self.shared_coin else 0, C.c_void_p(self._act_out.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc:
            raise RuntimeError(f"uavenv_epsilon_greedy failed ({rc})")
        return self._act_out

    def _act_device(self, x, eps_dev, features=None):
        """Actions for a vector of stacked observations on the GPU: the network up to its last hidden layer through the library
        GEMMs, then the output layer + epsilon-greedy selection in ONE launch (uavenv_q_head_select: the [envs x 256] . [256 x 5]
        product was 5 us as a library GEMM, the selection 6 us as a launch of its own)."""
        import ctypes as C
        f = self.q.features(x) if features is None else features(x)
        last = self.q.head[-1]
        if not (isinstance(last, nn.Linear) and last.bias is not None and last.out_features <= 8 and last.in_features % 4 == 0
                and last.in_features * last.out_features <= 16384 and last.weight.data_ptr() % 16 == 0):
            return self._select_actions(self.q.head_inference(f), eps_dev)
        h = self.q.head_inference(f, upto_last=True).contiguous()
        if self.__dict__.get("_act_out") is None:
            self._act_out = torch.zeros(self.E, dtype=torch.int32, device=self.dev)
            if self.__dict__.get("_act_counter") is None:      # (a checkpoint may have put it there already)
                self._act_counter = torch.zeros(1, dtype=torch.float32, device=self.dev)
            self._alib = N.lib()
        if self.__dict__.get("_act_ticket") is None:
            self._act_ticket = torch.zeros(1, dtype=torch.int32, device=self.dev)
        rc = self._alib.uavenv_q_head_select(C.c_void_p(h.data_ptr()), C.c_void_p(last.weight.data_ptr()), C.c_void_p(last.bias.data_ptr()),
                                             self.E, last.in_features, last.out_features, C.c_void_p(eps_dev.data_ptr()),
                                             C.c_void_p(self._act_counter.data_ptr()), C.c_void_p(self._act_ticket.data_ptr()),
                                             (self._sample_seed * 2654435761 + 97) & 0xFFFFFFFFFFFFFFFF, 1 if self.shared_coin else 0,
                                             C.c_void_p(self._act_out.data_ptr()), None,
                                             C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc:
            raise RuntimeError(f"uavenv_q_head_select failed ({rc})")
        return self._act_out

    @torch.no_grad()
    def act(self, stacked, epsilon):
        if self.dev.type == "cuda":
            if self.__dict__.get("_eps_eager") is None:
                self._eps_eager = torch.zeros((), device=self.dev)
            self._eps_eager.fill_(float(epsilon))
            return self._act_device(stacked, self._eps_eager)
        greedy = self.q.q_inference(stacked).argmax(1).to(torch.int32)
        if epsilon <= 0.0:
            return greedy
        rnd = torch.randint(0, 5, (self.E,), device=self.dev, dtype=torch.int32, generator=self.gen)
        if self.shared_coin:
            coin = torch.rand(1, device=self.dev, generator=self.gen).expand(self.E)
        else:
            coin = torch.rand(self.E, device=self.dev, generator=self.gen)
        return torch.where(coin < epsilon, rnd, greedy)

    # ---- graph replay of the two launch-bound loops ------------------------------------------------------
    _GRAPH_SLOT_LIMIT = 1024            # one graph per ring slot: rings longer than this stay eager

    def _set_lr(self, lr):
        if self._mlp is not None:
            self._mlp.set_lr(lr)
            return
        for g in self.opt.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    @torch.no_grad()
    def _sync_target(self):
        """SB3's hard target update (tau = 1), in place.  With the library's update the layers' parameters are two flat buffers: ONE
        contiguous copy (4 us) where a multi-tensor copy over their eight tensors is 11 us -- at 4096 environments
        target_update_interval 5000 means after EVERY vector step."""
        if self.__dict__.get("_sync_lists") is None:
            if self._mlp is not None:                         # (its flat buffers hold every parameter; buffers, if any, go one by one)
                src, dst = [], []
            else:
                src, dst = list(self.q.parameters()), list(self.q_target.parameters())
            self._sync_lists = ([t for t in dst] + list(self.q_target.buffers()), [t.detach() for t in src] + list(self.q.buffers()))
        if self._mlp is not None:
            self._mlp.sync_target()
        if self._sync_lists[0]:
            torch._foreach_copy_(self._sync_lists[0], self._sync_lists[1])

    def _after_vector_step(self):
        self.num_timesteps += self.n_envs_total
        self.n_calls += 1

    def _capture_act_graphs(self):
        """One graph per ring slot: act on the stacked observation, step the environments into that slot, push the frame
        stack, copy the target network when it is due every step.  All graphs share one memory pool (they never overlap)."""
        ring, E = self.ring, self.E
        self._g_eps = torch.zeros((), device=self.dev)
        # the attention extractor acts through the fused inference kernel (csrc/uavenv_attention.hip: one launch of ~50 us for
        # 4096 stacked observations where the eager module spends ~860 us of GPU time); its weight block is re-packed after
        # every update (train graph / eager train)
        self._fused = None
        if isinstance(self.q.features, AttentionFeatures) and self.D == 153:
            from .attention import FusedAttentionFeatures
            self._fused = FusedAttentionFeatures(self.q.features, self.k, self.dev)

        pool = torch.cuda.graph_pool_handle()
        graphs = [None] * ring.capacity
        torch.cuda.synchronize(self.dev)
        # (raw capture_begin / capture_end on one side stream: the torch.cuda.graph context manager synchronises, collects
        # garbage and empties the allocator cache around EVERY capture -- 20 ms each, a second for the 45 slots of the
        # reference configuration)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            for slot in range(ring.capacity):
                g = torch.cuda.CUDAGraph()
                g.register_generator_state(self.gen)
                g.capture_begin(pool=pool)
                try:
                    ring._point_env(slot)
                    actions = self._act_device(self.fs.stacked, self._g_eps, self._fused)
                    o, _, d = self.env.step(actions, obs_out=ring.local_obs_slot(slot))
                    self.fs.step(o, d, None)
                    if self.target_every == 1:
                        self._sync_target()
                finally:
                    g.capture_end()
                graphs[slot] = g
                del actions, o, d
        torch.cuda.current_stream(self.dev).wait_stream(side)
        ring._point_env()                # capturing executed nothing
        self._act_graphs = graphs
        self._act_epoch = self.env.launch_epoch

    def _flatten_grads(self):
        off = 0
        for p in self.q.parameters():
            n = p.numel()
            self._flat_grad[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def _unflatten_grads(self):
        off = 0
        for p in self.q.parameters():
            n = p.numel()
            p.grad.copy_(self._flat_grad[off:off + n].view_as(p.grad))
            off += n

    def _allreduce_grads(self):
        """ONE collective for all gradients; afterwards every rank holds the mean over ranks = the gradient of the mean loss
        over the world's batch_size transitions."""
        dist.all_reduce(self._flat_grad)
        self._flat_grad.div_(self.world)

    def _sample(self, window=None):
        """A batch of this rank's share of the update.  With the library's own update the draw happens inside the gather kernel,
        keyed by (seed, optimiser step): one launch, the same draw whether the step runs eagerly or as a graph replay; otherwise
        the ring draws with the learner's torch generator."""
        if self._mlp is not None:
            if window is None:
                self.ring.drain()
                n, oldest = self.ring.window_state()
                if self.__dict__.get("_win2") is None:
                    self._win2 = torch.zeros(2, dtype=torch.int64, device=self.dev)
                self._win2[0] = n; self._win2[1] = oldest
                window = self._win2
            self._keyed_out = self.ring.sample_stacked_keyed(self.local_batch, self.k, window, self._mlp.scalars[N.UPD_STEP:N.UPD_STEP + 1],
                                                             self._sample_seed, out=self.__dict__.get("_keyed_out"))
            return self._keyed_out
        return self.ring.sample_stacked(self.local_batch, self.k, generator=self.gen, window=window)

    def _target_features(self, x):
        """The target network's extractor on a batch (no gradients): the fused inference kernel, from a weight block re-packed
        here from the target network's parameters as they stand (2 launches; the hard target updates happen in three places, this
        is the one place the block is read)."""
        if self.__dict__.get("_fused_target") is None:
            from .attention import FusedAttentionFeatures
            self._fused_target = FusedAttentionFeatures(self.q_target.features, self.k, self.dev)
        self._fused_target.refresh(self.q_target.features)
        return self._fused_target(x)

    def _target_q(self, x):
        """Q-values of the target network (no gradients).  The attention extractor's forward runs in the fused inference kernel,
        from a weight block re-packed here from the target network's parameters as they stand (2 launches; the hard target
        updates happen in three places, this is the one place the block is read) -- ~15 eager launches less per update."""
        if self.dev.type != "cuda" or not isinstance(self.q_target.features, AttentionFeatures) or self.D != 153:
            return self.q_target(x)
        return self.q_target.head(self._target_features(x))

    def _backward(self, batch):
        """Loss and gradients of one batch (several ranks: the gradients end up in the flat buffer the all-reduce works on).
        Returns the detached loss."""
        if self._hybrid:
            feat = self.q.features(batch["obs"])
            with torch.no_grad():
                feat_next = self._target_features(batch["next_obs"])
            self._mlp.backward(dict(batch, obs=feat.detach(), next_obs=feat_next))
            for p in self._mlp.extra:
                p.grad = None                                  # (autograd then hands its gradient tensors over: no accumulate launches)
            feat.backward(self._mlp.dx0)
            self._mlp.collect_extra_grads()                    # -> the library's flat gradient buffer, one launch
            return self._mlp.loss
        if self._mlp is not None:
            self._mlp.backward(batch)
            return self._mlp.loss
        loss = td_loss(self.q, self._target_q, batch, self.gamma, self.reward_scale)
        # (one rank inside a capture: fresh gradient tensors from the graph's pool; several ranks: the tensors are part of both graphs)
        self.opt.zero_grad(set_to_none=self.world == 1)
        loss.backward()
        if self.world > 1:
            self._flatten_grads()
        # (detached: a loss that keeps its autograd graph keeps the parameters' AccumulateGrad nodes alive, and those remember
        # the stream they were created on -- a later backward inside a graph capture would run them on that other stream,
        # which ends the capture with a segmentation fault in the HIP runtime)
        return loss.detach()

    def _apply(self):
        """clip_grad_norm_ + Adam on the (averaged) gradients."""
        if self._hybrid:
            # clip_grad_norm_ + Adam over ALL parameters in one launch: the extractor's parameters, gradients and moments live in the
            # same flat buffers as the head's; its share of the squared norm joins the head's partial sums
            self._mlp.apply()
            if self._fused is not None:
                self._fused.refresh(self.q.features)
            return
        if self._mlp is not None:
            self._mlp.apply(grads_changed=self.world > 1)       # (several ranks: the partial sums of the norm predate the all-reduce)
            return
        if self.world > 1:
            self._unflatten_grads()
        nn.utils.clip_grad_norm_(self.q.parameters(), self.max_grad_norm)
        self.opt.step()
        if self._fused is not None:
            self._fused.refresh(self.q.features)

    def _train_graph_safe(self):
        """May the update be replayed as a HIP graph?  The library's own update (MLP policy): always.  Updates with PyTorch autograd in
        them (the attention extractor; `fused_update=False`): only up to a batch of 256 per rank -- on this PyTorch 2.10 / ROCm 7
        build, reductions that autograd spreads over several workgroups (a bias gradient over >= 512 rows, the 12 800 token rows of
        nn.Linear over [batch x 50]) come back STALE from graph replays (tools/graph_grad_check.py, BATCH=512); at 256 every
        gradient replays exactly (tests/test_gpu_attention.py).  Above that the update runs eagerly; acting is replayed regardless."""
        return (self._mlp is not None and not self._hybrid) or self.local_batch <= 256

    def _capture_train_graph(self):
        """One gradient step -- sample, TD loss, backward, clip, Adam -- as a graph.  Captured after eager updates have run
        (optimizer state and library workspaces exist); the ring's sampling window and the learning rate are device scalars
        refreshed before each replay.  With several ranks the step is TWO graphs around the host-issued gradient all-reduce:
        [sample -> loss -> backward (-> flatten)] and [(unflatten ->) clip -> Adam]."""
        self._g_win2 = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self._g_win = (self._g_win2[0], self._g_win2[1])
        self._g_loss = torch.zeros((), device=self.dev)
        self._g_index = torch.zeros(4, self.local_batch, dtype=torch.int64, device=self.dev)    # the last replay's draw (tests)
        n, oldest = self.ring.window_state()
        self._g_win[0].fill_(n); self._g_win[1].fill_(oldest)
        self.ring.drain()                                 # (inside the capture sample_stacked must not wait on collectives)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        g.register_generator_state(self.gen)
        pool = torch.cuda.graph_pool_handle()
        with torch.cuda.graph(g, pool=pool):
            batch = self._sample(window=self._g_win2 if self._mlp is not None else self._g_win)
            loss = self._backward(batch)
            if self._mlp is not None:                     # (the library's update leaves both in buffers of its own: no copies)
                self._g_loss, self._g_index = loss, batch["index_block"]
            else:
                self._g_loss.copy_(loss)
                self._g_index.copy_(torch.stack(batch["index"]))
            if self.world == 1:
                self._apply()
        self._train_graph = g
        if self.world > 1:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=pool):
                self._apply()
            self._train_graph_b = gb
        if self.tune_gemms and self._act_graphs is not None:
            self._finish_tuning()                         # every shape of the two loops has been seen

    def _finish_tuning(self):
        """Stop timing candidates and persist what was chosen (atomically: several ranks / processes may share the cache)."""
        import torch.cuda.tunable as tunable
        tunable.tuning_enable(False)
        try:
            tmp = f"{self._tune_path}.{os.getpid()}.tmp"
            if tunable.write_file(tmp):
                os.replace(tmp, self._tune_path)
        except Exception:
            pass

    def _graphs_usable(self):
        return self.use_graphs and self.dev.type == "cuda" and self.ring.capacity <= self._GRAPH_SLOT_LIMIT

    def collect(self, vector_steps):
        """SB3 collect_rollouts: `vector_steps` steps of every environment into the replay ring."""
        if self._stacked is None:
            self._start()
        for _ in range(vector_steps):
            if self._act_graphs is not None and self._act_epoch != self.env.launch_epoch:
                self._act_graphs = None          # env.seed() etc. since the capture: the launches carry stale arguments
            if self._act_graphs is None and self._warm_act >= 3 and self._graphs_usable():  # (libraries are warm after 3 eager steps OF THIS PROCESS: a loaded checkpoint brings its counters)
                self._capture_act_graphs()
            if self._act_graphs is not None:
                self._g_eps.fill_(self.exploration_rate())
                self._act_graphs[self.ring.head].replay()
                self.ring._advance(1)
                self.ring._point_env()
                self._after_vector_step()
                if self.target_every != 1 and self.n_calls % self.target_every == 0:
                    self._sync_target()
                continue
            actions = self.act(self._stacked, self.exploration_rate())
            o, _, d = self.env.step(actions, obs_out=self.ring.local_obs_slot())
            self.ring.commit()
            self._stacked = self.fs.step(o, d, None)               # terminal rows live in the ring, not in env.terminal_obs
            self.num_timesteps += self.n_envs_total
            self.n_calls += 1
            self._warm_act += 1
            if self.n_calls % self.target_every == 0:
                self._sync_target()

    # ---- learning -------------------------------------------------------------------------------------
    def train(self, gradient_steps=None):
        self._set_lr(self.lr_schedule(self.progress_remaining()))
        steps = self.gradient_steps if gradient_steps is None else gradient_steps
        if self._graphs_usable() and self._train_graph is None and self._warm_upd >= 3 and self._train_graph_safe():
            self._capture_train_graph()
        if self._train_graph is not None:
            self.ring.drain()                              # the gathers of finished chunks (host side; no-op alone)
            n, oldest = self.ring.window_state()
            self._g_win[0].fill_(n); self._g_win[1].fill_(oldest)
            for _ in range(steps):
                self._train_graph.replay()
                if self._train_graph_b is not None:
                    self._allreduce_grads()
                    self._train_graph_b.replay()
                self.n_updates += 1
            self.last_loss = self._g_loss
            return self._g_loss
        loss = None
        for _ in range(steps):
            batch = self._sample()
            loss = self._backward(batch)
            if self.world > 1:                                     # replicas stay identical: one all-reduce of the flat gradient
                self._allreduce_grads()
            self._apply()
            self.n_updates += 1
            self._warm_upd += 1
        self.last_loss = loss
        return self.last_loss

    def learn(self, total_timesteps=None, callback=None):
        """SB3 OffPolicyAlgorithm.learn: rollouts of `train_freq` vector steps, each followed -- once `learning_starts`
        transitions have been collected -- by `gradient_steps` updates."""
        if total_timesteps is not None:
            self.total_timesteps = int(total_timesteps)
        while self.num_timesteps < self.total_timesteps:
            self.collect(self.train_freq)
            if self.num_timesteps > self.learning_starts and self.ring.sampleable() >= self.k + 2:
                self.train()
            if callback is not None and callback(self) is False:
                break
        return self

    # ---- checkpoints ----------------------------------------------------------------------------------
    def _moments(self):
        """name -> (exp_avg, exp_avg_sq) views and the optimiser's step count, whichever optimiser holds them."""
        named = dict(self.q.named_parameters())
        if self._mlp is not None:
            head = [(f"head.{n}", p) for n, p in self.q.head.named_parameters()]
            extra = [(f"features.{n}", p) for n, p in self.q.features.named_parameters()] if self._hybrid else []
            out, off = {}, 0
            for name, p in head + extra:
                n = p.numel()
                out[name] = (self._mlp.exp_avg[off:off + n].view_as(p), self._mlp.exp_avg_sq[off:off + n].view_as(p))
                off += n
            assert off == self._mlp.n_params and set(out) == set(named)
            return out, float(self._mlp.scalars[N.UPD_STEP])
        out, step = {}, 0.0
        for name, p in named.items():
            st = self.opt.state.get(p)
            if st:
                out[name] = (st["exp_avg"], st["exp_avg_sq"])
                step = float(st["step"])
        return out, step

    def save(self, path):
        """What the reference's `model.save(...)` keeps of a DQN (dqn.py:1181, :1339: policy + optimiser + counters), as ONE
        file of plain tensors and numbers (`torch.load(..., weights_only=True)` reads it): both networks, Adam's moments and
        step, the time-step / update counters the schedules run on, the sampling generator.  Not the replay ring (SB3 keeps that
        apart too, `save_replay_buffer`): a resumed learner collects `n_stack + 2` vector steps before its first update."""
        moments, step = self._moments()
        ck = {"format": 1, "extractor": "attention" if isinstance(self.q.features, AttentionFeatures) else "mlp", "n_stack": self.k, "obs_dim": self.D,
              "q": {k_: v.detach().clone() for k_, v in self.q.state_dict().items()},
              "q_target": {k_: v.detach().clone() for k_, v in self.q_target.state_dict().items()},
              "exp_avg": {k_: m[0].detach().clone() for k_, m in moments.items()},
              "exp_avg_sq": {k_: m[1].detach().clone() for k_, m in moments.items()},
              "optimizer_step": step, "num_timesteps": self.num_timesteps, "n_calls": self.n_calls, "n_updates": self.n_updates,
              "total_timesteps": self.total_timesteps, "generator_state": self.gen.get_state(),
              "act_counter": float(self._act_counter) if self.__dict__.get("_act_counter") is not None else 0.0}
        tmp = f"{path}.{os.getpid()}.tmp"
        torch.save(ck, tmp)
        os.replace(tmp, path)

    def load(self, path):
        """Restore `save`'s file into this learner (same architecture; either update path may have written it).  In place: the
        captured graphs, the fused inference weight blocks' addresses and the library's flat buffers stay valid."""
        ck = torch.load(path, map_location=self.dev, weights_only=True)
        assert ck["format"] == 1 and ck["n_stack"] == self.k and ck["obs_dim"] == self.D, "checkpoint of another configuration"
        self.q.load_state_dict(ck["q"])
        self.q_target.load_state_dict(ck["q_target"])
        if self._mlp is None and not self.opt.state:         # torch Adam creates its state at the first step: create it now
            for p in self.q.parameters():
                p.grad = torch.zeros_like(p)
            lr = [g["lr"].clone() if torch.is_tensor(g["lr"]) else g["lr"] for g in self.opt.param_groups]
            self._set_lr(0.0)
            self.opt.step()                                    # (learning rate 0, zero gradients: changes nothing but the state)
            for g, v in zip(self.opt.param_groups, lr):
                (g["lr"].copy_(v) if torch.is_tensor(g["lr"]) else g.__setitem__("lr", v))
            self.opt.zero_grad(set_to_none=True)
        moments, _ = self._moments()
        for name, (m1, m2) in moments.items():
            m1.copy_(ck["exp_avg"][name]); m2.copy_(ck["exp_avg_sq"][name])
        if self._mlp is not None:
            self._mlp.scalars[N.UPD_STEP] = float(ck["optimizer_step"])
        else:
            for st in self.opt.state.values():
                st["step"].fill_(float(ck["optimizer_step"])) if torch.is_tensor(st["step"]) else st.__setitem__("step", ck["optimizer_step"])
        self.num_timesteps, self.n_calls, self.n_updates = int(ck["num_timesteps"]), int(ck["n_calls"]), int(ck["n_updates"])
        self.gen.set_state(ck["generator_state"].cpu())
        if self.__dict__.get("_act_counter") is None:
            self._act_counter = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self._act_counter.fill_(float(ck["act_counter"]))
        if self._fused is not None:
            self._fused.refresh(self.q.features)
        return self

    # ---- evaluation -----------------------------------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, eval_env, episodes_per_env=1, policy="greedy"):
        """Mean return (sum of the rewards the step returns) over the first `episodes_per_env` complete episodes of every
        environment of `eval_env`; policy "greedy" (argmax Q, SB3 predict(deterministic=True)) or "random"."""
        E, D = eval_env.num_envs, eval_env.obs_dim
        fs = FrameStack(E, D, self.k, self.dev)
        stacked = fs.reset(eval_env.reset())
        done_count = torch.zeros(E, device=self.dev)
        returns = []
        cur = torch.zeros(E, dtype=torch.float64, device=self.dev)
        gen = torch.Generator(device=self.dev).manual_seed(12345)
        guard = 0
        while bool((done_count < episodes_per_env).any()) and guard < 100000:
            guard += 1
            if policy == "greedy":
                a = self.q.q_inference(stacked).argmax(1).to(torch.int32)
            else:
                a = torch.randint(0, 5, (E,), device=self.dev, dtype=torch.int32, generator=gen)
            o, r, d = eval_env.step(a)
            cur += r
            db = d.bool()
            take = db & (done_count < episodes_per_env)
            if bool(take.any()):
                returns.append(cur[take].clone())
            done_count += db
            cur = torch.where(db, torch.zeros_like(cur), cur)
            stacked = fs.step(o, d, eval_env.terminal_obs)
        allr = torch.cat(returns)
        return float(allr.mean()), int(allr.numel())

    @torch.no_grad()
    def evaluate_episodes(self, eval_env, policy="greedy", max_steps=4200):
        """One complete episode of every environment of `eval_env` (auto-reset on) under `policy` -- "greedy" (argmax Q, SB3
        predict(deterministic=True)), "random", or an on-device heuristic id (_native.POLICY_NEAREST / _MAX_THROUGHPUT_V2) -- and
        what the reference's curriculum gate looks at (dqn.py:921-984): means of the episode return, NDR (% of sensors visited),
        Jain's index, bytes collected and episode length over the environments' FIRST episodes."""
        E = eval_env.num_envs
        fs = FrameStack(E, eval_env.obs_dim, self.k, self.dev)
        stacked = fs.reset(eval_env.reset())
        finished = torch.zeros(E, dtype=torch.bool, device=self.dev)
        first = None
        gen = torch.Generator(device=self.dev).manual_seed(12345)
        for _ in range(max_steps):
            if policy == "greedy":
                a = self.q.q_inference(stacked).argmax(1).to(torch.int32)
                o, r, d = eval_env.step(a)
            elif policy == "random":
                o, r, d = eval_env.step(torch.randint(0, 5, (E,), device=self.dev, dtype=torch.int32, generator=gen))
            else:
                o, r, d = eval_env.step_policy(int(policy))
            stacked = fs.step(o, d, eval_env.terminal_obs)
            db = d.bool()
            newly = db & ~finished
            if bool(newly.any()):
                st = eval_env.episode_stats()
                if first is None:
                    first = st.copy()
                idx = newly.cpu().numpy()
                first[idx] = st[idx]
                finished |= db
                if bool(finished.all()):
                    break
        assert first is not None and bool(finished.all()), "not every environment finished an episode"
        return {"episodes": int(E), "mean_return": float(first["episode_return"].mean()),
                "ndr": float((first["sensors_visited"] / first["num_sensors"] * 100).mean()),
                "jains": float(first["jains_index"].mean()), "mean_collected_bytes": float(first["total_collected"].mean()),
                "mean_lost_bytes": float(first["total_lost"].mean()), "mean_episode_length": float(first["length"].mean())}

