"""GPU box: fused UAVAttentionExtractor forward (csrc/uavenv_attention.hip) vs the eager PyTorch module, same weights,
batch 256 and 4096, with the kernel's roofline line: FLOPs and bytes per sample are counted from the architecture
(dqn.py:548-650), the time is HIP events around back-to-back launches on the launch stream.
    python3 tools/attention_rate.py [n_stack ...]        (under rocprofv3 --kernel-trace --stats for the kernel-side time)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd.learner import AttentionFeatures

FP32_VECTOR_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md (the f32-input MFMA runs at the same rate)
L2_PEAK_TBPS = 34.5


def flops_per_sample(k):
    uav = 2 * 3 * k * 64                              # Linear(3k -> 64)
    q = 2 * 64 * 64                                   # query projection
    qk = 2 * 64 * 64                                  # key projection folded into the query (4 heads x 16 x 64)
    tokens = 2 * (50 * 64 * 4)                        # e_s[i] = relu(w_i . token_s + b_i), evaluated in both sweeps (3 fma + relu)
    scores = 2 * 4 * 64 * 50                          # score[s, h] += qk_h[i] * e_s[i]
    mix = 2 * 4 * 64 * 50                             # mix_h[i] += a[s, h] * e_s[i]
    v = 2 * 64 * 64                                   # value projection of the mixes (one head per 16 lanes)
    o = 2 * 64 * 64                                   # output projection
    fuse = 2 * 128 * 128
    return uav + q + qk + tokens + scores + mix + v + o + fuse


out = {"note": "16 samples per workgroup; shared-weight projections on v_mfma_f32_16x16x4_f32, weights (fp32) read once per workgroup through L2"}
for k in [int(a) for a in sys.argv[1:]] or [4, 10]:
    m = AttentionFeatures(k).cuda().eval()
    fused = U.FusedAttentionFeatures(m, k, "cuda:0")
    wfloats = fused.weights.numel()
    for B in (16, 256, 1024, 4096, 16384):
        x = torch.rand(B, k * 153, device="cuda")

        def t(f, n=300):
            for _ in range(20):
                f(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f(x)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        with torch.no_grad():
            eager, hip = t(m), t(fused)
        fl = flops_per_sample(k) * B
        hbm = (k * 153 * 4 + 128 * 4) * B + wfloats * 4                 # observations in, features out, weights once
        cache = wfloats * 4 * ((B + 15) // 16)                           # every workgroup reads the whole block once
        out[f"n_stack_{k}_batch_{B}"] = {
            "torch_eager_us": eager, "fused_hip_us": hip, "speedup": eager / hip,
            "flops": fl, "achieved_TFLOPs": fl / (hip * 1e-6) / 1e12, "frac_of_fp32_vector_peak": fl / (hip * 1e-6) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            "algorithmic_hbm_bytes": hbm, "achieved_hbm_GBps": hbm / (hip * 1e-6) / 1e9,
            "weight_bytes_streamed_through_cache": cache, "cache_TBps": cache / (hip * 1e-6) / 1e12,
            "frac_of_L2_peak": cache / (hip * 1e-6) / 1e12 / L2_PEAK_TBPS}
print(json.dumps(out))
