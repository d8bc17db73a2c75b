// uavenv_noise.h -- counter-based noise for the batched UAV-IoT environment (device + host).
//
// Replaces the reference's three process-global RNG streams (np.random.normal at
// iot_sensors.py:192, stdlib random.random() at uav_env.py:549, gymnasium np_random.uniform at
// uav_env.py:410) by Philox4x32-7 (UAVENV_PHILOX_ROUNDS, include/uavenv.h) keyed by the 64-bit seed and addressed by
//     counter = (global env index, episode, step, lane | call << 16).
//
//   call 0 (every step; step 0 = the observation built inside reset):
//            w0,w1 -> (zD, zE) observation ADR sample / in-range sample;  w2 -> lottery uniform u;
//            lane 0's w3 -> the uniform-random policy's action of the NEXT step (scaled to 0..4)
//   call 1 (collect steps only): w0,w1 -> (zA, zB);  w2,w3 -> (zC, unused)
//   call 2 (reset, step 0):      w0 -> buffer-fill uniform; w1,w2 -> sensor layout x,y;
//                                lane 0's w3 -> curriculum grid choice
//   call 4 (lane field = try):   w0,w1 -> far-start candidate
//   call 5 (policy steps):       w0,w1 -> (zP, -) in-range sample drawn by a heuristic policy before the step
//   call 6 (reset, lane 0):      w0,w1 -> (zS, -) the sample behind the SF that fresh DomainRandEnv sensors inherit
//
// Normals come from a Box-Muller transform written WITHOUT transcendental instructions: only IEEE
// float32 +,-,*,fma,sqrt and integer ops in a fixed order, compiled with -ffp-contract=off, so the
// device and any IEEE host produce bit-identical values (the CPU oracle restates the same
// specification independently in oracle/uavenv_oracle.c).
#pragma once
#include <stdint.h>
#include "../../include/uavenv.h"       // UAVENV_PHILOX_ROUNDS

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UAV_HD __host__ __device__ __forceinline__
#else
#define UAV_HD inline
#endif

namespace uavenv {

struct Words4 { uint32_t w0, w1, w2, w3; };

UAV_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

template <int kRounds>
UAV_HD Words4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Words4{c0, c1, c2, c3};
}

UAV_HD Words4 noise_words(uint64_t seed, uint32_t env, uint32_t episode, uint32_t step, uint32_t lane, uint32_t call) {
#ifdef UAV_ABL_PHILOX      // timing-only ablation build (tools/ablate.py): fewer rounds for the per-lane draws only
    if (call <= 1u) return philox4x32<UAV_ABL_PHILOX>(env, episode, step, lane | (call << 16), (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
    return philox4x32<UAVENV_PHILOX_ROUNDS>(env, episode, step, lane | (call << 16), (uint32_t)seed, (uint32_t)(seed >> 32));
}

UAV_HD float u24(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }      // [0,1), 24 bits, exact

UAV_HD float bits_to_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { uint32_t u; float f; } x; x.u = u; return x.f;
#endif
}
UAV_HD uint32_t float_to_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { uint32_t u; float f; } x; x.f = f; return x.u;
#endif
}
// Correctly rounded float32 square root on both sides.  NOTE: on gfx950 `__fsqrt_rn` lowers to a bare
// v_sqrt_f32 (1 ulp); `__builtin_sqrtf` gets the fma fix-up sequence and IS correctly rounded (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt), which numpy's float32 sqrt requires.
// On the device the generic expansion also rescales denormal inputs and patches 0/inf with a class test (17
// VALU instructions); this path only ever takes the root of 0 or of normal numbers far from the range ends, so
// `sqrt_rn` keeps the part that matters (9 instructions): the hardware estimate (<= 1 ulp) and the exact-residual
// choice between it and its two neighbours.  tools/check_sqrt.hip compares it with __builtin_sqrtf over every
// float: 0 mismatches for x in {0} U [2^-104, 2^127] (below that the residuals underflow; the path's arguments
// are 0 or lie in [1e-8, 1e12]).
UAV_HD float sqrt_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UAV_ABL_LIBSQRT)
    const float y = __builtin_amdgcn_sqrtf(x);
    const float y_dn = __uint_as_float(__float_as_uint(y) - 1u), y_up = __uint_as_float(__float_as_uint(y) + 1u);
    const float r_dn = __builtin_fmaf(-y_dn, y, x), r_up = __builtin_fmaf(-y_up, y, x);
    float r = r_dn <= 0.0f ? y_dn : y;        // x = 0: y_dn is a NaN pattern, the comparison is false
    r = r_up > 0.0f ? y_up : r;
    return r;
#else
    return __builtin_sqrtf(x);
#endif
}

// IEEE fused multiply-add (one rounding), explicit: the file is compiled with -ffp-contract=off, so nothing else fuses.
UAV_HD float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Two standard normals from two 32-bit words.
//   radius: u1 = ((a>>8)+1) * 2^-24 in (0,1];  -ln u1 by exponent split + degree-9 minimax polynomial
//           on the mantissa folded into [sqrt(1/2), sqrt(2));
//   angle:  top 2 bits of (b>>8) choose the quadrant, the low 22 bits the angle in [-pi/4, pi/4);
//           sin/cos by degree-7/8 minimax polynomials.
UAV_HD void normal_pair(uint32_t a, uint32_t b, float& z0, float& z1) {
#ifdef UAV_ABL_NORMAL      // timing-only ablation build
    z0 = (float)(a >> 8) * 0x1p-23f - 1.0f; z1 = (float)(b >> 8) * 0x1p-23f - 1.0f; return;
#endif
#if defined(UAV_HW_TRANSCENDENTALS) && defined(__HIP_DEVICE_COMPILE__)
    // Experiment (tools/exp.sh hwtrans=-DUAV_HW_TRANSCENDENTALS): Box-Muller on the hardware's v_log_f32 / v_sqrt_f32 / v_sin_f32 /
    // v_cos_f32 (1 ulp-ish, NOT reproducible by the CPU oracle: the keyed parity tests would have to consume device-dumped tapes).
    // Measured worth: DESIGN.md section 4 (round 3).
    {
        const float u1h = (float)((a >> 8) + 1u) * 0x1p-24f;
        const float rh = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1h));     // -2 ln u = -2 ln 2 * log2 u
        const float turns = (float)(b >> 8) * 0x1p-24f;                                          // v_sin / v_cos take revolutions
        z0 = rh * __builtin_amdgcn_cosf(turns); z1 = rh * __builtin_amdgcn_sinf(turns);
        return;
    }
#endif
    uint32_t k = (a >> 8) + 1u;
    float u1 = (float)k * 0x1p-24f;
    uint32_t bits = float_to_bits(u1);
    int ex = (int)(bits >> 23) - 127;
    float m = bits_to_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; ex += 1; }
    float t = m - 1.0f;
    float z = t * t;
    float p = 7.0376836292E-2f;
    p = fma32(p, t, -1.1514610310E-1f);
    p = fma32(p, t, 1.1676998740E-1f);
    p = fma32(p, t, -1.2420140846E-1f);
    p = fma32(p, t, 1.4249322787E-1f);
    p = fma32(p, t, -1.6668057665E-1f);
    p = fma32(p, t, 2.0000714765E-1f);
    p = fma32(p, t, -2.4999993993E-1f);
    p = fma32(p, t, 3.3333331174E-1f);
    float y = (t * z) * p;
    y = fma32(-0.5f, z, y);
    float ln = fma32((float)ex, 0.693147182f, t + y);
    float r2 = -2.0f * ln;
    if (!(r2 > 0.0f)) r2 = 0.0f;
    float r = sqrt_rn(r2);

    uint32_t kb = b >> 8;
    uint32_t q = kb >> 22;
    float f = (float)(kb & 0x003FFFFFu) * 0x1p-22f;
    float phi = (f - 0.5f) * 1.57079637f;
    float zz = phi * phi;
    float s = -1.9515295891E-4f;
    s = fma32(s, zz, 8.3321608736E-3f);
    s = fma32(s, zz, -1.6666654611E-1f);
    s = fma32(s * zz, phi, phi);
    float c = 2.443315711809948E-5f;
    c = fma32(c, zz, -1.388731625493765E-3f);
    c = fma32(c, zz, 4.166664568298827E-2f);
    c = fma32(c * zz, zz, fma32(-0.5f, zz, 1.0f));
    float cs = (q & 1u) ? s : c;
    float sn = (q & 1u) ? c : s;
    if (q == 1u || q == 2u) cs = -cs;
    if (q >= 2u) sn = -sn;
    z0 = r * cs;
    z1 = r * sn;
}

}  // namespace uavenv
