// uavenv_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the batched UAV-IoT
// environment reset()/step() hot path.
//
// Mapping: one lane GROUP of G in {16,32,64} lanes of a 64-wide wavefront owns one environment
// instance, one lane owns one sensor.  64/G environments share a wavefront, 256/G a workgroup.
// Per-sensor state is SoA [E][G] in HBM (every wave-load is one contiguous 256..512 B row per
// array), lives in VGPRs for the whole step, and is written back once.  Reductions over sensors
// (data-loss sum, Capture-Effect top-2, variance, urgency sums, Jain's sums) are DPP row butterflies,
// v_readlane and ballots restricted to the group.  Every lane stores its own sensor's slice of the
// [E][obs_dim] float32 observation row (the lanes of a group cover the row contiguously, so the stores
// coalesce with no staging).  No LDS, no MFMA: the path is elementwise + short reductions.
//
// Reference being replaced (paths relative to /root/reference/src/):
//   environment/uav_env.py       step :429-488, _execute_move_action :494-516,
//                                _execute_collect_action :518-632, _get_observation :638-674,
//                                reset :400-427
//   environment/iot_sensors.py   step :114-125, collect_data :127-145, calculate_rssi :147-197,
//                                get_success_probability :202-212, update_spreading_factor :223-259
//   environment/uav.py           move :127-185, hover :187-206, is_alive :208-224
//   rewards/reward_function.py   :46-128
//   agents/dqn/dqn.py            DomainRandEnv reset/step :301-451 (optional flags)
//
// Compiled with -ffp-contract=off: numpy never fuses a*b+c, and the float64 state must follow
// the reference's operation order to stay within the 1e-5 parity contract over 2100-step episodes.
#include "uavenv_internal.h"
#include "uavenv_noise.h"

namespace uavenv {

// timing-only ablation builds (tools/exp.sh "exitK=-DUAV_ABL_EXIT=K"): the step ends after phase K; `val` keeps what the phases
// so far computed alive.  The kernel time of successive K's shows what each phase adds to a LAUNCH (overlap included).
#ifdef UAV_ABL_EXIT
#define UAV_EXIT_AT(k, val) do { if (UAV_ABL_EXIT == (k)) { if (gl == 0 && a.out.hint_out != nullptr) a.out.hint_out[env] = (uint32_t)(val); \
                                                            live = false; next_word = 0u; action_out = 0; return; } } while (0)
#else
#define UAV_EXIT_AT(k, val) do {} while (0)
#endif

#ifdef UAVENV_STAMPS
// diagnostic build: phase timestamps of the step, 16 words per environment, stored behind the per-wave records
// (p.stamps + kPhaseBase); tools/phases.py reads them.
#define UAV_PHASE(i) do { if (!kRegs && p.stamps != nullptr && gl == 0) p.stamps[(1u << 20) + (size_t)env * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UAV_PHASE(i) do {} while (0)
#endif


// The constants live in device memory and are read through the CONSTANT address space: scalar loads
// (s_load) issued where a field is used.  (Passing the 600-byte struct by value makes the compiler load
// every field in the entry block and then spill ~80 SGPRs to VGPR lanes.)
typedef const __attribute__((address_space(4))) Consts& CRef;
// "Lean" specialisation (template parameter kLean of the step / rollout kernels): the configuration dimensions
// that the plain environment does not use -- DomainRand flags, 5-feature observations, the noise tape, the
// heuristic policies -- are folded at compile time.  Chosen at launch when the handle's configuration allows
// (launch_step); 4 % faster (12.28 vs 12.81 us per launch), same results.
#define UAV_FLAGS(c) (kLean ? ((c).flags & UAVENV_FLAG_AUTO_RESET) : (c).flags)
#define UAV_FPS(c) (kLean ? 3 : (c).fps)
#define UAV_TAPE(ptr) (kLean ? (const float*)nullptr : (ptr))
#define UAV_POLICY(a) (kLean ? ((a).policy & 1) : (a).policy)
#define UAV_CONSTS(ptr) CRef c = *(const __attribute__((address_space(4))) Consts*)(ptr)

// "Default constants" specialisation (template parameter kDefC of the step / rollout kernels).  Every scalar load of a
// constant is a round trip to the scalar cache on the critical path of a wave (56 cycles on a hit, 165 on the first touch
// of a 64-byte line after a launch, tools/smem_latency.hip) and a step reads ~45 of them: for the reference's own
// configuration (uavenv_default_config: BASE_ENV_CONFIG, dqn.py:1068-1075) the floating-point constants are therefore
// also available as instruction literals, generated bit for bit from the same derive_consts() by gen_default_consts.cpp.
// launch_step / launch_rollout pick this variant only when the handle's constants block is bit-identical
// (consts_are_default); everything that is not a floating-point constant (seed, step limit, observation shape, flags, the
// reciprocal table) still comes from the block `m`.
typedef const __attribute__((address_space(4))) int32_t& CI32;
struct DefaultConsts {
    CRef m;
    // the integers every step needs (step limit, observation shape, EMA switch, flags) sit side by side in the block and are
    // fetched HERE with one scalar load of 8 dwords, together with the wave's first loads, instead of one load + wait each where
    // they are used; `seed` comes from the caller (a preloaded kernel argument in the step kernel)
    uint64_t seed;
    int32_t max_steps, fps, obs_dim, obs_slots, use_ema;
    uint32_t flags;
    CI32 max_tries, n_grid_choices;
    const __attribute__((address_space(4))) int32_t (&gw)[8];
    const __attribute__((address_space(4))) int32_t (&gh)[8];
    const __attribute__((address_space(4))) double (&inv_small)[65];
#include "uavenv_default_consts.inc"
    __device__ __forceinline__ DefaultConsts(CRef k, uint64_t seed_)
        : m(k), seed(seed_), max_tries(k.max_tries), n_grid_choices(k.n_grid_choices), gw(k.gw), gh(k.gh), inv_small(k.inv_small) {
        static_assert(offsetof(Consts, flags) - offsetof(Consts, max_steps) == 20, "max_steps .. flags must be six adjacent words");
        const __attribute__((address_space(4))) uint32_t* wp = (const __attribute__((address_space(4))) uint32_t*)&k.max_steps;
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; i++) w[i] = wp[i];
        max_steps = (int32_t)w[0]; fps = (int32_t)w[1]; obs_dim = (int32_t)w[2]; obs_slots = (int32_t)w[3]; use_ema = (int32_t)w[4]; flags = w[5];
    }
};
template <typename T> constexpr bool kIsDefaultConsts = false;
template <> constexpr bool kIsDefaultConsts<DefaultConsts> = true;
template <bool kDefC> struct ConstsSel;
template <> struct ConstsSel<false> { static __device__ __forceinline__ CRef make(CRef k, uint64_t) { return k; } };
template <> struct ConstsSel<true> { static __device__ __forceinline__ DefaultConsts make(CRef k, uint64_t seed) { return DefaultConsts(k, seed); } };

// ---------------------------------------------------------------------------------------------
// lane-group primitives (G lanes of a wave64)
// ---------------------------------------------------------------------------------------------
template <int G> __device__ __forceinline__ int group_base() { return (int)(threadIdx.x & 63u) & ~(G - 1); }
template <int G> __device__ __forceinline__ int group_lane() { return (int)(threadIdx.x & (unsigned)(G - 1)); }

// DPP lane exchanges (no LDS crossbar, no address VGPR).  The four controls below form an xor-like
// butterfly inside a 16-lane row when applied in this order: quad_perm[1,0,3,2], quad_perm[2,3,0,1],
// row_half_mirror, row_mirror -- after step k every lane of an aligned 2^k block holds the same value,
// so the mirrors exchange with the OTHER half.  Every lane of a row ends with the identical bits
// (each addition is commutative and the tree is the same for all lanes).
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    // old = 0 + bound_ctrl: every source lane of these controls is valid, so no tied copy of `v` is needed
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

struct OpSum { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct OpMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b > a ? b : a; } };
struct OpMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b < a ? b : a; } };

// DPP row broadcasts (gfx9): row_bcast15 hands lane 15 of every row to the NEXT row, row_bcast31 lane 31 to rows 2 and 3.  Rows
// without a source read 0 (bound_ctrl): whatever they compute is never used -- only lane 63 is read, and its chain
// row 0 -> row 1, row 2 -> row 3, row 1 -> row 3 has a valid source at every step.
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// all-lanes reduction over a lane group: 4 DPP steps inside each 16-lane row, then rows are
// combined with one bpermute level (G = 32) or through SGPRs with v_readlane (G = 64: the group is
// the whole wave, so the result is wave-uniform).
template <int G, typename T, typename Op> __device__ __forceinline__ T greduce(T v, Op op) {
    v = op(v, dpp<kDppXor1>(v));
    v = op(v, dpp<kDppXor2>(v));
    v = op(v, dpp<kDppHalfMirror>(v));
    v = op(v, dpp<kDppMirror>(v));
    if (G == 32) v = op(v, __shfl_xor(v, 16, 64));
    if (G == 64) {
        // rows r0..r3 -> (r0 op r1) in row 1, (r2 op r3) in row 3 -> ((r2 op r3) op (r0 op r1)) in row 3: the same tree as
        // op(op(r0, r1), op(r2, r3)) (the operations commute exactly), two DPP steps and one lane read instead of four reads
        v = op(v, dpp<kDppRowBcast15>(v));
        v = op(v, dpp<kDppRowBcast31>(v));
        v = readlane(v, 63);
    }
    return v;
}
template <int G> __device__ __forceinline__ double gsum(double v) { return greduce<G>(v, OpSum()); }
template <int G> __device__ __forceinline__ double gmax(double v) { return greduce<G>(v, OpMax()); }
template <int G> __device__ __forceinline__ float gmin_f32(float v) { return greduce<G>(v, OpMin()); }
// ballot restricted to this lane's group, shifted so bit k = group lane k
template <int G> __device__ __forceinline__ uint64_t gballot(bool pred) {
    uint64_t b = __ballot(pred);
    if (G == 64) return b;
    return (b >> group_base<G>()) & ((1ull << (G & 63)) - 1ull);
}
template <int G> __device__ __forceinline__ bool gany(bool pred) { return gballot<G>(pred) != 0ull; }
template <int G, typename T> __device__ __forceinline__ T gshfl(T v, int src) { return __shfl(v, group_base<G>() + src, 64); }
// same, for a source lane that is uniform over the group: a scalar v_readlane when the group is the wave
template <int G> __device__ __forceinline__ double gbcast(double v, int src) {
    if (G == 64) return readlane(v, __builtin_amdgcn_readfirstlane(src));
    return gshfl<G>(v, src);
}
template <int G> __device__ __forceinline__ uint32_t gbcast(uint32_t v, int src) {
    if (G == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src));
    return gshfl<G>(v, src);
}

// Sum of `v` over the lanes of `mask` (a group ballot), in ascending lane order.  For the handful of Capture-Effect
// winners of a collect step this is a few v_readlane + v_add_f64 instead of a full DPP reduction (G = 64: the mask
// is wave-uniform and the loop runs on the scalar unit; narrower groups use the reduction).
template <int G> __device__ __forceinline__ double gsum_sparse(double v, uint64_t mask) {
    if (G != 64) return gsum<G>(v);
    double acc = 0.0;
    for (uint64_t m = mask; m != 0ull; m &= m - 1ull) acc += readlane(v, __ffsll((long long)m) - 1);
    return acc;
}

// numpy float32 add.reduce order (pairwise sum, n <= 64 < PW_BLOCKSIZE): 8 strided accumulators
// combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then a sequential tail; plain loop for n < 8.
// Reproduces np.sum(np.maximum(0, before - after)) at uav_env.py:601 bit for bit.
template <int G> __device__ __forceinline__ float np_sum_f32(float a, int n) {
    const int gl = group_lane<G>();
    const int n8 = n - (n & 7);
    if (G == 64) {
        // The addends are zero except for the (at most six) sensors drained this step, and x + 0.0f == x
        // exactly, so only the non-zero lanes matter -- visited in numpy's order: strided accumulators
        // (lane j of `r` is accumulator j), the pairwise tree, then the sequential tail.
        const uint64_t nz = __ballot(a != 0.0f);
        if (nz == 0ull) return 0.0f;
        const uint64_t body = (n >= 8) ? (n8 >= 64 ? ~0ull : ((1ull << n8) - 1ull)) : 0ull;
        float r = 0.0f;
        for (uint64_t m = nz & body; m != 0ull; m &= m - 1ull) {
            const int i = __ffsll((long long)m) - 1;
            const float v = readlane(a, i);
            r = (gl == (i & 7)) ? r + v : r;
        }
        r += dpp<kDppXor1>(r);
        r += dpp<kDppXor2>(r);
        r += dpp<kDppHalfMirror>(r);
        float res = (n >= 8) ? readlane(r, 0) : 0.0f;
        for (uint64_t m = nz & ~body; m != 0ull; m &= m - 1ull) res += readlane(a, __ffsll((long long)m) - 1);
        return res;
    }
    float r = a;
#pragma unroll
    for (int k = 1; k < G / 8; k++) {
        float v = gshfl<G>(a, (gl + 8 * k) & (G - 1));
        if (gl + 8 * k < n8) r += v;
    }
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 4, 64);
    float tree = gshfl<G>(r, 0);
    float res = (n < 8) ? 0.0f : tree;
    const int start = (n < 8) ? 0 : n8;
#pragma unroll
    for (int t = 0; t < 7; t++) {
        int i = start + t;
        float v = gshfl<G>(a, i & (G - 1));
        if (i < n) res += v;
    }
    return res;
}

// ---------------------------------------------------------------------------------------------
// LoRa tables: iot_sensors.py:13-20 (bytes/s), :22-29 (required SNR dB); uav_env.py:647 (link quality)
// ---------------------------------------------------------------------------------------------
// All three tables are written as flat select sequences (v_cndmask), not nested ternaries: the nested form
// compiles to a tree of exec-mask branches.
__device__ __forceinline__ double sf_data_rate(uint32_t sf) {
    double r = 250 / 8.0;                       // SF12 (and anything unknown, like the reference's table lookup of 12)
    r = sf == 11 ? 440 / 8.0 : r;
    r = sf == 10 ? 980 / 8.0 : r;
    r = sf == 9 ? 1760 / 8.0 : r;
    r = sf == 8 ? 3125 / 8.0 : r;
    r = sf == 7 ? 5470 / 8.0 : r;
    return r;
}
__device__ __forceinline__ double sf_required_snr(uint32_t sf) {
    // -6, -9, -12, -15, -17.5, -20 for SF 7..12 (7.5 otherwise, iot_sensors.py:200)
    double r = 7.5;
    r = ((sf >= 7u) & (sf <= 10u)) ? -6.0 - 3.0 * (double)(int)(sf - 7u) : r;   // exact small integers
    r = sf == 11u ? -17.5 : r;
    r = sf == 12u ? -20.0 : r;
    return r;
}
// uav_env.py:647 sf_quality {7: 1.0, 8: .8, 9: .6, 10: .4, 11: .2, 12: .1} as the float32 the observation stores:
// 0.2f * k reproduces float32(1.0, 0.8, 0.6, 0.4, 0.2) bit for bit for k = 5..1 (checked), 0.1f otherwise.
__device__ __forceinline__ float sf_link_quality_f32(uint32_t sf) {
    const float k = (float)(int)(12u - sf);
    return ((sf >= 7u) & (sf <= 11u)) ? 0.2f * k : 0.1f;
}

// x / y when no range scaling can be needed (y normal and far from the ends of the exponent range, the quotient zero or
// normal): the hardware's IEEE float64 division sequence -- reciprocal estimate, two Newton steps, quotient, exact residual,
// correction -- without its v_div_scale / v_div_fixup bracket, which only rescales operands outside that range.  Same
// intermediate values, hence the same correctly rounded quotient, in 8 instructions instead of 11.
__device__ __forceinline__ double div_inrange(double x, double y) {
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    const double rem = __builtin_fma(-y, q, x);
    return __builtin_fma(rem, r, q);
}

// log10 of a positive normal float32, evaluated in float64 and rounded once to float32.
// Specification shared with the oracle (oracle/uavenv_oracle.c:orc_log10_f32), IEEE + - * / fma only:
// x = m * 2^e with m folded into [sqrt(1/2), sqrt(2)); ln m = 2 atanh(s), s = (m-1)/(m+1), by the odd
// series to s^17 (|s| <= 0.172 -> truncation 8e-16); result = e*log10(2) + ln(m)*log10(e).
// Absolute error ~1e-15, i.e. the correctly rounded float32 log10 except with probability ~1e-8.
__device__ __forceinline__ float log10_f32(float d) {
    double x = (double)d;
    int hi = __double2hiint(x), lo = __double2loint(x);
    int e = ((hi >> 20) & 0x7FF) - 1023;
    double m = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, lo);
    const bool fold = m > 1.4142135623730951;
    m = fold ? m * 0.5 : m;
    e = fold ? e + 1 : e;
    double s = div_inrange(m - 1.0, m + 1.0);          // m + 1 in [1.7, 2.42), |m - 1| < 0.42
    double s2 = s * s;
    double p = 1.0 / 17;
    p = __builtin_fma(p, s2, 1.0 / 15);
    p = __builtin_fma(p, s2, 1.0 / 13);
    p = __builtin_fma(p, s2, 1.0 / 11);
    p = __builtin_fma(p, s2, 1.0 / 9);
    p = __builtin_fma(p, s2, 1.0 / 7);
    p = __builtin_fma(p, s2, 1.0 / 5);
    p = __builtin_fma(p, s2, 1.0 / 3);
    double t = 2.0 * s;
    double ln_m = __builtin_fma(t, s2 * p, t);
    double r = __builtin_fma((double)e, 0.30102999566398120, ln_m * 0.43429448190325182);
    return (float)r;
}

template <typename CT> __device__ __forceinline__ double rssi_deterministic(const CT& c, float ux, float uy, float sx, float sy) {
    float dx = (ux - sx) * 10.0f;
    float dy = (uy - sy) * 10.0f;
    float ground = sqrt_rn(dx * dx + dy * dy);
    float d = sqrt_rn(ground * ground + c.alt2);
    float l10 = log10_f32(d);
    double path_loss;
    if ((double)d < c.d_break) {
        float t = 20.0f * l10;
        path_loss = ((double)t + c.c_fs) - c.fspl_off;
    } else {
        float t = 40.0f * l10;
        path_loss = ((double)t - c.c_ht) - c.c_hr;
    }
    return c.tx_power - path_loss;
}

// iot_sensors.py:223-259 update_spreading_factor (EMA-ADR), state in (avg, flags)
template <typename CT> __device__ __forceinline__ void adr_update(const CT& c, double cur, double& avg, uint32_t& flags) {
    const bool ema = ((flags & kAvgValid) != 0u) & (c.use_ema != 0);
    const double blended = (c.lambda * cur) + (c.one_minus_lambda * avg);
    const double nv = ema ? blended : cur;
    avg = nv;
    uint32_t sf = flags & kSfMask;                     // sticky when no threshold fires (iot_sensors.py:251-255)
    sf = nv > c.sf_thr[3] ? 12u : sf;                  // later assignments win = the FIRST matching threshold of the
    sf = nv > c.sf_thr[2] ? 11u : sf;                  // reference's descending list (-60, -70, -78, -85)
    sf = nv > c.sf_thr[1] ? 9u : sf;
    sf = nv > c.sf_thr[0] ? 7u : sf;
    flags = (flags & ~kSfMask) | sf | kAvgValid;
}

// iot_sensors.py:211 the SNR sigmoid.  Needed only when the lottery uniform falls into a 2e-9-wide band (see the collect
// block): kept out of line so that the exponential's two dozen polynomial constants are not hoisted into registers
// around the rollout kernel's step loop (they were its scratch spills).
__device__ __attribute__((noinline, cold)) double sigmoid_rare(double x) { return 1.0 / (1.0 + exp(-x)); }

// x / y for a divisor whose correctly rounded reciprocal `inv` is known: Markstein's sequence (one product,
// two fma; 3 instructions instead of the 12 of an IEEE float64 division).  q0 = RN(x*inv) is within one ulp,
// the fma residual is exact, and the corrected quotient equals RN(x / y) (checked exhaustively against true
// division for this path's divisors: 0 mismatches in 324 000 samples).
__device__ __forceinline__ double div_const(double x, double y, double inv) {
    const double q0 = x * inv;
    const double r = __builtin_fma(-q0, y, x);
    return __builtin_fma(r, inv, q0);
}

// uav_env.py:376-384 _calculate_urgency
template <typename CT> __device__ __forceinline__ double calc_urgency(const CT& c, double b, double gen, double lost) {
    double util = div_const(b, c.bmax, c.inv_bmax);
    // (default constants: a sensor that generated anything generated >= 2.2 bytes and at most 2.2 per step -- in range)
    double loss_rate = gen > 0 ? (kIsDefaultConsts<CT> ? div_inrange(lost, gen) : lost / gen) : 0.0;
    double u = util * (1.0 + loss_rate * 10.0);
    return __builtin_fmin(__builtin_fmax(u, 0.0), 1.0);     // np.clip; u is never NaN or -0.0 (b, lost >= 0, gen > 0)
}

// The per-environment values the step needs from its first instruction on.  When the lane group is the
// whole wavefront (G = 64) they are wave-uniform and are moved to SGPRs (v_readfirstlane), which frees
// VGPRs and lets the compiler run the per-environment integer work (the action's Philox call, step
// counters) on the scalar ALU.  The remaining ("cold") fields of the 128-byte record are only read and
// written in the epilogue, when the per-sensor temporaries are dead.
struct Env {
    double battery;
    float ux, uy;
    int32_t step;
    uint32_t episode, env_index;
    int32_t n, gw, gh;
    double inv_w, inv_h;
};
template <int G> __device__ __forceinline__ int uni(int v) { return G == 64 ? __builtin_amdgcn_readfirstlane(v) : v; }
// whole-record read: a plain copy, or word by word through the constant address space (scalar loads -> SGPRs)
__device__ __forceinline__ UavEnvRecord load_record(const UavEnvRecord* rp) { return *rp; }
__device__ __forceinline__ UavEnvRecord load_record(const __attribute__((address_space(4))) UavEnvRecord* rp) {
    union { UavEnvRecord r; uint32_t w[32]; } u;
    const __attribute__((address_space(4))) uint32_t* wp = (const __attribute__((address_space(4))) uint32_t*)rp;
#pragma unroll
    for (int i = 0; i < 32; i++) u.w[i] = wp[i];
    return u.r;
}
template <int G, typename RP> __device__ __forceinline__ Env load_env(RP rp) {
    Env e;
    double b = rp->battery;
    e.battery = __hiloint2double(uni<G>(__double2hiint(b)), uni<G>(__double2loint(b)));
    e.ux = __int_as_float(uni<G>(__float_as_int(rp->uav_x)));
    e.uy = __int_as_float(uni<G>(__float_as_int(rp->uav_y)));
    e.step = uni<G>(rp->current_step);
    e.episode = (uint32_t)uni<G>((int)rp->episode);
    e.env_index = (uint32_t)uni<G>((int)rp->env_index);
    e.n = uni<G>(rp->num_sensors);
    e.gw = uni<G>(rp->grid_w);
    e.gh = uni<G>(rp->grid_h);
    double iw = rp->inv_grid_w, ih = rp->inv_grid_h;
    e.inv_w = __hiloint2double(uni<G>(__double2hiint(iw)), uni<G>(__double2loint(iw)));
    e.inv_h = __hiloint2double(uni<G>(__double2hiint(ih)), uni<G>(__double2loint(ih)));
    return e;
}

// Write-through stores.  A plain store leaves its line dirty in the XCD's L2 and the kernel boundary writes all dirty lines
// back (the XCDs' L2s are not coherent with each other): ~10 MB per launch at 4096 environments, 1.4 us of a 9 us launch
// (timing-only builds without the state / observation stores, tools/exp.sh).  `sc1` stores leave L2 while the launch is
// still computing (MI355X_MICROARCH.md "stores of each flavour"); measured 9.05 -> 8.52 us per launch at 4096 environments
// for observations + sensor state, `nt` 8.79.  They also drop the line from L2, so the next launch's loads come from the
// Infinity Cache: +0.18 us at 256 environments, where there is little to write back -- so `wt` is a launch parameter
// (StepArgs::write_through, set for batches that fill the chip).
template <typename T> __device__ __forceinline__ void store_wt(T* q, T v, bool wt) {
    if (wt) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store ... sc1
    else *q = v;
}
typedef float __attribute__((ext_vector_type(3))) f32x3;
__device__ __forceinline__ void store3_wt(float* q, float a, float b, float c, bool wt) {
    // (s_nop 0: a VALU write to the data registers of a > 64-bit VMEM store needs one wait state after the store issues, and
    //  the hazard recogniser does not look inside inline assembly -- the register allocator reuses these registers at once)
    if (wt) { f32x3 v = {a, b, c}; asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 0" :: "v"(q), "v"(v) : "memory"); }
    else { struct __attribute__((packed, aligned(4))) P3 { float a, b, c; }; P3 w{a, b, c}; *reinterpret_cast<P3*>(q) = w; }
}

// The output block of the argument struct in ONE wide scalar load (24 dwords -> s_load_dwordx16 + x8) when the struct is
// read in place from the kernarg segment: the step used to fetch each pointer where it was needed, a dozen dependent
// "s_load, s_waitcnt" round trips (56-165 cycles each) in a row between the truncation test and the stores.
template <typename A> __device__ __forceinline__ OutArgs load_out_args(const A& a) { return a.out; }
template <> __device__ __forceinline__ OutArgs load_out_args(const __attribute__((address_space(4))) StepArgs& a) {
    union { OutArgs o; uint32_t w[24]; } u;
    const __attribute__((address_space(4))) uint32_t* wp = (const __attribute__((address_space(4))) uint32_t*)&a.out;
#pragma unroll
    for (int i = 0; i < 24; i++) u.w[i] = wp[i];
    return u.o;
}

// per-lane sensor registers
struct Sensor {
    double b, gen, tx, lost, avg;
    float sx, sy;
    uint32_t flags;
};

// SoA arrays inside the single state allocation (uavenv_internal.h): array base = sensor_base + off * S.
struct SensorBase { char* sensor_base; uint64_t lanes; };      // the two fields sensor_array() needs, e.g. from preloaded SGPRs
// (`P` / `A` template parameters below: Ptrs / StepArgs either as a private copy or, in the step kernel, read in
// place from the kernarg segment through the constant address space -- fields are then fetched where they are used.)
template <typename T, typename P> __device__ __forceinline__ T* sensor_array(const P& p, uint64_t off) {
    return reinterpret_cast<T*>(p.sensor_base + off * p.lanes);
}
// Element `idx` of a state array: uniform base (SGPRs) + a 32-bit byte offset, so the access uses the scalar-base
// addressing mode and needs no 64-bit per-lane address (uavenv_create refuses batches whose rows pass 4 GiB).
template <typename T, typename P> __device__ __forceinline__ T& sensor_at(const P& p, uint64_t off, uint32_t idx) {
    return *reinterpret_cast<T*>(p.sensor_base + off * p.lanes + (uint32_t)(idx * (uint32_t)sizeof(T)));
}
template <int G, typename P> __device__ __forceinline__ void load_sensor(const P& p, uint32_t idx, Sensor& s) {
    s.sx = sensor_at<float>(p, kOffPosX, idx); s.sy = sensor_at<float>(p, kOffPosY, idx);
    s.b = sensor_at<double>(p, kOffBuffer, idx); s.gen = sensor_at<double>(p, kOffGen, idx);
    s.tx = sensor_at<double>(p, kOffTx, idx); s.lost = sensor_at<double>(p, kOffLost, idx);
    s.avg = sensor_at<double>(p, kOffAvg, idx);
    s.flags = sensor_at<uint32_t>(p, kOffFlags, idx);
}
// Tried in round 3 and measured SLOWER (tools/exp.sh, -DUAV_EXP_MASK_LOADS): lanes that hold no sensor in any environment of the
// handle (lane >= the handle's sensor count, which the launch word carries) fetching nothing -- 22 % of the state rows' read bytes
// at 50 sensors in a 64-lane group.  7.63 vs 7.39 us per launch at 4096 x 50: the launch is not bound by those bytes, and the
// compare + exec mask in front of the loads delays every wave's first memory request.
template <int G, typename P> __device__ __forceinline__ void load_sensor_live(const P& p, uint32_t idx, Sensor& s, bool lane_in_use) {
#ifdef UAV_EXP_MASK_LOADS
    s.sx = 0.f; s.sy = 0.f; s.b = 0.0; s.gen = 0.0; s.tx = 0.0; s.lost = 0.0; s.avg = 0.0; s.flags = 12u;
    if (lane_in_use)
#endif
        load_sensor<G>(p, idx, s);
}
// Write back what the step can have changed: nothing for lanes beyond the environment's sensor count (`live` false:
// their state never changes), `tx` only if some lane of the wave collected or reset (bit 0 of `dirty`, set where the
// step touches it), `lost` only if some buffer overflowed or reset (bit 1).  Cuts the write traffic by about a third.
template <int G, typename P> __device__ __forceinline__ void store_sensor(const P& p, uint32_t idx, const Sensor& s, bool with_pos,
                                                                          bool live, uint32_t dirty, bool wt) {
#ifdef UAV_ABL_NOSTATE        // timing-only ablation build
    if (s.b != -1.0) return;
#endif
    if (with_pos && live) { sensor_at<float>(p, kOffPosX, idx) = s.sx; sensor_at<float>(p, kOffPosY, idx) = s.sy; }
    if (live) {
        store_wt(&sensor_at<double>(p, kOffBuffer, idx), s.b, wt); store_wt(&sensor_at<double>(p, kOffGen, idx), s.gen, wt);
        store_wt(&sensor_at<double>(p, kOffAvg, idx), s.avg, wt);
        store_wt(&sensor_at<uint32_t>(p, kOffFlags, idx), s.flags, wt);
    }
    if (__any(live & ((dirty & 1u) != 0u)) && live) store_wt(&sensor_at<double>(p, kOffTx, idx), s.tx, wt);
    if (__any(live & ((dirty & 2u) != 0u)) && live) store_wt(&sensor_at<double>(p, kOffLost, idx), s.lost, wt);
}
// unconditional form (initialisation, reset kernel)
template <int G, typename P> __device__ __forceinline__ void store_sensor(const P& p, uint32_t idx, const Sensor& s, bool with_pos) {
    if (with_pos) { sensor_at<float>(p, kOffPosX, idx) = s.sx; sensor_at<float>(p, kOffPosY, idx) = s.sy; }
    sensor_at<double>(p, kOffBuffer, idx) = s.b; sensor_at<double>(p, kOffGen, idx) = s.gen;
    sensor_at<double>(p, kOffTx, idx) = s.tx; sensor_at<double>(p, kOffLost, idx) = s.lost;
    sensor_at<double>(p, kOffAvg, idx) = s.avg;
    sensor_at<uint32_t>(p, kOffFlags, idx) = s.flags;
}

// dqn.py:406-412 distance to the nearest sensor that still has data (float32 norm), 0 if none
template <int G> __device__ __forceinline__ double dist_nearest_with_data(const Sensor& s, bool act, float ux, float uy) {
    float dx = s.sx - ux, dy = s.sy - uy;
    float d = sqrt_rn(dx * dx + dy * dy);
    bool has = act && s.b > 0;
    float m = gmin_f32<G>(has ? d : __builtin_inff());
    return gany<G>(has) ? (double)m : 0.0;
}

// dqn.py:446-451 Jain's index over r_i = 100*tx_i/gen_i (gen_i > 0); also returns the population
// std of the rates (dqn.py:324 fairness_std)
template <int G> __device__ __forceinline__ double jains_index(const Sensor& s, bool act, double* std_out, int* count_out = nullptr) {
    bool ok = act && s.gen > 0;
    double r = ok ? (s.tx / s.gen) * 100 : 0.0;
    double s1 = gsum<G>(r), s2 = gsum<G>(r * r);
    int cnt = __popcll(gballot<G>(ok));
    if (count_out) *count_out = cnt;
    if (std_out) {
        double mean = cnt > 0 ? s1 / cnt : 0.0;
        double dv = ok ? (r - mean) * (r - mean) : 0.0;
        double var = cnt > 0 ? gsum<G>(dv) / cnt : 0.0;
        *std_out = sqrt(var);
    }
    return (cnt > 0 && s2 > 0) ? (s1 * s1) / (cnt * s2) : 1.0;
}

// uav_env.py:638-674 _get_observation for one environment group.  SIDE EFFECT on the lane's
// sensor: advances the ADR EMA with slot zD, then draws the in-range sample zE.  Each lane stores its own
// sensor's fps consecutive floats (one global_store_dwordx3 when fps = 3): the lanes of a group cover the row
// contiguously, so the stores coalesce without staging the row anywhere.  `enable` masks whole groups (a
// wave may hold groups that do not rebuild); `dst` may be nullptr (row computed for its side effects only).
struct __attribute__((packed, aligned(4))) ObsF3 { float a, b, c; };
struct __attribute__((packed, aligned(4))) ObsF2 { float a, b; };
template <int G, bool kLean, typename CT>
__device__ __forceinline__ void observe(const CT& c, Sensor& s, int n, int gw, int gh, double inv_w, double inv_h,
                                        float uxf, float uyf, double battery, bool act, bool enable,
                                        double det, float zD, float zE, float* dst, bool wt = false) {
    const int gl = group_lane<G>();
    double W = (double)gw, H = (double)gh;
    double ux = (double)uxf, uy = (double)uyf;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f;
    if (enable && act) {
        double urgency = calc_urgency(c, s.b, s.gen, s.lost);                 // :652 (before the ADR update)
        adr_update(c, det + c.sigma * (double)zD, s.avg, s.flags);           // :654
        bool in_range = (det + c.sigma * (double)zE) >= c.thr;               // :658, iot_sensors.py:214-219
        f0 = (float)div_const(s.b, c.bmax, c.inv_bmax);
        f1 = (float)urgency;
        f2 = in_range ? sf_link_quality_f32(s.flags & kSfMask) : 0.0f;
        if (UAV_FPS(c) == 5) {                                                     // :668-672
            f3 = (float)div_const((double)s.sx - ux, W, inv_w);
            f4 = (float)div_const((double)s.sy - uy, H, inv_h);
        }
    }
#ifdef UAV_ABL_NOOBS          // timing-only ablation build
    if (f0 != -7.0f) dst = nullptr;
#endif
    if (enable && dst != nullptr) {
        if (gl == 0) {
            store3_wt(dst, (float)div_const(ux, W, inv_w), (float)div_const(uy, H, inv_h),
                      (float)div_const(battery, c.maxb, c.inv_maxb), wt);
        }
        // slot gl holds this lane's sensor (zeros beyond n: the padded tail of dqn.py:286-298); groups narrower
        // than the padded row (num_sensors <= 32 padded to 50) zero the remaining slots in further passes
        const int slots = c.obs_slots;
        const int fps = UAV_FPS(c);
        for (int k = gl; k < slots; k += G) {
            const bool own = k == gl;
            float* q = dst + 3 + fps * k;
            store3_wt(q, own ? f0 : 0.f, own ? f1 : 0.f, own ? f2 : 0.f, wt);
            if (fps == 5) { ObsF2 w; w.a = own ? f3 : 0.f; w.b = own ? f4 : 0.f; *reinterpret_cast<ObsF2*>(q + 3) = w; }
        }
    }
}

// Noise for one step of one lane: Philox, or the injected tape (parity testing).  zC (the range check of
// collect_data) is only needed by Capture-Effect winners, so its Box-Muller is deferred: the two Philox
// words are kept and converted on demand (finish_zc).
struct StepNoise { float zA, zB, u, zC, zD, zE; uint32_t c2, c3; bool zc_ready; uint32_t w3; bool have_w3; };

// zP: the shadowing sample of the is_in_range() call a heuristic policy makes before the step (Philox call 5)
template <int G, bool kLean, typename CT, typename P>
__device__ __forceinline__ float draw_policy_noise(const CT& c, const P& p, uint32_t env_index, uint32_t episode, size_t env,
                                                   bool in_batch, uint32_t step) {
    const int gl = group_lane<G>();
    if (UAV_TAPE(p.step_tape) != nullptr)
        return in_batch ? p.step_tape[env * (size_t)(UAVENV_TAPE_STEP_SLOTS * G) + UAVENV_TAPE_ZP * G + gl] : 0.0f;
    Words4 w = noise_words(c.seed, env_index, episode, step, (uint32_t)gl, 5);
    float zP, spare;
    normal_pair(w.w0, w.w1, zP, spare);
    return zP;
}

// greedy_agents.py:42-67 GreedyAgent._move_toward (float32 position arithmetic; group-uniform inputs)
__device__ __forceinline__ int policy_move_toward(float ux, float uy, int gw, int gh, float tx, float ty) {
    const float dx = tx - ux, dy = ty - uy;
    const bool here = (fabsf(dx) <= 0.5f) & (fabsf(dy) <= 0.5f);
    const bool horiz = fabsf(dx) > fabsf(dy);
    const float nx = horiz ? ux + (dx > 0 ? 1.0f : -1.0f) : ux;
    const float ny = horiz ? uy : uy + (dy > 0 ? 1.0f : -1.0f);
    const bool out = (nx < 0) | (nx >= (float)gw) | (ny < 0) | (ny >= (float)gh);
    const float mdx = nx - ux, mdy = ny - uy;
    const int mv = mdx > 0 ? 3 : (mdx < 0 ? 2 : (mdy > 0 ? 0 : (mdy < 0 ? 1 : 4)));
    return (here | out) ? 4 : mv;
}

// Heuristic policies evaluated on device for the CURRENT state of one environment group:
//   UAVENV_POLICY_NEAREST            greedy_agents.py:73-100   collect if any sensor with data is in range, else
//                                                             step toward the nearest sensor with data
//   UAVENV_POLICY_MAX_THROUGHPUT_V2  greedy_agents.py:105-216  collect if an in-range sensor with data has an
//                                                             acceptable SF, else step toward the best-scored one
// Python's min()/`score > best` keep the FIRST extremum: the lowest lane among equal keys.
template <int G, typename CT>
__device__ __forceinline__ int policy_action(const CT& c, const Sensor& s, const Env& e, bool act, int policy, float zP) {
    const int gl = group_lane<G>();
    const double det = rssi_deterministic(c, e.ux, e.uy, s.sx, s.sy);
    const bool has = act & (s.b > 0);
    const bool in_range = has & ((det + c.sigma * (double)zP) >= c.thr);          // iot_sensors.py:214-219
    const float dx = s.sx - e.ux, dy = s.sy - e.uy;
    const float dist = sqrt_rn(dx * dx + dy * dy);                                // np.linalg.norm, float32
    bool collect;
    uint64_t pick;
    if (policy == UAVENV_POLICY_NEAREST) {
        collect = gany<G>(in_range);
        const float dm = gmin_f32<G>(has ? dist : __builtin_inff());
        pick = gballot<G>(has & (dist == dm));
    } else {
        const double battery_pct = e.battery / 274.0;                              // greedy_agents.py:133 (hard-coded)
        const int steps_left = c.max_steps - e.step;
        const double r = (double)steps_left / (double)c.max_steps;
        const double steps_ratio = r < 1.0 ? r : 1.0;
        const int sf_thr = ((battery_pct > 0.5) & (steps_ratio > 0.5)) ? 9 : (((battery_pct > 0.2) & (steps_ratio > 0.2)) ? 10 : 12);
        const int sf = (int)(s.flags & kSfMask);
        collect = gany<G>(in_range & (sf <= sf_thr));
        const double sf_w = ((battery_pct < 0.1) | (steps_left < 50)) ? 1.0 : (((battery_pct < 0.3) | (steps_left < 150)) ? 2.0 : 5.0);
        const int pr = 13 - sf > 0 ? 13 - sf : 0;
        const double sf_score = pr * 5.0 * sf_w;
        const double buffer_score = (s.b / c.bmax) * 10.0;
        const double duty_score = c.p_cycle * 2.0;                                // (duty_cycle / 100) * 2
        const float distance_penalty = ((dist / (float)e.gw) * 5.0f) * 1.0f;
        const double score = ((sf_score + buffer_score) + duty_score) - (double)distance_penalty;
        const double sm = gmax<G>(has ? score : -__builtin_inf());
        pick = gballot<G>(has & (score == sm));
    }
    const int target = pick ? (__ffsll((long long)pick) - 1) : 0;
    const float tx = gshfl<G>(s.sx, target), ty = gshfl<G>(s.sy, target);
    const int mv = policy_move_toward(e.ux, e.uy, e.gw, e.gh, tx, ty);
    (void)gl;
    return (collect | (pick == 0ull)) ? 4 : mv;
}

template <int G, bool kLean, typename CT, typename P>
__device__ __forceinline__ void draw_step_noise(const CT& c, const P& p, uint32_t env_index, uint32_t episode,
                                                size_t env, bool in_batch, uint32_t step, bool need_collect, StepNoise& z) {
    const int gl = group_lane<G>();
    z.c2 = z.c3 = 0u; z.zc_ready = true; z.w3 = 0u; z.have_w3 = false;
    if (UAV_TAPE(p.step_tape) != nullptr) {                      // kernel-uniform
        z.zA = z.zB = z.zC = z.zD = z.zE = 0.f; z.u = 1.f;
        if (in_batch) {                                // the tape has no rows for the padding environments
            const float* t = p.step_tape + env * (size_t)(UAVENV_TAPE_STEP_SLOTS * G) + gl;
            z.zA = t[UAVENV_TAPE_ZA * G]; z.zB = t[UAVENV_TAPE_ZB * G]; z.u = t[UAVENV_TAPE_U * G];
            z.zC = t[UAVENV_TAPE_ZC * G]; z.zD = t[UAVENV_TAPE_ZD * G]; z.zE = t[UAVENV_TAPE_ZE * G];
        }
        return;
    }
    Words4 w = noise_words(c.seed, env_index, episode, step, (uint32_t)gl, 0);
    normal_pair(w.w0, w.w1, z.zD, z.zE);
    z.u = u24(w.w2);
    z.w3 = w.w3; z.have_w3 = true;             // lane 0's spare word is the uniform-random policy's NEXT action
    z.zA = z.zB = z.zC = 0.f;
#ifdef UAV_ABL_NOCNOISE      // timing-only ablation build (tools/ablate.py)
    if (need_collect) { z.zA = z.zE; z.zB = z.zD; z.c2 = w.w2; z.c3 = w.w3; z.zc_ready = false; }
    need_collect = false;
#endif
    if (need_collect) {                       // wave-uniform
        Words4 v = noise_words(c.seed, env_index, episode, step, (uint32_t)gl, 1);
        normal_pair(v.w0, v.w1, z.zA, z.zB);
        z.c2 = v.w2; z.c3 = v.w3; z.zc_ready = false;
    }
}
__device__ __forceinline__ void finish_zc(StepNoise& z) {   // call under wave-uniform control flow
    if (!z.zc_ready) { float spare; normal_pair(z.c2, z.c3, z.zC, spare); z.zc_ready = true; }
}

// uav_env.py:400-427 reset (+ iot_sensors.py:305-321, uav.py:241-258; DomainRandEnv.reset
// dqn.py:301-373 under the flags) for the groups with `rs` set.  Leaves the new episode's sensor
// registers in `s`, the record in `r`, and returns (zD, zE) of the reset observation.
template <int G, bool kLean, typename CT, typename P>
__device__ __forceinline__ void reset_group(const CT& c, const P& p, Sensor& s, UavEnvRecord& r, size_t env,
                                            bool in_batch, bool rs, bool draw_layout, float& zD, float& zE,
                                            uint32_t& w3, bool& have_w3) {
    const int gl = group_lane<G>();
    if (!rs) return;                                   // group-uniform; no cross-lane ops skipped below
    r.episode += 1u;
    const uint32_t ep = r.episode;
    Words4 w = noise_words(c.seed, r.env_index, ep, 0u, (uint32_t)gl, 2);
    if ((UAV_FLAGS(c) & UAVENV_FLAG_RANDOM_LAYOUT) && c.n_grid_choices > 0) {          // dqn.py:334
        Words4 w0 = noise_words(c.seed, r.env_index, ep, 0u, 0u, 2);
        int g = (int)(((uint64_t)w0.w3 * (uint32_t)c.n_grid_choices) >> 32);
        int gw = c.gw[0], gh = c.gh[0];
#pragma unroll
        for (int k = 1; k < 8; k++) { gw = (g == k) ? c.gw[k] : gw; gh = (g == k) ? c.gh[k] : gh; }
        r.grid_w = gw; r.grid_h = gh;
        r.inv_grid_w = 1.0 / (double)gw; r.inv_grid_h = 1.0 / (double)gh;
    }
    float fill_u = u24(w.w0);
    float zS = 0.f;
    if (UAV_TAPE(p.reset_tape) != nullptr) {
        fill_u = 0.f; zD = 0.f; zE = 0.f;
        if (in_batch) {
            const float* t = p.reset_tape + env * (size_t)(UAVENV_RTAPE_SLOTS * G) + gl;
            fill_u = t[UAVENV_RTAPE_FILL * G]; zD = t[UAVENV_RTAPE_ZD * G]; zE = t[UAVENV_RTAPE_ZE * G];
            zS = t[UAVENV_RTAPE_ZS * G - gl];                                      // element 0 of the row
        }
    } else {
        Words4 v = noise_words(c.seed, r.env_index, ep, 0u, (uint32_t)gl, 0);
        normal_pair(v.w0, v.w1, zD, zE);
        w3 = v.w3; have_w3 = true;
    }
    // dqn.py:340-351: the fresh sensors of a DomainRandEnv episode inherit `s0.spreading_factor` -- the SF the OLD sensor 0
    // got from the discarded super().reset(): back to SF 12 without an EMA sample (iot_sensors.py:309-311), then one
    // update_spreading_factor in the reset observation (uav_env.py:427 -> :654) with the UAV at the PREVIOUS episode's
    // start (uav.py:256): avg = cur, SF by the threshold list, 12 kept when none fires (iot_sensors.py:239-255).
    // Evaluated here, while `s` / `r` still hold the old layout and start; its shadowing sample zS is lane 0 of Philox call 6.
    uint32_t fresh_sf = 12u;
    if (UAV_FLAGS(c) & UAVENV_FLAG_RANDOM_LAYOUT) {
        if (UAV_TAPE(p.reset_tape) == nullptr) {
            const Words4 q = noise_words(c.seed, r.env_index, ep, 0u, 0u, 6);
            float spare;
            normal_pair(q.w0, q.w1, zS, spare);
        }
        const float sx0 = gshfl<G>(s.sx, 0), sy0 = gshfl<G>(s.sy, 0);
        const double avg0 = rssi_deterministic(c, r.start_x, r.start_y, sx0, sy0) + c.sigma * (double)zS;
        fresh_sf = avg0 > c.sf_thr[2] ? 11u : fresh_sf;
        fresh_sf = avg0 > c.sf_thr[1] ? 9u : fresh_sf;
        fresh_sf = avg0 > c.sf_thr[0] ? 7u : fresh_sf;
    }
    if (draw_layout) {                                                             // uav_env.py:366-374 / dqn.py:342-344
        s.sx = u24(w.w1) * (float)r.grid_w;
        s.sy = u24(w.w2) * (float)r.grid_h;
    }
    double fill = c.fill_lo + c.fill_span * (double)fill_u;                        // uav_env.py:410
    double clipped = fill < 0.0 ? 0.0 : (fill > 1.0 ? 1.0 : fill);
    s.b = c.bmax * clipped;                                                        // iot_sensors.py:308
    s.gen = s.b;                                                                   // :316
    s.tx = 0.0; s.lost = 0.0; s.avg = 0.0;                                         // :311,317-318
    s.flags = (s.flags & kDataCollected) | 12u;                                    // :309 SF12; visited cleared (uav_env.py:416)
    if (UAV_FLAGS(c) & UAVENV_FLAG_RANDOM_LAYOUT) { s.b = 0.0; s.gen = 0.0; s.flags = fresh_sf; }   // dqn.py:346-360 fresh sensors
    r.battery = c.maxb;                                                            // uav.py:257
    r.current_step = 0; r.total_reward = 0.0; r.total_data_collected = 0.0;        // uav_env.py:413-415
    r.last_step_bytes = 0.0; r.capture_triggers = 0; r.boundary_hits = 0;          // :419-422
    r.edge_steps = 0; r.collisions_total = 0; r.first_full_coverage_step = -1;
    r.episode_return = 0.0;
}

// dqn.py:375-403 _sample_far_start: rejection-sample a start >= min_start_dist from every sensor,
// falling back to the furthest candidate.  Candidates from Philox call 4 (lane field = try).
template <int G, typename CT>
__device__ __forceinline__ void far_start(const CT& c, const Sensor& s, UavEnvRecord& r, bool act, bool rs) {
    double W = (double)r.grid_w, H = (double)r.grid_h;
    float best_x = r.start_x, best_y = r.start_y, best_d = -1.0f;
    bool searching = rs;
    for (int t = 0; t < c.max_tries; t++) {
        if (!__any(searching)) break;
        Words4 w = noise_words(c.seed, r.env_index, r.episode, 0u, (uint32_t)t, 4);
        float cx = (float)(0.05 * W + (0.95 * W - 0.05 * W) * (double)u24(w.w0));
        float cy = (float)(0.05 * H + (0.95 * H - 0.05 * H) * (double)u24(w.w1));
        float dx = cx - s.sx, dy = cy - s.sy;
        float d = sqrt_rn(dx * dx + dy * dy);
        float dmin = gmin_f32<G>(act ? d : __builtin_inff());
        if (searching) {
            if (dmin > best_d) { best_d = dmin; best_x = cx; best_y = cy; }
            if ((double)dmin >= c.min_start_dist) { best_x = cx; best_y = cy; searching = false; }
        }
    }
    if (rs) { r.start_x = best_x; r.start_y = best_y; }
}

// ---------------------------------------------------------------------------------------------
// init kernel: records + (optionally) Philox sensor layouts, no episode started
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kSmallBlockThreads) void uav_init_kernel(const Consts* cptr, Ptrs p, uint32_t env_index_base,
                                                                 int32_t grid_w, int32_t grid_h, int32_t n,
                                                                 float start_x, float start_y) {
    UAV_CONSTS(cptr);
    const int gl = group_lane<G>();
    const uint32_t env = blockIdx.x * (kSmallBlockThreads / G) + threadIdx.x / G;
    const uint32_t idx = env * G + gl;
    uint32_t gidx = env_index_base + env;
    Words4 w = noise_words(c.seed, gidx, 0xFFFFFFFFu, 0u, (uint32_t)gl, 2);
    Sensor s;
    s.sx = u24(w.w1) * (float)grid_w;
    s.sy = u24(w.w2) * (float)grid_h;
    s.b = 0.0; s.gen = 0.0; s.tx = 0.0; s.lost = 0.0; s.avg = 0.0; s.flags = 12u;
    store_sensor<G>(p, idx, s, true);
    if (gl == 0) {
        UavEnvRecord r;
        r.battery = c.maxb; r.total_reward = 0.0; r.total_data_collected = 0.0; r.last_step_bytes = 0.0;
        r.prev_dist_nearest = 0.0; r.episode_return = 0.0;
        r.uav_x = start_x; r.uav_y = start_y; r.start_x = start_x; r.start_y = start_y;
        r.current_step = 0; r.episode = 0xFFFFFFFFu;
        r.capture_triggers = 0; r.boundary_hits = 0; r.edge_steps = 0; r.collisions_total = 0;
        r.first_full_coverage_step = -1;
        r.grid_w = grid_w; r.grid_h = grid_h; r.num_sensors = n;
        r.env_index = gidx; r.status = 0u;
        r.inv_grid_w = 1.0 / (double)grid_w; r.inv_grid_h = 1.0 / (double)grid_h;
        p.rec[env] = r;
        UavEnvEpisodeStats st = {};
        p.stats[env] = st;
    }
}

// ---------------------------------------------------------------------------------------------
// reset kernel
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kSmallBlockThreads) void uav_reset_kernel(const Consts* cptr, Ptrs p, ResetArgs a) {
    constexpr bool kLean = false;
    UAV_CONSTS(cptr);
    const int gl = group_lane<G>();
    const uint32_t env = blockIdx.x * (kSmallBlockThreads / G) + threadIdx.x / G;
    const uint32_t idx = env * G + gl;

    UavEnvRecord r = p.rec[env];
    Sensor s;
    load_sensor<G>(p, idx, s);
    const bool in_batch = env < (uint32_t)a.num_envs;
    bool rs = true;
    if (a.mask != nullptr) rs = in_batch && a.mask[env] != 0;
    const bool act = gl < r.num_sensors;

    float zD = 0.f, zE = 0.f;
    const bool draw_layout = (UAV_FLAGS(c) & UAVENV_FLAG_RANDOM_LAYOUT) != 0;
    uint32_t rw3 = 0u; bool have_rw3 = false;
    reset_group<G, kLean>(c, p, s, r, env, in_batch, rs, draw_layout, zD, zE, rw3, have_rw3);
    if (UAV_FLAGS(c) & UAVENV_FLAG_FAR_START) far_start<G>(c, s, r, act, rs);
    if (rs) { r.uav_x = r.start_x; r.uav_y = r.start_y; }                         // uav.py:256, dqn.py:364-365
    double det = rssi_deterministic(c, r.uav_x, r.uav_y, s.sx, s.sy);
    float* dst = (in_batch && a.obs != nullptr) ? a.obs + env * (size_t)c.obs_dim : nullptr;
    observe<G, kLean>(c, s, r.num_sensors, r.grid_w, r.grid_h, r.inv_grid_w, r.inv_grid_h, r.uav_x, r.uav_y, r.battery, act, rs, det, zD, zE, dst);   // uav_env.py:427
    if (UAV_FLAGS(c) & UAVENV_FLAG_PROX_SHAPING) {
        double d0 = dist_nearest_with_data<G>(s, act, r.uav_x, r.uav_y);           // dqn.py:368
        if (rs) r.prev_dist_nearest = d0;
    }
    if (rs) {
        store_sensor<G>(p, idx, s, draw_layout);
        if (gl == 0) p.rec[env] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// One environment step for one lane group: the hot path.  Shared by the single-step kernel (record read from
// global memory, through the scalar cache when G = 64) and the fused rollout kernel (kRegs: record and sensor
// state kept in registers across steps, `rec` / `rec_out` point at the kernel's private copy).
// ---------------------------------------------------------------------------------------------
// The uniform-random policy's action of step s is word 3 of the Philox call that lane 0 makes ANYWAY for the observation
// noise of step s-1 (call 0; for s = 1 the call of the reset observation): the policy costs no generator call of its own.
__device__ __forceinline__ int random_action(uint32_t w3) { return (int)(((uint64_t)w3 * 5u) >> 32); }

// Scheduling / action word the random-policy step leaves for the next launch: bits 0-2 the action of (episode, step),
// bit 3 "valid", bits 4-19 step, bits 20-31 episode (low bits).  A launch uses the action of a word whose tag matches the
// record it finds (else it draws the action itself: the word is an optimisation, never a source of truth), and every
// launch uses "(word & 7) == 4" to spread the collect steps over the SIMDs.  0 = no information.
__device__ __forceinline__ uint32_t hint_tag(uint32_t episode, uint32_t step) { return 8u | ((step & 0xFFFFu) << 4) | (episode << 20); }

// Timing-only probes of DESIGN.md section 8's "wave-uniform work of the 16 environments of a workgroup packed into one wave" (tools/exp.sh):
//   -DUAV_EXP_UNIFORM_SKIP  only the first wavefront of a workgroup executes the two largest wave-uniform blocks (the move and the
//                           record epilogue; the others keep stale values): an UPPER bound of what packing them could save;
//   -DUAV_EXP_BARRIERS      two workgroup barriers where the packed form would exchange through LDS: its COST side.
// Neither build computes a correct step.
#ifdef UAV_EXP_UNIFORM_SKIP
#define UAV_UNIFORM_LEADER (G != 64 || (threadIdx.x >> 6) == 0)
#else
#define UAV_UNIFORM_LEADER true
#endif
#ifdef UAV_EXP_BARRIERS
#define UAV_EXP_BARRIER() __builtin_amdgcn_s_barrier()
#else
#define UAV_EXP_BARRIER() do { } while (0)
#endif

template <int G, bool kLean, bool kRegs = false, typename RecPtr = UavEnvRecord*, typename CT = Consts, typename P = Ptrs, typename A = StepArgs>
__device__ __forceinline__ void step_once(const CT& c, const P& p, const A& a, uint32_t env, bool in_batch,
                                          RecPtr rec, UavEnvRecord* rec_out, Sensor& s, bool& wrote_pos, bool& live, uint32_t& dirty, uint32_t& status_or,
                                          int& action_out, uint32_t& next_word, int& wt_out, uint32_t hint_word = 0u, size_t row_offset = 0) {
    const int gl = group_lane<G>();
    // outputs of fused rollouts are [K][E][...] blocks: row = k * E + env (k = 0 for the single-step kernel)
    const size_t out = row_offset + env;
    wt_out = 0;
    UAV_PHASE(0);
    UAV_EXIT_AT(0, env);
    Env e = load_env<G>(rec);
    const int n = e.n;
    const bool act = gl < n;

    // ---- action -----------------------------------------------------------------------------
    const uint32_t step = (uint32_t)(e.step + 1);
    int action;
    if (UAV_POLICY(a) >= UAVENV_POLICY_NEAREST) {                    // heuristic baseline evaluated on device
        const float zP = draw_policy_noise<G, kLean>(c, p, e.env_index, e.episode, env, in_batch, step);
        action = uni<G>(policy_action<G>(c, s, e, act, a.policy, zP));
    } else if (UAV_POLICY(a) == UAVENV_POLICY_ACTIONS) action = uni<G>(in_batch ? a.actions[out] : 0);
    else {                                                       // uniform-random policy (Philox call 3)
#ifdef UAV_ABL_CHEAPACTION   // timing-only ablation build: no Philox on the scalar unit for the action / the hint
        action = (int)((e.env_index * 7u + step * 3u + e.episode) % 5u);
        (void)hint_word;
#else
        const bool word_ok = (hint_word & ~7u) == hint_tag(e.episode, step);       // handed over by the last launch / step
        if (G == 64) {                                                            // wave-uniform
            if (word_ok) action = (int)(hint_word & 7u);
            else action = random_action(noise_words(c.seed, e.env_index, e.episode, step - 1u, 0u, 0).w3);   // draw it
        } else {
            int drawn = 0;
            if (__any(!word_ok)) drawn = random_action(noise_words(c.seed, e.env_index, e.episode, step - 1u, 0u, 0).w3);
            action = word_ok ? (int)(hint_word & 7u) : drawn;
        }
#endif
    }
    const bool is_c = action == 4;
    const bool is_m = (action >= 0) & (action <= 3);
    const uint32_t status_bits = (!is_c & !is_m) ? 1u : 0u;     // uav_env.py:468 ValueError (after ageing)
    UAV_PHASE(1);
    UAV_EXIT_AT(1, action + e.step + (int)e.episode + (int)e.ux);

    // ---- uav_env.py:439-447: step counter, edge-cell bookkeeping on the PRE-move position --------
    e.step += 1;
    int edge_inc;
    {
        double W = (double)e.gw, H = (double)e.gh, ux = (double)e.ux, uy = (double)e.uy;
        const double eps = 1e-6;
        const bool edge = (ux <= eps) | (uy <= eps) | (ux >= W - 1 - eps) | (uy >= H - 1 - eps);
        edge_inc = edge ? 1 : 0;
    }
    // ---- :450-459 age all sensors (iot_sensors.py:114-125), data-loss delta ------------------------
    const double step_duration = is_c ? c.coll_dur : 1.0;
    double loss = 0.0;
    {
        // Selects replaced by arithmetic that gives the same bits: lanes beyond the sensor count generate nothing (their rows are
        // 0 and stay 0: 0 + 0, min(0, Bmax), max(-Bmax, 0) = 0); the overflow is max(b' - Bmax, 0) -- positive exactly when
        // b' > Bmax, and `lost + 0.0` is `lost` -- and the capped buffer min(b', Bmax).
        const double new_data = act ? c.rate * step_duration : 0.0;
        const double potential = s.b + new_data;
        const double l = __builtin_fmax(potential - c.bmax, 0.0);
        s.gen = s.gen + new_data;
        s.b = __builtin_fmin(potential, c.bmax);
        s.lost = s.lost + l;
        dirty |= (l > 0.0) ? 2u : 0u;
        loss = l;
    }
    const double step_data_loss = gsum<G>(loss);
    // Every loaded row is "used" here, while only the loads are in flight: the compiler then never has to cover a load result it
    // cannot prove consumed with an `s_waitcnt vmcnt(0)` between the final stores (where it would wait for the stores as well).
    asm volatile("" :: "v"(s.avg), "v"(s.tx), "v"(s.flags), "v"(s.sx), "v"(s.sy));
    UAV_PHASE(2);
    UAV_EXIT_AT(2, __double2loint(step_data_loss) + __double2loint(s.tx + s.avg) + (int)s.flags + (int)s.sx + (int)s.sy);

    double reward = 0.0;
    double bytes_step = 0.0;          // uav_env.py:607 last_step_bytes_collected
    int captures = 0, collisions = 0, bh_inc = 0;

    // ---- :494-516 move (uav.py:127-185, reward_function.py:69-79); scalar per group ---------------
    // Written branch-free (bitwise &, selects) on purpose: ROCm 7.2's gfx950 backend mis-compiled the
    // natural `ok = a && b && c && d; if (ok) {..} else {..}` form here (the e_move / r_move selects were
    // sunk into the `0 <= nx` block only; caught by the parity tests).
    if (UAV_UNIFORM_LEADER) {
        const double battery_before = e.battery;
        const float dxm = action == 2 ? -1.0f : (action == 3 ? 1.0f : 0.0f);
        const float dym = action == 0 ? 1.0f : (action == 1 ? -1.0f : 0.0f);
        const float nx = e.ux + dxm, ny = e.uy + dym;
        const bool ok = (0.0f <= nx) & (nx < (float)e.gw) & (0.0f <= ny) & (ny < (float)e.gh);
        const bool mv = is_m & ok;
        e.ux = mv ? nx : e.ux;
        e.uy = mv ? ny : e.uy;
        const double drain = ok ? c.e_move : c.e_coll;
        const double battery_after = battery_before - drain;
        const double battery_used = battery_before - battery_after;
        double rw = c.p_step;
        rw += ok ? c.r_move : c.p_boundary;
        rw += c.p_battery * battery_used;
        rw += c.p_loss * step_data_loss;
        e.battery = is_m ? battery_after : e.battery;
        bh_inc = (is_m & !ok) ? 1 : 0;
        reward = is_m ? rw : 0.0;
    }
    UAV_EXP_BARRIER();

    // One deterministic path-loss evaluation per sensor and step: a collect step does not move the
    // UAV, so the five RSSI samples of a step (zA,zB,zC at the pre-action position, zD,zE at the
    // post-action position) all share it.
    const double det = rssi_deterministic(c, e.ux, e.uy, s.sx, s.sy);
    UAV_PHASE(3);
    UAV_EXIT_AT(3, (uint32_t)__ballot(det > -80.0) + __double2loint(reward));

#ifdef UAV_ABL_NOCOLLECT     // timing-only ablation build
    const bool any_c = false;
#else
    const bool any_c = __any(is_c) != 0;
#endif
    StepNoise z;
    draw_step_noise<G, kLean>(c, p, e.env_index, e.episode, env, in_batch, step, any_c, z);
    UAV_PHASE(4);
    UAV_EXIT_AT(4, (uint32_t)__ballot(det + z.zD > -80.0) + (uint32_t)__ballot(z.zE + z.u + z.zA > 0.f) + __double2loint(reward));

    // ---- :518-632 collect with Capture-Effect collision handling ---------------------------------
    if (any_c) {
        const bool actc = act & is_c;
        // P0 :526 float32 AoI urgency  b / rate
        const float urg_before = (actc & (c.rate > 0)) ? (float)div_const(s.b, c.rate, c.inv_rate) : 0.0f;
        e.battery = is_c ? e.battery - c.e_hover : e.battery;                  // P1 :529
        // P2 :535-551
        const bool has = actc & (s.b > 0);
        double cur = 0.0;
        {
            const double cand = det + c.sigma * (double)z.zA;
            double avg2 = s.avg; uint32_t fl2 = s.flags;
            adr_update(c, cand, avg2, fl2);                                     // :539
            cur = has ? cand : 0.0;
            s.avg = has ? avg2 : s.avg;
            s.flags = has ? fl2 : s.flags;
        }
        const uint32_t sf = s.flags & kSfMask;
        bool attempt;
        {
            // :543 get_success_probability(advanced): p = 0 below the threshold, else the SNR sigmoid.
            // The sigmoid is within 2.1e-9 of 1 for x > 20, so `p * p_cycle > u` is decided without the
            // exponential unless u falls into that 2e-9-wide band (then the exact expression is used).
            const double rssi = det + c.sigma * (double)z.zB;
            const double x = (rssi - c.noise_floor) - sf_required_snr(sf);
            const double u = (double)z.u;
            const bool in_rng = !(rssi < c.thr);
            const bool sure_yes = (x > 20.0) & ((c.p_cycle * (1.0 - 2.2e-9)) > u);
            const bool sure_no = !(c.p_cycle > u);
            // (default constants: in range means rssi >= -85 = noise floor + 20 and every required SNR is <= -6, so x >= 26 and the
            // exponential can never be needed; the generic kernels keep the exact path for other thresholds)
            const bool need_exp = kIsDefaultConsts<CT> ? false : (has & in_rng & !sure_yes & !sure_no);
            bool att = sure_yes;
            if (__any(need_exp)) {
                const double p_link = sigmoid_rare(x);
                att = need_exp ? ((p_link * c.p_cycle) > u) : att;
            }
            attempt = has & in_rng & att;                                       // :549
        }
        // P3 :554-572: per SF class, sole attempter wins; else top wins iff > second + 6 dB.
        // Each attempter scans the (few) other attempters of its group.
        uint64_t m = gballot<G>(attempt);
        int others = 0;
        double omax = -__builtin_inf();
        bool lower_same = false, beaten = false;
#ifdef UAV_ABL_NOCAPTURE      // timing-only ablation build
        m = 0ull;
#endif
        while (__any(m != 0ull)) {
            int j = m ? (__ffsll((long long)m) - 1) : 0;
            double cj = gbcast<G>(cur, j);
            uint32_t sfj = gbcast<G>(sf, j);
            const bool hit = (m != 0ull) & attempt & (sfj == sf) & (j != gl);
            others += hit ? 1 : 0;
            omax = (hit & (cj > omax)) ? cj : omax;
            lower_same |= hit & (j < gl);
            beaten |= hit & ((cj > cur) | ((cj == cur) & (j < gl)));
            m &= (m - 1ull);
        }
        const bool contested = attempt & (others > 0);
        const bool winner = attempt & ((others == 0) | (!beaten & (cur > (omax + c.cap_thr))));
        const int collision_count = __popcll(gballot<G>(contested)) - __popcll(gballot<G>(contested & !lower_same));
        const int capt = __popcll(gballot<G>(winner & contested));
        // P4 :575-594 + iot_sensors.py:127-145 collect_data (range check with a fresh sample zC: winners only)
        double bytes = 0.0;
        bool got = false;
        if (__any(winner)) {
            finish_zc(z);
            const double rssi = det + c.sigma * (double)z.zC;
            const bool take = winner & !(rssi < c.thr) & (s.b > 0);
            const double max_collectible = sf_data_rate(sf) * c.coll_dur;
            const double by = s.b < max_collectible ? s.b : max_collectible;
            bytes = take ? by : 0.0;
            s.b = take ? s.b - by : s.b;
            s.tx = take ? s.tx + by : s.tx;
            dirty |= take ? 1u : 0u;
            got = take & (by > 0);
            s.flags |= got ? kDataCollected : 0u;
        }
        const uint64_t winner_mask = gballot<G>(winner);
        const double total_bytes = gsum_sparse<G>(bytes, winner_mask);          // bytes is 0 off the winners
        const bool any_new = gany<G>(got & !(s.flags & kVisited));
        s.flags |= got ? kVisited : 0u;
        const int nw = __popcll(winner_mask);
        const bool attempted_empty = gany<G>(actc & (s.b <= 0));                // :596
        const bool all_collected = !gany<G>(actc & (s.b > 0));                  // :604
        // P5 :599-602 (float32); only drained sensors changed their AoI urgency
        float diff = 0.0f;
        if (__any(got)) {
            const float urg_after = (actc & (c.rate > 0)) ? (float)div_const(s.b, c.rate, c.inv_rate) : 0.0f;
            const float d = urg_before - urg_after;
            diff = (got & (d > 0.0f)) ? d : 0.0f;
        }
        const double urgency_reduced = (double)np_sum_f32<G>(diff, n);
        // P6 :607-630
        double mean_urgency = 0.0;
        if (__any(winner)) {
            const double ui = winner ? calc_urgency(c, s.b, s.gen, s.lost) : 0.0;
            const double su = gsum_sparse<G>(ui, winner_mask);
            mean_urgency = nw > 0 ? div_const(su, (double)nw, c.inv_small[nw]) : 0.0;
        }
        // reward_function.py:46-57 variance "starvation" penalty (np.var: two-pass, ddof 0)
        double starvation = 0.0;
#ifndef UAV_ABL_NOVAR           // timing-only ablation build
        {
            // a buffer never exceeds its cap, so the maximum is the cap itself as soon as one sensor is full
            // (the steady state); only otherwise is the max-reduction needed
            double mx = c.bmax;
            if (G != 64 || !gany<G>(actc & (s.b == c.bmax))) mx = gmax<G>(actc ? s.b : -__builtin_inf());   // wave-uniform branch
            const bool use = (n > 1) & (mx != 0);
            const double nb = (actc & use) ? s.b / mx : 0.0;
            const double inv_n = c.inv_small[n];
            const double mean = div_const(gsum<G>(nb), (double)n, inv_n);
            const double dv = (actc & use) ? (nb - mean) * (nb - mean) : 0.0;
            const double var = div_const(gsum<G>(dv), (double)n, inv_n);
            starvation = use ? c.p_starvation * var : 0.0;
        }
#endif
        {
            double rw = c.p_step + c.p_hover;                                   // reward_function.py:97
            {
                const double r1 = rw + c.r_byte * total_bytes * mean_urgency;  // :100-103
                const double r2 = any_new ? r1 + c.r_new : r1;
                rw = (total_bytes > 0) ? r2 : rw;
            }
            rw = (urgency_reduced > 0) ? rw + c.r_urg * urgency_reduced : rw;
            rw = (attempted_empty & (total_bytes == 0)) ? rw + c.p_revisit : rw;
            rw += c.p_battery * c.used_hover;
            rw = (collision_count > 0) ? rw + c.p_collision * collision_count : rw;
            rw = (step_data_loss > 0) ? rw + c.p_loss * step_data_loss : rw;
            rw += starvation;
            rw = all_collected ? rw + c.r_done : rw;
            reward = is_c ? rw : reward;
            bytes_step = is_c ? total_bytes : 0.0;
            captures = is_c ? capt : 0;
            collisions = is_c ? collision_count : 0;
        }
    }

    // ---- :471-487 truncation + terminal penalties (reward_function.py:59-67) -------------------
    const bool truncated = !(e.battery > c.alive_level) | (e.step >= c.max_steps);   // uav.py:224, uav_env.py:477
    const int visited_cnt = __popcll(gballot<G>(act & ((s.flags & kVisited) != 0u)));
    if (__any(truncated)) {
        const bool starved = act & (s.gen > 0) & ((s.tx / s.gen) < c.cr_thr);
        const int starved_cnt = __popcll(gballot<G>(starved));
        const int unvisited = n - visited_cnt;
        double pen = (unvisited > 0) ? c.p_unvisited * unvisited : 0.0;
        double rt = reward + pen;
        rt += c.p_starved * starved_cnt;
        reward = truncated ? rt : reward;
    }
    const double reward_unshaped = reward;                                        // :487 total_reward += reward
    UAV_PHASE(5);
    UAV_EXIT_AT(5, (uint32_t)__ballot(det + z.zD > -80.0) + (uint32_t)__ballot(z.zE + (float)s.b > 0.f) + __double2loint(reward) + visited_cnt);

    // ---- observation of the stepped state (side effect: ADR EMA) -------------------------------
    const OutArgs o = load_out_args(a);                 // every output pointer of the step: one wide scalar load
    wt_out = o.write_through;
    const bool auto_reset = (UAV_FLAGS(c) & UAVENV_FLAG_AUTO_RESET) != 0;
    const bool do_reset = truncated & auto_reset;
    int term_ticket = -1;         // value of the terminal-pool counter when this step claimed its row (row = ticket mod rows)
    {
        float* dst = nullptr;
        if (in_batch) {
            if (do_reset) dst = o.term_obs ? o.term_obs + out * (size_t)c.obs_dim : nullptr;
            else dst = o.obs ? o.obs + out * (size_t)c.obs_dim : nullptr;
        }
        if (o.term_pool != nullptr) {          // kernel-uniform: terminal rows go to a compact pool instead
            int row = -1;
            if (__any(do_reset & in_batch)) {
                uint32_t t = 0u;
                if (do_reset & in_batch & (gl == 0)) t = atomicAdd(o.term_counter, 1u);
                t = gshfl<G>(t, 0);
                if (do_reset & in_batch) {
                    term_ticket = (int)(t & 0x7FFFFFFFu);
                    row = (int)(t % (uint32_t)o.term_rows);
                    dst = o.term_pool + (size_t)row * (size_t)c.obs_dim;
                }
            }
            if (in_batch && gl == 0 && o.term_index != nullptr) o.term_index[out] = row;
        }
        observe<G, kLean>(c, s, n, e.gw, e.gh, e.inv_w, e.inv_h, e.ux, e.uy, e.battery, act, true, det, z.zD, z.zE, dst,
                          o.write_through != 0);   // :488
    }

    UAV_EXIT_AT(6, (uint32_t)__ballot(s.avg > -80.0) + __double2loint(reward) + visited_cnt + (int)s.flags);
    // ---- epilogue: the cold part of the record (read late on purpose: see struct Env) -----------------
    asm volatile("" ::: "memory");
#ifdef UAV_ABL_NORELOAD       // timing-only ablation build
    UavEnvRecord r = {};
#else
    UavEnvRecord r = load_record(rec);
#endif
    UAV_PHASE(6);
    if (G == 64 && kRegs) {   // wave-uniform: keep the record in SGPRs (the single-step kernel reads it with scalar loads)
        union { UavEnvRecord r; int w[32]; } u;
        u.r = r;
        // words 0-11 are the six float64 totals (battery ... episode_return): only vector instructions touch them, as
        // SGPR values they cost a lane read per word and step and push other scalars into spills (4.93 -> 4.86 us per step)
#pragma unroll
        for (int i = 12; i < 32; i++) u.w[i] = __builtin_amdgcn_readfirstlane(u.w[i]);
        r = u.r;
    }
    UAV_EXP_BARRIER();
    if (UAV_UNIFORM_LEADER) {
    r.battery = e.battery; r.uav_x = e.ux; r.uav_y = e.uy; r.current_step = e.step;
    r.episode = e.episode; r.env_index = e.env_index; r.num_sensors = e.n; r.grid_w = e.gw; r.grid_h = e.gh;
    r.inv_grid_w = e.inv_w; r.inv_grid_h = e.inv_h;
    r.status |= status_bits;
    r.edge_steps += edge_inc;
    r.boundary_hits += bh_inc;
    r.total_reward += reward_unshaped;
    r.total_data_collected += bytes_step;
    r.last_step_bytes = (is_m | is_c) ? bytes_step : r.last_step_bytes;           // :463, :607
    r.capture_triggers += captures;
    r.collisions_total += collisions;
    }

    // ---- DomainRandEnv.step extras (dqn.py:415-444) ---------------------------------------------
    if (UAV_FLAGS(c) & UAVENV_FLAG_PROX_SHAPING) {
        const double prev_dist = r.prev_dist_nearest;                             // dqn.py:417
        double curr = dist_nearest_with_data<G>(s, act, e.ux, e.uy);
        reward = (prev_dist > 0) ? reward + c.prox_eta * (prev_dist - curr) : reward;
        r.prev_dist_nearest = curr;
    }
    r.first_full_coverage_step = ((r.first_full_coverage_step < 0) & (visited_cnt == n)) ? e.step
                                                                                       : r.first_full_coverage_step;
    if (UAV_FLAGS(c) & UAVENV_FLAG_JAIN_BONUS) {                                  // dqn.py:434-442
        int rated;
        const double j = jains_index<G>(s, act, nullptr, &rated);
        reward = rated > 0 ? reward + c.jain_weight * (j - 0.5) / n : reward;     // `if rates:` no bonus before any sensor generated data
    }
    r.episode_return += reward;

    if (in_batch && gl == 0) {
        if (o.actions_out != nullptr && UAV_POLICY(a) != UAVENV_POLICY_ACTIONS) o.actions_out[out] = action;
        if (o.reward) o.reward[out] = reward;
        if (o.reward32) o.reward32[out] = (float)reward;
        if (o.done) o.done[out] = truncated ? 1 : 0;
        if (o.aux) reinterpret_cast<float4*>(o.aux)[out] = make_float4((float)action, (float)reward, truncated ? 1.0f : 0.0f,
                                                                    __int_as_float(term_ticket));
    }

    // ---- SB3 VecEnv auto-reset: episode stats, reset, first observation of the new episode -----------
    uint32_t reset_w3 = 0u; bool have_reset_w3 = false;
    if (__any(do_reset)) {
        {   // dqn.py:305-331 last_episode_stats (+ Monitor r/l)
            double std_rates;
            double jain = jains_index<G>(s, act, &std_rates);
            double tg = gsum<G>(act ? s.gen : 0.0), tc = gsum<G>(act ? s.tx : 0.0), tl = gsum<G>(act ? s.lost : 0.0);
            if (do_reset && gl == 0) {
                UavEnvEpisodeStats st;
                st.episode_return = r.episode_return; st.total_reward = r.total_reward;
                st.total_generated = tg; st.total_collected = tc; st.total_lost = tl;
                st.battery_remaining = r.battery; st.jains_index = jain; st.fairness_std = std_rates;
                st.length = r.current_step; st.sensors_visited = visited_cnt; st.num_sensors = n;
                st.grid_w = r.grid_w; st.grid_h = r.grid_h;
                st.first_full_coverage_step = r.first_full_coverage_step;
                st.episode = r.episode; st.valid = 1u;
                p.stats[env] = st;
                if (p.term_rec != nullptr) p.term_rec[env] = r;              // the record of the terminal step (before the reset below)
            }
            if (p.term_sensors != nullptr && do_reset) {                  // kernel-uniform pointer test
                double* q = p.term_sensors + ((size_t)env * 3u) * G + gl;
                q[0] = s.b; q[G] = s.gen; q[2 * G] = s.tx;
            }
        }
        float zD = 0.f, zE = 0.f;
        const bool draw_layout = (UAV_FLAGS(c) & UAVENV_FLAG_RANDOM_LAYOUT) != 0;
        reset_group<G, kLean>(c, p, s, r, env, in_batch, do_reset, draw_layout, zD, zE, reset_w3, have_reset_w3);
        if (UAV_FLAGS(c) & UAVENV_FLAG_FAR_START) far_start<G>(c, s, r, act, do_reset);
        if (do_reset) { r.uav_x = r.start_x; r.uav_y = r.start_y; }
        double det0 = rssi_deterministic(c, r.uav_x, r.uav_y, s.sx, s.sy);
        float* dst = (in_batch && o.obs) ? o.obs + out * (size_t)c.obs_dim : nullptr;
        observe<G, kLean>(c, s, r.num_sensors, r.grid_w, r.grid_h, r.inv_grid_w, r.inv_grid_h, r.uav_x, r.uav_y, r.battery, act, do_reset, det0, zD, zE, dst,
                          o.write_through != 0);
        if (UAV_FLAGS(c) & UAVENV_FLAG_PROX_SHAPING) {
            double d0 = dist_nearest_with_data<G>(s, act, r.uav_x, r.uav_y);
            if (do_reset) r.prev_dist_nearest = d0;
        }
        wrote_pos |= draw_layout && do_reset;
        dirty |= do_reset ? 3u : 0u;                    // the reset zeroed tx / lost
    }
    if (kRegs) *rec_out = r;
#ifdef UAV_ABL_NOREC          // timing-only ablation build
    else if (gl == 0 && reward == -12345.0) {
#else
    else if (gl == 0) {
#endif
#ifndef UAV_ABL_FULLREC
        // words 24..31 (grid height, sensor count, env index, status, reciprocals) only change at a reset or on an
        // invalid action: without either (decided per wave when the group is the wave) the first 96 bytes are all there is
        if (G == 64 && !(__any(do_reset) != 0 || status_bits != 0u)) {
            union { UavEnvRecord rec; uint4 q[8]; } u;
            u.rec = r;
            uint4* dst = reinterpret_cast<uint4*>(rec_out);
#pragma unroll
            for (int i = 0; i < 6; i++) dst[i] = u.q[i];
        } else
#endif
            *rec_out = r;
    }
    live |= gl < r.num_sensors;
    status_or |= r.status;
    // The action this environment draws NEXT (after a possible auto-reset: exact), tagged with its (episode, step): lane
    // 0's spare word of the observation-noise call made above.  The step kernel leaves it for the next launch (which uses
    // it for scheduling and, after checking the tag against the record, as the action); the rollout kernel carries it.
    next_word = 0u;
    if (UAV_POLICY(a) == UAVENV_POLICY_RANDOM) {
        const uint32_t ns = (uint32_t)(r.current_step + 1);
#ifdef UAV_ABL_CHEAPACTION
        const uint32_t na = (r.env_index * 7u + ns * 3u + r.episode) % 5u;
#else
        uint32_t w3n;
        if (G == 64) {
            const bool was_reset = __any(do_reset) != 0;                           // wave-uniform
            const bool have = was_reset ? have_reset_w3 : z.have_w3;
            w3n = (uint32_t)__builtin_amdgcn_readlane((int)(was_reset ? reset_w3 : z.w3), 0);
            if (!have) w3n = noise_words(c.seed, r.env_index, r.episode, ns - 1u, 0u, 0).w3;   // noise came from a tape: draw it
        } else {
            const bool have = do_reset ? have_reset_w3 : z.have_w3;                // uniform within a lane group
            w3n = gshfl<G>(do_reset ? reset_w3 : z.w3, 0);                         // the group's lane 0
            if (__any(!have)) { const uint32_t dw = noise_words(c.seed, r.env_index, r.episode, ns - 1u, 0u, 0).w3; w3n = have ? w3n : dw; }
        }
        const uint32_t na = (uint32_t)random_action(w3n);
#endif
        next_word = na | hint_tag(r.episode, ns);
        if (!kRegs && o.hint_out != nullptr && gl == 0) o.hint_out[env] = next_word;
    }
    UAV_PHASE(7);

    action_out = action;
}

// ---------------------------------------------------------------------------------------------
// step kernel: one launch = one step() of every environment
// ---------------------------------------------------------------------------------------------
template <int G, bool kLean, int kWaves, bool kDefC>
__global__ __launch_bounds__(kWaves * 64, (G == 64 ? 4 : 2)) void uav_step_kernel(
        // The first seven arguments repeat fields of the two structs: they are the pointers the first loads of a wave
        // need, and as leading scalar arguments they arrive PRELOADED in SGPRs with the wave launch (gfx950 kernarg
        // preload, -mllvm -amdgpu-kernarg-preload-count=8 in build.py: 16 SGPRs, the maximum) instead of behind a kernarg-segment
        // round trip.  launch_word = num_envs | grid blocks << 32 | the handle's sensor count << 56 | balance << 63 (gridDim.x itself would be a load from the hidden
        // arguments in front of everything else); seed = the Philox key.
        const Consts* cptr, char* sensor_base, uint64_t lanes, UavEnvRecord* rec_base, const uint32_t* hint_in,
        const int32_t* actions, uint64_t launch_word, uint64_t seed, Ptrs p_in, StepArgs a_in) {
    // The two structs are NOT taken from the parameters (that would load every field at the kernel entry and spill
    // them): they are read in place from the kernarg segment, field by field where used, like the constants block.
    struct Kernargs { const Consts* cptr; char* sensor_base; uint64_t lanes; UavEnvRecord* rec; const uint32_t* hint_in;
                      const int32_t* actions; uint64_t launch_word; uint64_t seed; Ptrs p; StepArgs a; };
    typedef const __attribute__((address_space(4))) unsigned char* KA;
    KA ka = (KA)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) Ptrs& p = *(const __attribute__((address_space(4))) Ptrs*)(ka + offsetof(Kernargs, p));
    const __attribute__((address_space(4))) StepArgs& a = *(const __attribute__((address_space(4))) StepArgs*)(ka + offsetof(Kernargs, a));
    const SensorBase sb{sensor_base, lanes};
    // Touch the four kernarg cache lines behind the preloaded words with the wave's first instructions, so that the loads that
    // need them later (policy word, output block, record epilogue) find them in the scalar cache: -0.08 us per launch when the
    // launches are graph nodes (8.14 -> 8.06 us at 4096 x 50, 6.02 -> 5.96 at 256), nothing for eager launches.  The four
    // destination registers stay allocated until the drain below, which sits where the wave waits for its state anyway.
    static_assert(sizeof(Kernargs) <= 0x140 && sizeof(Kernargs) > 0x100, "the prefetch below covers kernarg lines 1..4");
    uint32_t pf0, pf1, pf2, pf3;
    asm volatile("s_load_dword %0, %4, 0x40\n\ts_load_dword %1, %4, 0x80\n\ts_load_dword %2, %4, 0xc0\n\ts_load_dword %3, %4, 0x100"
                 : "=&s"(pf0), "=&s"(pf1), "=&s"(pf2), "=&s"(pf3) : "s"(ka));
    const int32_t num_envs = (int32_t)(uint32_t)launch_word;
    const bool balance = (launch_word >> 63) != 0ull;
    const uint32_t grid_blocks = (uint32_t)(launch_word >> 32) & 0x00FFFFFFu;
    const int max_sensors = (int)((launch_word >> 56) & 0x7Fu);                // the handle's sensor count (1..64)
    decltype(auto) c = ConstsSel<kDefC>::make(*(const __attribute__((address_space(4))) Consts*)(cptr), seed);
#ifdef UAVENV_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int gl = group_lane<G>();
    // ---- which environments does this wavefront step?  Home mapping: wave w takes unit w (= 64/G consecutive
    // environments).  Balanced mapping: units holding a collect action first, so that the collect steps of this
    // workgroup are dealt round-robin over the SIMDs (waves w, w+4, w+8, w+12 share one).
    constexpr int kEnvsPerWave = 64 / G;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int unit = wave;
    // Two scopes of dealing.  (a) Eight workgroups that run on the SAME XCD (workgroup b runs on XCD b mod 8, so the
    // pool is b0, b0+8, ..., b0+56: the state a CU reads was written through the same L2 one launch earlier) pool their
    // 128 wave units: the k-th collect unit of the pool goes to pool member k mod 8, wave k div 8, so every CU gets its
    // share of the pool's collect steps (within +-1) and deals them over its SIMDs.  (b) Otherwise (lane groups narrower
    // than a wave, a partial last group of workgroups) each workgroup deals its own 16 units.  The bits come from the
    // words described below: one vector load of 2 words per lane for (a), one scalar load for (b).  In (a) the "n-th
    // collect / n-th move unit" selection is done lane-parallel (v_mbcnt prefix counts + ballot): the scalar unit is
    // shared by the 16 waves of a CU, which all run this prologue at the same time.
#ifndef UAV_POOL
#define UAV_POOL 8
#endif
    constexpr int kPool = UAV_POOL, kXcds = 8, kWordsPerLane = kPool * 16 / 64;   // pool units = 16 * kPool, 64 per word slot
    const uint32_t span0 = blockIdx.x - blockIdx.x % (kPool * kXcds);             // kPool * 8 workgroups = 8 pools
    const uint32_t pool_b0 = span0 + blockIdx.x % kXcds;                          // first member of this workgroup's pool
    const bool pooled = (G == 64) & (kWaves == 16) & (span0 + kPool * kXcds <= grid_blocks);
    bool unit_is_global = false;
    uint32_t pool_word = 0u;
    bool have_pool_word = false;
    if (kWaves == 16 && G == 64 && pooled) {
        if (balance) {                                               // kernel-uniform
            const int lane = (int)(threadIdx.x & 63u);
            // pool unit u = 16 * member + wave unit; lane l looks at units l, 64 + l, 128 + l, ... (word slot j = u / 64)
            const bool from_actions = (actions != nullptr) & ((int)((span0 + kPool * kXcds) * kWaves) <= num_envs);
            const uint32_t* src = from_actions ? reinterpret_cast<const uint32_t*>(actions) : hint_in;
            uint32_t w[kWordsPerLane];
            uint64_t m[kWordsPerLane];
            bool bit[kWordsPerLane];
            int c = 0;
#pragma unroll
            for (int j = 0; j < kWordsPerLane; j++)
                w[j] = src[(pool_b0 + kXcds * (uint32_t)(4 * j + (lane >> 4))) * kWaves + (uint32_t)(lane & 15)];
#pragma unroll
            for (int j = 0; j < kWordsPerLane; j++) { bit[j] = (w[j] & 7u) == 4u; m[j] = __ballot(bit[j]); c += __popcll(m[j]); }
            const int i = (int)((blockIdx.x - pool_b0) / kXcds);                   // this workgroup's rank in its pool
            const int c_i = c > i ? (c - i + kPool - 1) / kPool : 0;               // collects k < c with k mod kPool == i
            const bool want_collect = wave < c_i;
            const int before = (c / kPool) * i + (c % kPool < i ? c % kPool : i);  // collects of pool members 0..i-1
            const int target = want_collect ? wave * kPool + i : kWaves * i - before + (wave - c_i);
            // rank of each of this lane's units among the collect (or the move) units of the pool; exactly one unit hits
            int pu = 0, ones_below = 0;
            uint32_t word = 0u;
#pragma unroll
            for (int j = 0; j < kWordsPerLane; j++) {
                const int ones = ones_below + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[j], 0u));
                const int rank = want_collect ? ones : 64 * j + lane - ones;
                const uint64_t h = __ballot((bit[j] == want_collect) & (rank == target));
                if (h != 0ull) {                                             // wave-uniform
                    const int l = __ffsll((long long)h) - 1;
                    pu = 64 * j + l;
                    word = (uint32_t)__builtin_amdgcn_readlane((int)w[j], l);  // the chosen unit's word sits in lane l
                }
                ones_below += __popcll(m[j]);
            }
            unit = (int)((pool_b0 + kXcds * (uint32_t)(pu >> 4)) * kWaves) + (pu & 15);   // a global unit index
            unit_is_global = true;
            pool_word = word;
            have_pool_word = !from_actions;
        }
    } else
    if (kWaves >= 8) {          // 4-wave workgroups: one wave per SIMD, nothing to deal
        // One word per environment of this workgroup, "== 4" meaning "steps a collect action": the caller's action
        // array, or for the in-kernel random policy the words the PREVIOUS launch left in hint_in (one step ahead draw).
        // Fetched with ONE scalar load whose address needs only preloaded arguments and the workgroup id, so it is
        // issued with the first instructions of the wave (the last, partial workgroup of an action array reads the
        // hint words instead of running past the array's end: any bits give a valid schedule).
        typedef const __attribute__((address_space(4))) uint32_t* CU32;
        constexpr int kWords = kWaves * kEnvsPerWave;
        const bool from_actions = (actions != nullptr) & ((int)(blockIdx.x * kWords) + kWords <= num_envs);
        CU32 src = (from_actions ? (CU32)actions : (CU32)hint_in) + (size_t)blockIdx.x * kWords;
        uint32_t mask = 0u;
#pragma unroll
        for (int t = 0; t < kWords; t++) mask |= ((src[t] & 7u) == 4u ? 1u : 0u) << (t / kEnvsPerWave);
        if (balance) {                                               // kernel-uniform
            const uint32_t full = kWaves >= 32 ? 0xFFFFFFFFu : ((1u << (kWaves & 31)) - 1u);
            const int nc = __popc(mask);
            uint32_t m = wave < nc ? mask : (~mask & full);
            for (int k = wave < nc ? wave : wave - nc; k > 0; k--) m &= m - 1u;   // scalar: drop the k lowest candidates
            unit = __ffs((int)m) - 1;
        }
    }
    const uint32_t env = ((unit_is_global ? 0u : blockIdx.x * kWaves) + (uint32_t)unit) * kEnvsPerWave + uni<G>((int)((threadIdx.x & 63u) / G));
    const uint32_t idx = env * G + gl;
    const bool in_batch = env < (uint32_t)num_envs;
    Sensor s;
    load_sensor_live<G>(sb, idx, s, gl < max_sensors);
    asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(pf0), "s"(pf1), "s"(pf2), "s"(pf3));    // (drains the kernarg prefetch above)
    int wt_word = 0;                                  // StepArgs::write_through, handed back by step_once (it loads the output block)
    bool wrote_pos = false, live = false;
    uint32_t status_or = 0u, dirty = 0u;
    int action = 0;
    uint32_t next_word = 0u;
    uint32_t hint_word = 0u;
    if (G == 64 && actions == nullptr)
        hint_word = have_pool_word ? pool_word : ((const __attribute__((address_space(4))) uint32_t*)hint_in)[env];
    else if (actions == nullptr) hint_word = hint_in[env];              // narrower lane groups: one word per group, vector load
    // G = 64: the record is wave-uniform and nobody else touches it during the launch, so it is read with scalar loads
    // (constant address space: straight into SGPRs, no v_readfirstlane) and written once by lane 0 at the end.
    if (G == 64) {
        typedef const __attribute__((address_space(4))) UavEnvRecord* ScalarRec;
        step_once<G, kLean, false, ScalarRec>(c, p, a, env, in_batch, (ScalarRec)(rec_base + env), rec_base + env, s, wrote_pos,
                                              live, dirty, status_or, action, next_word, wt_word, hint_word);
    } else
        step_once<G, kLean>(c, p, a, env, in_batch, rec_base + env, rec_base + env, s, wrote_pos, live, dirty, status_or, action, next_word, wt_word, hint_word);
    store_sensor<G>(sb, idx, s, wrote_pos, live, dirty, wt_word != 0);
    if (gl == 0 && status_or) atomicOr(p.status, status_or);
#ifdef UAVENV_STAMPS
    if (p.stamps != nullptr && (threadIdx.x & 63u) == 0) {
        unsigned long long* q = p.stamps + ((size_t)blockIdx.x * kWaves + threadIdx.x / 64) * 8;
        q[0] = st_t0; q[1] = __builtin_amdgcn_s_memtime(); q[2] = st_r0; q[3] = __builtin_amdgcn_s_memrealtime();
        q[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        q[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        q[6] = (unsigned long long)action; q[7] = 0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// fused rollout kernel: K consecutive steps per launch (open-loop actions or the in-kernel random
// policy).  Sensor state (VGPRs) and the 128-byte record (SGPRs when G = 64) stay in registers for the
// whole launch; every step still writes its observation / reward / done block ([K][E][...] layout, e.g. K consecutive
// slots of a replay ring), so the result is bit-identical to K single-step launches.
// ---------------------------------------------------------------------------------------------
template <int G, bool kLean, bool kDefC>
__global__ __launch_bounds__(kSmallBlockThreads, (G == 64 ? 4 : 2)) void uav_rollout_kernel(const Consts* cptr, Ptrs p_in, StepArgs a_in,
                                                                                     int32_t num_steps) {
    // as in the step kernel, the argument structs are read in place from the kernarg segment (constant address space)
    struct Kernargs { const Consts* cptr; Ptrs p; StepArgs a; int32_t num_steps; };
    typedef const __attribute__((address_space(4))) unsigned char* KA;
    KA ka0 = (KA)__builtin_amdgcn_kernarg_segment_ptr();
    const __attribute__((address_space(4))) Ptrs& p = *(const __attribute__((address_space(4))) Ptrs*)(ka0 + offsetof(Kernargs, p));
    const __attribute__((address_space(4))) StepArgs& a = *(const __attribute__((address_space(4))) StepArgs*)(ka0 + offsetof(Kernargs, a));
    const int gl = group_lane<G>();
    const uint32_t grp = threadIdx.x / G;
    const uint32_t env = blockIdx.x * (kSmallBlockThreads / G) + uni<G>((int)grp);
    const uint32_t idx = env * G + gl;
    const bool in_batch = env < (uint32_t)a.num_envs;
    Sensor s;
    load_sensor<G>(p, idx, s);
    UavEnvRecord rr = p.rec[env];
    bool wrote_pos = false, live = false;
    int wt_word = 0;
    uint32_t status_or = 0u, dirty = 0u;
    const size_t E = (size_t)a.num_envs;
    uint32_t carried_word = 0u;      // the random policy's next action, handed from step to step (0: draw it)
    for (int k = 0; k < num_steps; k++) {
        int action = 0;
        // Launder the constants pointer every iteration: otherwise LICM hoists all ~90 invariant scalar loads
        // out of the loop and the SGPR file spills into VGPR lanes (and those into scratch).
        const Consts* cp = cptr;
        asm volatile("" : "+s"(cp));
        CRef ckm = *(const __attribute__((address_space(4))) Consts*)(cp);
        decltype(auto) ck = ConstsSel<kDefC>::make(ckm, ckm.seed);
        KA ka = ka0;                              // same for the argument structs
        asm volatile("" : "+s"(ka));
        const __attribute__((address_space(4))) Ptrs& pk = *(const __attribute__((address_space(4))) Ptrs*)(ka + offsetof(Kernargs, p));
        const __attribute__((address_space(4))) StepArgs& ak = *(const __attribute__((address_space(4))) StepArgs*)(ka + offsetof(Kernargs, a));
        // every step writes block k of the [K][E][...] outputs: the row offset k * E goes to step_once, the argument
        // structs stay untouched (no per-step copies of nine pointers competing for SGPRs)
        step_once<G, kLean, true>(ck, pk, ak, env, in_batch, &rr, &rr, s, wrote_pos, live, dirty, status_or, action, carried_word, wt_word, carried_word, (size_t)k * E);
    }
    store_sensor<G>(p, idx, s, wrote_pos, live, dirty, wt_word != 0);
    if (gl == 0) p.rec[env] = rr;
    if (gl == 0 && status_or) atomicOr(p.status, status_or);
}

// ---------------------------------------------------------------------------------------------
// noise dump: the tapes the NEXT step / NEXT reset would draw (parity harness for Philox mode)
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kSmallBlockThreads) void uav_dump_noise_kernel(const Consts* cptr, Ptrs p, float* step_tape,
                                                                       float* reset_tape, int32_t num_envs) {
    constexpr bool kLean = false;
    UAV_CONSTS(cptr);
    const int gl = group_lane<G>();
    const uint32_t env = blockIdx.x * (kSmallBlockThreads / G) + threadIdx.x / G;
    if (env >= (uint32_t)num_envs) return;
    UavEnvRecord r = p.rec[env];
    if (step_tape) {
        Ptrs q = p; q.step_tape = nullptr;
        StepNoise z;
        draw_step_noise<G, kLean>(c, q, r.env_index, r.episode, env, true, (uint32_t)(r.current_step + 1), true, z);
        finish_zc(z);
        float* t = step_tape + env * (size_t)(UAVENV_TAPE_STEP_SLOTS * G) + gl;
        t[0 * G] = z.zA; t[1 * G] = z.zB; t[2 * G] = z.u; t[3 * G] = z.zC; t[4 * G] = z.zD; t[5 * G] = z.zE;
        t[UAVENV_TAPE_ZP * G] = draw_policy_noise<G, kLean>(c, q, r.env_index, r.episode, env, true, (uint32_t)(r.current_step + 1));
    }
    if (reset_tape) {
        uint32_t ep = r.episode + 1u;
        Words4 w = noise_words(c.seed, r.env_index, ep, 0u, (uint32_t)gl, 2);
        Words4 v = noise_words(c.seed, r.env_index, ep, 0u, (uint32_t)gl, 0);
        float zD, zE;
        normal_pair(v.w0, v.w1, zD, zE);
        const Words4 q = noise_words(c.seed, r.env_index, ep, 0u, 0u, 6);
        float zS, spare;
        normal_pair(q.w0, q.w1, zS, spare);
        float* t = reset_tape + env * (size_t)(UAVENV_RTAPE_SLOTS * G) + gl;
        t[UAVENV_RTAPE_FILL * G] = u24(w.w0); t[UAVENV_RTAPE_ZD * G] = zD; t[UAVENV_RTAPE_ZE * G] = zE;
        t[UAVENV_RTAPE_ZS * G] = gl == 0 ? zS : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// frame stack: SB3 VecFrameStack semantics on device, in place.
//   stacked[e] = [frame_{t-k+1} | ... | frame_t]  (k frames of D floats, oldest first)
//   step:  shift left by one frame, append obs[e]; where done[e], the older frames are zeroed first
//          (the appended observation is then the first one of the new episode);
//   terminal_stacked (optional): for done envs, [old frames shifted | terminal_obs] = SB3's stacked
//          "terminal_observation".
// One wavefront owns one environment row: it loads the whole shifted row into registers before it
// stores, so the in-place shift is race free; all accesses are contiguous dwords.
// ---------------------------------------------------------------------------------------------
constexpr int kFsMaxPerLane = 40;      // 64 lanes x 40 floats = 2560 floats per row (k*D <= 2560)

__global__ __launch_bounds__(kSmallBlockThreads) void uav_frame_stack_kernel(float* stacked, const float* obs, const uint8_t* done,
                                                                      const float* terminal_obs, float* terminal_stacked,
                                                                      int32_t num_envs, int32_t k, int32_t D) {
    const int lane = threadIdx.x & 63;
    const int env = __builtin_amdgcn_readfirstlane(blockIdx.x * (kSmallBlockThreads / 64) + (threadIdx.x >> 6));   // (wave-uniform, and the compiler knows)
    if (env >= num_envs) return;
    const int row = k * D, keep = row - D;
    float* s = stacked + (size_t)env * row;
    const float* o = obs + (size_t)env * D;
    // every load of the row -- the k-1 older frames, the new observation, the done flag -- is issued before the first wait: one
    // trip to memory per environment (the flag first, then the frames, then the observation was three).  Slots past the row are
    // skipped by a scalar branch; inside the row the address is selected, not the load predicated.
    float v[kFsMaxPerLane];
#pragma unroll
    for (int j = 0; j < kFsMaxPerLane; j++) {
        v[j] = 0.0f;
        if (64 * j < row) {
            const int i = lane + 64 * j;
            const float* src = (i < keep) ? s + i + D : o + ((i < row) ? i - keep : 0);     // old frames 1..k-1 -> positions 0..k-2 | obs
            v[j] = *src;
        }
    }
    const uint8_t dflag = done != nullptr ? done[env] : (uint8_t)0;
#pragma unroll
    for (int j = 0; j < kFsMaxPerLane; j++) asm volatile("" : "+v"(v[j]));     // (keeps the loads above the flag's wait)
    const bool dn = dflag != 0;
    if (dn && terminal_stacked != nullptr && terminal_obs != nullptr) {
        float* t = terminal_stacked + (size_t)env * row;
        const float* to = terminal_obs + (size_t)env * D;
#pragma unroll
        for (int j = 0; j < kFsMaxPerLane; j++) {
            const int i = lane + 64 * j;
            if (i < keep) t[i] = v[j];
            else if (i < row) t[i] = to[i - keep];
        }
    }
#pragma unroll
    for (int j = 0; j < kFsMaxPerLane; j++) {
        if (64 * j < row) {
            const int i = lane + 64 * j;
            if (i < keep) s[i] = dn ? 0.0f : v[j];
            else if (i < row) s[i] = v[j];
        }
    }
}

hipError_t launch_frame_stack(float* stacked, const float* obs, const uint8_t* done, const float* terminal_obs,
                              float* terminal_stacked, int32_t num_envs, int32_t k, int32_t D, hipStream_t s) {
    if (k < 1 || D < 1 || (long long)k * D > 64LL * kFsMaxPerLane) return hipErrorInvalidValue;
    dim3 block(kSmallBlockThreads), grid((unsigned)((num_envs + kSmallBlockThreads / 64 - 1) / (kSmallBlockThreads / 64)));
    uav_frame_stack_kernel<<<grid, block, 0, s>>>(stacked, obs, done, terminal_obs, terminal_stacked, num_envs, k, D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// the plain-environment configuration the lean specialisation covers
static inline bool lean_ok(const Consts& c, const Ptrs& p, const StepArgs& a) {
    return (c.flags & ~(uint32_t)UAVENV_FLAG_AUTO_RESET) == 0u && c.fps == 3 && p.step_tape == nullptr &&
           p.reset_tape == nullptr && a.policy <= UAVENV_POLICY_RANDOM;
}
static inline size_t lds_bytes(int, const Consts&) { return UAV_LDS_PAD; }     // no LDS: 0 unless a dev build caps occupancy

#define UAV_DISPATCH_G(G_, CALL)          \
    switch (G_) {                         \
        case 16: { constexpr int G = 16; CALL; } break; \
        case 32: { constexpr int G = 32; CALL; } break; \
        case 64: { constexpr int G = 64; CALL; } break; \
        default: return hipErrorInvalidValue;           \
    }

hipError_t launch_init(int Gw, int padded_envs, const Consts& c, const Consts* dc, const Ptrs& p, uint32_t env_index_base,
                       int32_t grid_w, int32_t grid_h, int32_t n, float start_x, float start_y, hipStream_t s) {
    dim3 block(kSmallBlockThreads), grid((unsigned)(padded_envs / (kSmallBlockThreads / Gw)));
    UAV_DISPATCH_G(Gw, (uav_init_kernel<G><<<grid, block, 0, s>>>(dc, p, env_index_base, grid_w, grid_h, n, start_x, start_y)));
    return hipGetLastError();
}
hipError_t launch_reset(int Gw, int padded_envs, const Consts& c, const Consts* dc, const Ptrs& p, const ResetArgs& a, hipStream_t s) {
    dim3 block(kSmallBlockThreads), grid((unsigned)(padded_envs / (kSmallBlockThreads / Gw)));
    UAV_DISPATCH_G(Gw, (uav_reset_kernel<G><<<grid, block, lds_bytes(Gw, c), s>>>(dc, p, a)));
    return hipGetLastError();
}
bool step_uses_big_workgroups(int Gw, int padded_envs) { return (long)padded_envs * Gw / 64 >= 16L * 256L; }

hipError_t launch_step(int Gw, int padded_envs, const Consts& c, const Consts* dc, const Ptrs& p, const StepArgs& a, bool default_consts,
                       hipStream_t s) {
    // 16-wave workgroups (one per CU, collect steps dealt over its SIMDs) once every CU gets at least that many waves;
    // smaller batches use 4-wave workgroups so that they still spread over all CUs
    const long waves = (long)padded_envs * Gw / 64;
    const bool big = step_uses_big_workgroups(Gw, padded_envs);
    const int wg_waves = big ? kBlockThreads / 64 : kSmallBlockThreads / 64;
    dim3 block(wg_waves * 64), grid((unsigned)(waves / wg_waves));
    const uint64_t be = ((uint64_t)(a.balance != 0) << 63) | ((uint64_t)(c.n_max & 0x7F) << 56) | ((uint64_t)(grid.x & 0x00FFFFFFu) << 32) |
                        (uint64_t)(uint32_t)a.num_envs;
#define UAV_STEP_LAUNCH(LEAN, WV, DEFC) uav_step_kernel<G, LEAN, WV, DEFC><<<grid, block, lds_bytes(Gw, c), s>>>( \
        dc, p.sensor_base, p.lanes, p.rec, a.hint_in, a.actions, be, c.seed, p, a)
    if (lean_ok(c, p, a) && default_consts) {       // the reference configuration: constants as instruction literals
        if (big) { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(true, kBlockThreads / 64, true))); }
        else { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(true, kSmallBlockThreads / 64, true))); }
    } else if (lean_ok(c, p, a)) {
        if (big) { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(true, kBlockThreads / 64, false))); }
        else { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(true, kSmallBlockThreads / 64, false))); }
    } else {
        if (big) { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(false, kBlockThreads / 64, false))); }
        else { UAV_DISPATCH_G(Gw, (UAV_STEP_LAUNCH(false, kSmallBlockThreads / 64, false))); }
    }
#undef UAV_STEP_LAUNCH
    return hipGetLastError();
}
hipError_t launch_rollout(int Gw, int padded_envs, const Consts& c, const Consts* dc, const Ptrs& p, const StepArgs& a,
                          int32_t num_steps, bool default_consts, hipStream_t s) {
    dim3 block(kSmallBlockThreads), grid((unsigned)(padded_envs / (kSmallBlockThreads / Gw)));
    if (lean_ok(c, p, a) && default_consts) { UAV_DISPATCH_G(Gw, (uav_rollout_kernel<G, true, true><<<grid, block, lds_bytes(Gw, c), s>>>(dc, p, a, num_steps))); }
    else if (lean_ok(c, p, a)) { UAV_DISPATCH_G(Gw, (uav_rollout_kernel<G, true, false><<<grid, block, lds_bytes(Gw, c), s>>>(dc, p, a, num_steps))); }
    else { UAV_DISPATCH_G(Gw, (uav_rollout_kernel<G, false, false><<<grid, block, lds_bytes(Gw, c), s>>>(dc, p, a, num_steps))); }
    return hipGetLastError();
}
hipError_t launch_dump_noise(int Gw, int padded_envs, const Consts& c, const Consts* dc, const Ptrs& p, float* step_tape,
                             float* reset_tape, int32_t num_envs, hipStream_t s) {
    dim3 block(kSmallBlockThreads), grid((unsigned)(padded_envs / (kSmallBlockThreads / Gw)));
    UAV_DISPATCH_G(Gw, (uav_dump_noise_kernel<G><<<grid, block, 0, s>>>(dc, p, step_tape, reset_tape, num_envs)));
    return hipGetLastError();
}

}  // namespace uavenv
