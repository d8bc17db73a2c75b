"""CPU: the noise / log10 specification shared by the oracle and the HIP kernels (oracle/uavenv_oracle.c,
csrc/uavenv_noise.h).  The generator is the published Philox4x32 (Salmon et al., SC'11) with 7 rounds: pinned here by the
Random123 known-answer vectors for 10 and for 7 rounds; the Box-Muller and log10 restatements are pinned by their defining properties."""
import ctypes as C
import math

import numpy as np

from oracle import oracle as O


def _philox(ctr, key, rounds=10):
    return [int(x) for x in O.philox(ctr, key, rounds)]


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, "philox4x32 10" rows
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox4x32_7_is_what_the_noise_specification_uses():
    # Random123 kat_vectors, "philox4x32 7" row for the zero counter / key
    assert _philox([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    # the other two inputs of the 10-round rows, 7 rounds of the round function those rows pin
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], 7) == \
        [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    # 10 rounds = 7 rounds followed by 3 more with the key advanced seven times (the Weyl sequence of the key schedule)
    ctr, key = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]
    key7 = [(key[0] + 7 * 0x9E3779B9) & 0xFFFFFFFF, (key[1] + 7 * 0xBB67AE85) & 0xFFFFFFFF]
    assert _philox(_philox(ctr, key, 7), key7, 3) == _philox(ctr, key, 10)
    # the oracle's round count and the library header's are two statements of one number
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "uavenv.h")).read()
    assert int(re.search(r"#define\s+UAVENV_PHILOX_ROUNDS\s+(\d+)", hdr).group(1)) == O.lib().orc_philox_rounds() == 7
    assert O.philox([1, 2, 3, 4], [5, 6]) == O.philox([1, 2, 3, 4], [5, 6], 7)


def _normals(a, b):
    L = O.lib()
    L.orc_normal_pair.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    z0 = np.empty(len(a), np.float32); z1 = np.empty(len(a), np.float32)
    x, y = C.c_float(), C.c_float()
    for i in range(len(a)):
        L.orc_normal_pair(int(a[i]), int(b[i]), C.byref(x), C.byref(y))
        z0[i], z1[i] = x.value, y.value
    return z0, z1


def test_box_muller_matches_its_definition():
    """z0 + i z1 = sqrt(-2 ln u1) * exp(i theta), u1 = ((a>>8)+1) 2^-24, theta = quadrant*pi/2 + (f - 1/2) pi/2."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64); b = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64)
    a[:4] = [0, 0xFFFFFFFF, 0xFFFFFF00, 0x100]; b[:4] = [0, 0xFFFFFFFF, 0x40000000, 0xC0000000]
    z0, z1 = _normals(a, b)
    u1 = ((a >> np.uint64(8)) + np.uint64(1)).astype(np.float64) * 2.0 ** -24
    r = np.sqrt(-2.0 * np.log(u1))
    kb = b >> np.uint64(8)
    q = (kb >> np.uint64(22)).astype(np.float64)
    f = (kb & np.uint64(0x3FFFFF)).astype(np.float64) * 2.0 ** -22
    th = q * (math.pi / 2) + (f - 0.5) * (math.pi / 2)
    assert np.max(np.abs(z0 - r * np.cos(th))) < 4e-6        # float32 evaluation of a value up to 5.8
    assert np.max(np.abs(z1 - r * np.sin(th))) < 4e-6
    assert np.all(np.isfinite(z0)) and np.all(np.isfinite(z1))
    # moments of the two streams (20 000 samples each: 5 sigma bands)
    for z in (z0[4:], z1[4:]):
        assert abs(z.mean()) < 5 / math.sqrt(len(z)) and abs(z.var() - 1.0) < 5 * math.sqrt(2 / len(z))


def test_log10_f32_is_the_correctly_rounded_float32_log10():
    L = O.lib()
    L.orc_log10_f32.argtypes = [C.c_float]; L.orc_log10_f32.restype = C.c_float
    rng = np.random.default_rng(6)
    d = np.concatenate([np.float32(100.0) + rng.random(60000, dtype=np.float32) * np.float32(14100.0),   # the path's range
                        np.exp(rng.uniform(-80, 80, 20000)).astype(np.float32),
                        np.array([1.0, 10.0, 100.0, 1000.0, 1e-3, 2.0, 0.5], dtype=np.float32)])
    got = np.array([L.orc_log10_f32(float(x)) for x in d], dtype=np.float32)
    # float64 log10 (error ~1e-16 relative) rounded once to float32 = correctly rounded except when the float64
    # value sits within ~1e-16 of a float32 rounding boundary (probability ~1e-8 per sample)
    want = np.log10(d.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, want)
