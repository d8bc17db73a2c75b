"""CPU: the drop-in boundary.  The C-ABI library loads, exports every symbol include/uavenv.h declares,
agrees with the Python struct mirrors, fails with a status code (not a crash) when no GPU is present,
and the product package never touches the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd")


def _native():
    import uavenv_amd  # noqa: F401
    from uavenv_amd import _native as N
    return N


def test_library_exports_every_declared_symbol():
    N = _native()
    header = open(os.path.join(ROOT, "include", "uavenv.h")).read()
    declared = sorted(set(re.findall(r"\b(uavenv_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 24
    lib = C.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/uavenv.h but not exported"
    assert sorted(N.EXPORTS) == declared, "python binding and header disagree on the entry points"
    assert N.lib().uavenv_abi_version() == N.ABI_VERSION == 2


def test_struct_mirrors_match_the_header():
    N = _native()
    cfg = N.default_config()
    assert cfg.struct_size == C.sizeof(N.UavEnvConfig)
    assert N.record_dtype().itemsize == 128
    assert N.episode_stats_dtype().itemsize == 96
    # every config field named in the header exists in the mirror, in the same order
    header = open(os.path.join(ROOT, "include", "uavenv.h")).read()
    body = header.split("typedef struct UavEnvConfig {")[1].split("} UavEnvConfig;")[0]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(uint32_t|int32_t|uint64_t|double)\s+", "", decl)
        names += [re.sub(r"\[.*\]", "", x).strip() for x in decl.split(",")]
    assert names == [f[0] for f in N.UavEnvConfig._fields_]


def test_defaults_equal_the_oracle_defaults():
    """Two independent transcriptions of the reference's constants (product header vs oracle) agree."""
    N = _native()
    from oracle import oracle as O
    a, b = N.default_config(), O.default_config()
    for name, _ in N.UavEnvConfig._fields_:
        va, vb = getattr(a, name), getattr(b, name)
        if hasattr(va, "__len__"):
            assert list(va) == list(vb), name
        else:
            assert va == vb, name
    assert N.lib().uavenv_obs_dim(a) == 63           # uav_env.py:348-355: 3 + 3*20
    a.num_sensors = 50
    assert N.lib().uavenv_obs_dim(a) == 153
    a.num_sensors, a.pad_sensors, a.include_sensor_positions = 10, 50, 1
    assert N.lib().uavenv_obs_dim(a) == 253          # dqn.py padded fps=5 variant


def test_create_rejects_bad_arguments_without_crashing():
    N = _native()
    L = N.lib()
    h = C.c_void_p()
    cfg = N.default_config()
    assert L.uavenv_create(C.byref(cfg), 0, 0, 0, C.byref(h)) == N.E_INVALID
    cfg.num_sensors = 65
    assert L.uavenv_create(C.byref(cfg), 4, 0, 0, C.byref(h)) == N.E_INVALID
    assert b"num_sensors" in L.uavenv_last_error(None)
    cfg = N.default_config()
    cfg.struct_size = 12
    assert L.uavenv_create(C.byref(cfg), 4, 0, 0, C.byref(h)) == N.E_INVALID
    assert L.uavenv_destroy(None) == 0
    assert L.uavenv_step(None, None, None, None, None, None, None, None) == N.E_INVALID


def test_no_gpu_means_an_error_code_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    N = _native()
    h = C.c_void_p()
    cfg = N.default_config()
    rc = N.lib().uavenv_create(C.byref(cfg), 8, 0, 0, C.byref(h))
    assert rc in (N.E_HIP, N.E_ALLOC) and not h.value
    import uavenv_amd as U
    with pytest.raises(RuntimeError):
        U.BatchedUAVEnv(8)                            # the product refuses to run without the HIP device


def test_product_never_reaches_into_the_oracle():
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if "import" in l or "#include" in l or "CDLL" in l)
                assert "oracle" not in code and "liborc" not in code, os.path.join(dirpath, f)
    for top in ("uavenv_amd.py",):
        assert "oracle" not in open(os.path.join(ROOT, top)).read()
