"""Builds the HIP extension IN-TREE: csrc/*.hip -> libuavenv_hip.so next to this file.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libuavenv_hip.so")
SOURCES = ["uavenv_kernels.hip", "uavenv_capi.hip", "uavenv_attention.hip"]
HEADERS = ["uavenv_internal.h", "uavenv_noise.h", os.path.join("..", "..", "include", "uavenv.h")]
# -ffp-contract=off: the float64 state has to follow the reference's (numpy, unfused) operation order.
# kernarg preload: the leading scalar kernel arguments arrive in SGPRs with the wave launch (see uav_step_kernel).
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-bitwise-instead-of-logical",
         "-mllvm", "-amdgpu-kernarg-preload-count=7"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the uavenv HIP extension cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
