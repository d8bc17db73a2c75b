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
SOURCES = ["uavenv_kernels.hip", "uavenv_capi.hip", "uavenv_attention.hip", "uavenv_replay.hip", "uavenv_learner.hip"]
HEADERS = ["uavenv_internal.h", "uavenv_noise.h", "uavenv_derive.h", "uavenv_default_consts.inc", "gen_default_consts.cpp",
           os.path.join("..", "..", "include", "uavenv.h")]
GENERATED = os.path.join(CSRC, "uavenv_default_consts.inc")
# -ffp-contract=off: the float64 state has to follow the reference's (numpy, unfused) operation order.
# kernarg preload: the leading scalar kernel arguments arrive in SGPRs with the wave launch (see uav_step_kernel).
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-bitwise-instead-of-logical",
         "-mllvm", "-amdgpu-kernarg-preload-count=8"]


# UAVENV_EXTRA_HIPCC_FLAGS in the environment is appended (it takes part in the source hash): e.g. -DUAV_HW_TRANSCENDENTALS builds
# the library whose in-kernel normals come from v_log_f32 / v_sin_f32 / v_cos_f32 -- 6 % faster steps, but a noise stream the CPU
# oracle cannot reproduce bit for bit, so the keyed parity tests only hold for the default build.
FLAGS += [f for f in os.environ.get("UAVENV_EXTRA_HIPCC_FLAGS", "").split() if f]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the uavenv HIP extension cannot be built")


STAMP = LIB + ".srchash"


def source_hash():
    """Hash of everything the library is built from (content, not mtimes: a snapshot copied to the GPU box gets new ones)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS]:
        if os.path.basename(d) == "uavenv_default_consts.inc":
            continue                      # generated from the other inputs
        h.update(open(d, "rb").read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_hash()


def generate_default_consts(verbose=False):
    """csrc/uavenv_default_consts.inc: the default configuration's derived constants as exact hex-float literals, printed by
    a host program built from the SAME uavenv_derive.h the C ABI uses (the kernels' literal specialisation includes it).
    The file is committed (the GPU box's dev tools compile the kernels too); it is rewritten only when its content changes."""
    gen = os.path.join(CSRC, f"_gen_default_consts.{os.getpid()}")          # per process: ranks may get here together
    subprocess.check_call([hipcc(), "-O1", "-std=c++17", "-o", gen, os.path.join(CSRC, "gen_default_consts.cpp")],
                          stderr=None if verbose else subprocess.DEVNULL)
    try:
        text = subprocess.check_output([gen]).decode()
    finally:
        os.remove(gen)
    if not os.path.exists(GENERATED) or open(GENERATED).read() != text:
        tmp = f"{GENERATED}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, GENERATED)
    return text


def build(force=False, verbose=False):
    """Several processes may call this at once (every rank of a torchrun job loads the library): one of them builds under
    an exclusive file lock, into a temporary file that is renamed into place, and the others find a fresh library when
    they get the lock -- nobody can load a half-written one."""
    if not force and not stale():
        return LIB
    import fcntl
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():                  # another process built it while this one waited
                return LIB
            generate_default_consts(verbose)
            tmp = f"{LIB}.{os.getpid()}.tmp"
            cmd = [hipcc()] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd).replace(tmp, LIB))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            stamp_tmp = f"{STAMP}.{os.getpid()}.tmp"
            with open(stamp_tmp, "w") as f:
                f.write(source_hash() + "\n")
            os.replace(stamp_tmp, STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
