"""In-tree builds of the native pieces (no pip install, no JIT cache).

  librmqtt_gpu_router.so  — the product: HIP kernels (gfx950) + host table compiler
                            + the C ABI of include/rmqtt_gpu_router.h   (hipcc)
  librmqtt_workload.so    — the seeded synthetic workload generator     (g++)

The ``.so`` files are git-ignored but travel to the GPU box with the tree snapshot.
"""
import contextlib
import fcntl
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")

GPU_LIB = os.path.join(HERE, "librmqtt_gpu_router.so")
WL_LIB = os.path.join(HERE, "librmqtt_workload.so")
HOST_LIB = os.path.join(HERE, "librmqtt_host_router.so")

GPU_SRCS = ["c_abi.cpp", "table.cpp", "retain.cpp", "kernels.hip", "order.hip"]
GPU_HDRS = ["table.hpp", "topic.hpp", "device.hpp", "kernels.hpp", "retain.hpp", "retain_abi.inc", "group_abi.inc", "match_core.hpp", "expand_compact.inc", "dedup.inc", "expand_tuple.inc", "prep_batched.inc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


@contextlib.contextmanager
def _build_lock():
    """Several ranks of one job may import the package at once (torch.distributed.run): only one of
    them compiles, the others wait and then find the library fresh."""
    fd = os.open(os.path.join(HERE, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def _compile(cmd, target):
    """Compile to a temporary name and rename: a reader never sees a half-written library."""
    tmp = f"{target}.tmp{os.getpid()}"
    subprocess.check_call([tmp if c == target else c for c in cmd])
    os.replace(tmp, target)


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_workload(force=False):
    src = os.path.join(CSRC, "workload.cpp")
    if force or _stale(WL_LIB, [src]):
        with _build_lock():
            if force or _stale(WL_LIB, [src]):
                _compile(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", WL_LIB, src], WL_LIB)
    return WL_LIB


def build_gpu(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in GPU_SRCS]
    deps = srcs + [os.path.join(CSRC, h) for h in GPU_HDRS] + [os.path.join(INCLUDE, "rmqtt_gpu_router.h")]
    if not (force or _stale(GPU_LIB, deps)):
        return GPU_LIB
    with _build_lock():
        if not (force or _stale(GPU_LIB, deps)):
            return GPU_LIB
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-Wall", "-Wno-unused-result", "-I", INCLUDE, "-I", CSRC, "-x", "hip"]
        cmd += os.environ.get("RGR_EXTRA_FLAGS", "").split()      # tuning sweeps: -DRGR_EXPAND_THREADS=... etc.
        cmd += srcs + ["-o", GPU_LIB]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        _compile(cmd, GPU_LIB)
    return GPU_LIB


def build_host_router(force=False):
    """C++ mirror of the reference's Router trait over the C ABI + its test shim."""
    hdir = os.path.join(HERE, "host")
    srcs = [os.path.join(hdir, f) for f in ("gpu_router.cpp", "gpu_shared.cpp", "gpu_retain.cpp", "raft_snapshot.cpp", "router_capi.cpp")]
    deps = srcs + [os.path.join(hdir, f) for f in ("gpu_router.hpp", "gpu_shared.hpp", "gpu_retain.hpp", "raft_snapshot.hpp")] + [os.path.join(INCLUDE, "rmqtt_gpu_router.h"), GPU_LIB]
    if force or _stale(HOST_LIB, deps):
        with _build_lock():
            if force or _stale(HOST_LIB, deps):
                cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", INCLUDE, "-I", hdir] + srcs
                cmd += ["-o", HOST_LIB, "-L", HERE, "-lrmqtt_gpu_router", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN"]
                _compile(cmd, HOST_LIB)
    return HOST_LIB


def build_all(force=False):
    build_workload(force)
    build_gpu(force)
    if os.path.exists(os.path.join(HERE, "host", "gpu_router.cpp")):
        build_host_router(force)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv)
    print("built:", GPU_LIB, WL_LIB)
