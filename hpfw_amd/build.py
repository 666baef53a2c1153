"""In-tree build of libhpfw_gpu.so: hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False, jobs=6):
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "-s", "clean"])
    subprocess.check_call(["make", "-C", csrc, "-s", f"-j{jobs}"])
    so = os.path.join(_HERE, "lib", "libhpfw_gpu.so")
    if not os.path.exists(so):
        raise RuntimeError("hipcc produced no libhpfw_gpu.so")
    if not os.path.exists(os.path.join(_HERE, "lib", "libhpfw_gpu_multi.so")):
        raise RuntimeError("no libhpfw_gpu_multi.so (the multi-GPU host path: multi.cpp + librccl)")
    return so
