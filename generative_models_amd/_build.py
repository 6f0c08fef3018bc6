"""Build libgm_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

In-tree so that the built .so travels to the GPU box with the repo snapshot and is visible to
the driver's "which native code was loaded" check."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgm_hip.so")
SOURCES = ("gm_gemm.hip", "gm_ops.hip", "gm_fused.hip")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "gm_common.h"), os.path.join(CSRC, "gm_head.h"), os.path.join(CSRC, "gm_gather.h"),
                        os.path.join(os.path.dirname(HERE), "include", "gm_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-value"] + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
