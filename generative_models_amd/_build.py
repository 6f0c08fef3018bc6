"""Build libgm_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

In-tree so that the built .so travels to the GPU box with the repo snapshot and is visible to
the driver's "which native code was loaded" check."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgm_hip.so")
SOURCES = ("gm_gemm.hip", "gm_ops.hip", "gm_fused.hip", "gm_comm.hip")
HOST_SOURCES = ("gm_hostrng.cpp",)       # host-only C++ (RNG protocol replay): g++, linked in


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    # every header under csrc/ (a header added later can never be forgotten here) + the public C-ABI header
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    deps = sources() + [os.path.join(CSRC, h) for h in HOST_SOURCES] + headers + \
        [os.path.join(os.path.dirname(HERE), "include", "gm_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    # compile every translation unit to an object (in parallel), then link: hipcc would otherwise
    # treat the host object as HIP source
    jobs = []
    for src in sources():
        obj = src.rsplit(".", 1)[0] + ".o"
        jobs.append(([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                      "-c", src, "-o", obj], obj))
    for h in HOST_SOURCES:
        # -ffp-contract=off: the float transformations must round exactly like ATen's
        obj = os.path.join(CSRC, h.rsplit(".", 1)[0] + ".o")
        jobs.append(([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                      "-pthread", "-c", os.path.join(CSRC, h), "-o", obj], obj))
    procs = []
    for cmd, _ in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd, cwd=CSRC))
    for (cmd, _), pr in zip(jobs, procs):
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for _, o in jobs] + \
        ["-lpthread", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


def build_stamps(verbose=False):
    """libgm_hip_stamps.so: the same library with gm_gemm.hip compiled under -DGM_STAMPS (per-wave timeline probe,
    tools/wave_timeline.py).  Never loaded by the product: the tool points GM_LIB_PATH at it."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.isfile(hipcc):
        hipcc = "hipcc"
    out = os.path.join(HERE, "libgm_hip_stamps.so")
    obj = os.path.join(CSRC, "gm_gemm_stamps.o")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DGM_STAMPS",
           "-c", os.path.join(CSRC, "gm_gemm.hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    others = [os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o") for s in SOURCES[1:]] + \
        [os.path.join(CSRC, h.rsplit(".", 1)[0] + ".o") for h in HOST_SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-lpthread", "-o", out],
                   check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
