"""MI355X-native hot path of shayneobrien/generative-models."""


def _cpu_quota_cores():
    """CPU bandwidth quota of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


_QUOTA_APPLIED = False


def _respect_cpu_quota(force=True):
    """torch sizes its OpenMP pool by the host's cores (128 on the MI355X boxes) even when the
    container may only use 16 of them: every parallel region then leaves 128 spinning threads that
    burn the whole CFS quota in ~12 ms, and the kernel parks the process -- launch thread included --
    for the rest of the 100 ms period.  Measured on the VAE epoch loop (the GAN loop makes no torch CPU
    calls): 40-80 ms stalls every third epoch, 4.7 -> 2.0 M img/s (profiles/r02_experiments.md).
    Capping the pool at the quota (rounded UP: 1.9 cores keep 2 threads) removes them.  Applied when
    the first fused engine is built -- importing the package changes nothing in the process -- and
    logged once.  GM_KEEP_THREADS=1 opts out."""
    global _QUOTA_APPLIED
    import os
    if os.environ.get("GM_KEEP_THREADS") == "1" or (_QUOTA_APPLIED and not force):
        return
    _QUOTA_APPLIED = True
    cores = _cpu_quota_cores()
    if cores is None:
        return
    import math

    import torch
    cap = max(1, int(math.ceil(cores)))
    before = torch.get_num_threads()
    if before > cap:
        torch.set_num_threads(cap)
        import logging
        logging.getLogger("generative_models_amd").info(
            "torch intra-op threads %d -> %d (container CPU quota %.2f cores; GM_KEEP_THREADS=1 keeps them)",
            before, cap, cores)
