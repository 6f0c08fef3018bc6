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


def _ranks_on_node():
    """Ranks that share THIS node's CPU quota: LOCAL_WORLD_SIZE (torchrun sets it).  WORLD_SIZE alone says nothing
    about this node -- a multi-node launcher, or independent single-rank jobs that inherited the variable, would make
    one rank give away its threads to ranks that are not here -- so without LOCAL_WORLD_SIZE the answer is 1."""
    import os
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE") or 1))
    except ValueError:
        return 1


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
    # several ranks on one node share the quota: each takes its share, less the two threads every rank keeps busy
    # whatever torch does -- the launch thread and the native fill worker (8 ranks x 2 = the 16-core quota of the
    # MI355X boxes: a third busy thread per rank is what gets the launch threads throttled)
    ranks = _ranks_on_node()
    cap = max(1, int(math.ceil(cores)))
    if ranks > 1:
        cap = max(1, int(cores // ranks) - 2)
    before = torch.get_num_threads()
    if before > cap:
        torch.set_num_threads(cap)
        import logging
        logging.getLogger("generative_models_amd").info(
            "torch intra-op threads %d -> %d (container CPU quota %.2f cores; GM_KEEP_THREADS=1 keeps them)",
            before, cap, cores)


def host_thread_plan():
    """What bench.py prints in `config.host_threads`: the container's CPU quota and the threads one rank keeps busy."""
    import os

    import torch
    ranks = _ranks_on_node()
    q = _cpu_quota_cores()
    return {"cgroup_quota_cores": q, "host_cores": os.cpu_count(), "ranks_on_node": ranks,
            "per_rank": {"launch_thread": 1, "native_fill_worker": 1,
                         "replay_pool": max(1, int(os.environ.get("GM_HOST_THREADS", "1"))),
                         "torch_intraop": torch.get_num_threads()},
            "busy_threads_on_node": ranks * 2,
            "fits_quota": (q is None) or (ranks * 2 <= q)}
