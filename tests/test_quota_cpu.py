"""The package caps torch's intra-op pool at the container's CPU quota (generative_models_amd/__init__.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_quota_parsers(tmp_path, monkeypatch):
    import builtins
    import generative_models_amd as gm
    real_open = builtins.open
    files = {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}

    def fake_open(path, *a, **k):
        if path in files:
            p = tmp_path / "f"
            p.write_text(files[path])
            return real_open(p, *a, **k)
        if str(path).startswith("/sys/fs/cgroup"):
            raise OSError(path)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    assert gm._cpu_quota_cores() == 16.0
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert gm._cpu_quota_cores() is None
    del files["/sys/fs/cgroup/cpu.max"]
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "400000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert gm._cpu_quota_cores() == 4.0


def test_cap_applies(monkeypatch):
    import torch
    import generative_models_amd as gm
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(max(4, before))
        monkeypatch.setattr(gm, "_cpu_quota_cores", lambda: 1.5)
        gm._respect_cpu_quota()
        assert torch.get_num_threads() == 2           # rounded up, never truncated
        monkeypatch.setattr(gm, "_cpu_quota_cores", lambda: 1.0)
        gm._respect_cpu_quota()
        assert torch.get_num_threads() == 1
        torch.set_num_threads(max(4, before))
        monkeypatch.setenv("GM_KEEP_THREADS", "1")
        gm._respect_cpu_quota()
        assert torch.get_num_threads() == max(4, before)
    finally:
        torch.set_num_threads(before)


def test_import_has_no_side_effect():
    """Importing the package must not touch torch's global thread pool (applied when the first
    fused engine is built)."""
    import subprocess
    code = ("import torch; torch.set_num_threads(7); import sys; sys.path.insert(0, %r); "
            "import generative_models_amd; assert torch.get_num_threads() == 7" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True, timeout=300)


def test_only_local_world_size_counts_as_ranks_on_this_node(monkeypatch):
    """WORLD_SIZE alone (multi-node launchers, inherited environments) must not shrink this rank's thread share."""
    import generative_models_amd as gm
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "64")
    assert gm._ranks_on_node() == 1 and gm.host_thread_plan()["ranks_on_node"] == 1
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert gm._ranks_on_node() == 8
