"""CPU tests of the host RNG replay (csrc/gm_hostrng.cpp through engine.HostReplay): every draw
kind and every variant's per-iteration draw program must reproduce torch's own draws BIT FOR BIT --
values and the final state of the global CPU generator -- because the reference's observable RNG
stream position is part of parity (ns_gan.py:183,208,220-226,
w_gp_gan.py:197, dra_gan.py:200-205, info_gan.py:312-323, vae.py:104)."""
import numpy as np
import pytest
import torch

from generative_models_amd import _lib, engine
from generative_models_amd._lib import DRAW_INFO, DRAW_NORMAL, DRAW_SAMPLER, DRAW_UNIFORM
from generative_models_amd.engine import GANEngine, HostReplay

pytestmark = pytest.mark.skipif(not HostReplay.available(),
                                reason="host replay does not reproduce this torch build's CPU draws")


def test_selfcheck_picks_a_flavour_and_leaves_the_generator_alone():
    torch.manual_seed(77)
    before = torch.get_rng_state().clone()
    HostReplay._flavour = None
    assert HostReplay.available()
    assert HostReplay._flavour in (0, 1, 2)
    assert torch.equal(before, torch.get_rng_state())


@pytest.mark.parametrize("n", [16, 17, 31, 32, 100, 624, 625, 5120, 5121, 10240, 200704])
def test_normal_and_uniform_match_torch(n):
    torch.manual_seed(5)
    torch.randn(3)                                   # odd position inside the mt19937 block
    s0 = torch.get_rng_state()
    ref_n, ref_u = torch.empty(n).normal_(), torch.empty(n).uniform_()
    s1 = torch.get_rng_state()
    torch.set_rng_state(s0)
    a, u = torch.empty(n), torch.empty(n)
    assert HostReplay.run([HostReplay.op(DRAW_NORMAL, n, a, 0), HostReplay.op(DRAW_UNIFORM, n, u, 0)], 1)
    assert torch.equal(a, ref_n) and torch.equal(u, ref_u)
    assert torch.equal(torch.get_rng_state(), s1)


def test_scalar_statement_equals_simd_statement():
    """Flavours 3/4 are the lane-by-lane statement of the same arithmetic as 1/2."""
    fl = HostReplay._flavour
    if fl not in (1, 2):
        pytest.skip("libm flavour in use")
    torch.manual_seed(11)
    s0 = torch.get_rng_state()
    a, b = torch.empty(4096), torch.empty(4096)
    HostReplay.run([HostReplay.op(DRAW_NORMAL, 4096, a, 0)], 1)
    torch.set_rng_state(s0)
    _lib.call("gm_host_replay_flavour", fl + 2)
    try:
        HostReplay.run([HostReplay.op(DRAW_NORMAL, 4096, b, 0)], 1)
    finally:
        _lib.call("gm_host_replay_flavour", fl)
    assert torch.equal(a, b)


def test_small_normal_is_refused_not_approximated():
    """n < 16 takes ATen's scalar double path (cached second sample): not restated -> unsupported."""
    torch.manual_seed(1)
    s0 = torch.get_rng_state().clone()
    out = torch.empty(8)
    assert HostReplay.run([HostReplay.op(DRAW_NORMAL, 8, out, 0)], 1) is False
    assert torch.equal(torch.get_rng_state(), s0)


def test_sampler_matches_dataloader():
    torch.manual_seed(7)
    s0 = torch.get_rng_state()
    ds = torch.utils.data.TensorDataset(torch.zeros(50000, 1), torch.arange(50000))
    dl = torch.utils.data.DataLoader(ds, batch_size=256, shuffle=True)
    ref = [next(iter(dl))[1].clone() for _ in range(4)]          # process_batch, ns_gan.py:222-226
    s1 = torch.get_rng_state()
    torch.set_rng_state(s0)
    idx = torch.empty(4, 256, dtype=torch.int64)
    assert HostReplay.run([HostReplay.op(DRAW_SAMPLER, 256, idx, 256 * 8, a=50000)], 4)
    assert all(torch.equal(idx[i], ref[i]) for i in range(4))
    assert torch.equal(torch.get_rng_state(), s1)


def test_partial_rows_match_full_draw():
    """A data-parallel rank materialises only its rows; the stream advances as for the whole tensor."""
    torch.manual_seed(11)
    s0 = torch.get_rng_state()
    ref_n, ref_u = torch.empty(2048, 20).normal_(), torch.empty(2048).uniform_()
    s1 = torch.get_rng_state()
    torch.set_rng_state(s0)
    d, u = torch.zeros(2048, 20), torch.zeros(2048)
    assert HostReplay.run([HostReplay.op(DRAW_NORMAL, 2048 * 20, d, 0, e0=512 * 20, e1=768 * 20),
                           HostReplay.op(DRAW_UNIFORM, 2048, u, 0, e0=512, e1=768)], 1)
    assert torch.equal(d[512:768], ref_n[512:768]) and torch.equal(u[512:768], ref_u[512:768])
    assert float(d[:512].abs().sum() + d[768:].abs().sum() + u[:512].sum() + u[768:].sum()) == 0.0
    assert torch.equal(torch.get_rng_state(), s1)


def _stub(variant, B, Z, N, d, world=1, rank=0, joint=False, I=24, info=None, R=7):
    """A GANEngine shell whose host rings are plain CPU tensors: enough for _host_views / _program /
    _fill (no device)."""
    from collections import deque
    e = GANEngine.__new__(GANEngine)
    e.variant, e.B, e.Z, e.N, e.I, e.D_steps = variant, B, Z, N, I, d
    e.world, e.rank, e.Bl = world, rank, B // world
    e.z_joint, e.R = joint, R
    e._trace, e._views, e._launched = None, {}, deque()
    e._gate_np = np.zeros(2, dtype=np.int64)
    if info:
        e.zd, e.nd, e.nc = info
    z = lambda *s, **k: torch.zeros(*s, **k)
    h = dict(idx=z(R * d, B, dtype=torch.int64))
    if joint:
        h["z"] = z(R, 2, B, Z)
    else:
        h["zD"], h["zG"] = z(R * d, B, Z), z(R, B, Z)
    if variant == "wgp":
        h["eps"] = z(R * d, B)
    if variant == "info":
        h["zQ"] = z(R, B, Z)
    if variant == "dra":
        h["delta"], h["U"] = z(R * d, B), z(R * d, B, I)
    e.hring = h
    return e


@pytest.mark.parametrize("variant,d,joint,world,rank", [
    ("ns", 1, True, 1, 0), ("ns", 1, False, 1, 0), ("w", 5, False, 1, 0), ("wgp", 1, False, 1, 0),
    ("wgp", 2, False, 1, 0), ("dra", 1, False, 1, 0), ("dra", 3, False, 1, 0),
    ("info", 1, False, 1, 0), ("ns", 1, False, 4, 2), ("wgp", 1, False, 2, 1)])
def test_variant_program_equals_torch_draw_order(variant, d, joint, world, rank):
    """C replay of sub-chunks into the host ring == the per-draw torch path (_draw_D/_draw_G):
    ring contents and RNG state, for a sub-chunk at slot 0 and one at a later slot."""
    info = (6, 10, 4) if variant == "info" else None
    B, Z, N = 64, (20 if info else 12), 5000
    mk = lambda: _stub(variant, B, Z, N, d, world, rank, joint, info=info)
    ea, eb = mk(), mk()
    torch.manual_seed(2024)
    torch.rand(5)
    s0 = torch.get_rng_state()
    ea._replay_ok = True
    ea._fill(0, 2)
    ea._fill(2, 5)
    assert ea._replay_ok, "program fell outside the restated paths"
    s_c = torch.get_rng_state()
    torch.set_rng_state(s0)
    eb._replay_ok = False
    eb._fill(0, 2)                                    # torch, draw by draw
    eb._fill(2, 5)
    assert torch.equal(torch.get_rng_state(), s_c)
    r0, r1 = ea.rank * ea.Bl, (ea.rank + 1) * ea.Bl
    for k in ea.hring:
        a, b = ea.hring[k], eb.hring[k]
        if world > 1 and k != "idx":                  # only this rank's rows are materialised
            a, b = a[..., r0:r1, :] if a.dim() > 2 else a[:, r0:r1], \
                b[..., r0:r1, :] if b.dim() > 2 else b[:, r0:r1]
        assert torch.equal(a, b), k


def test_vae_eps_program_with_ragged_batch():
    """vae.py:104: one randn(B, Z) per batch, the last batch of the epoch ragged."""
    B, Z = 32, 20
    torch.manual_seed(9)
    s0 = torch.get_rng_state()
    ref = [torch.randn(B, Z) for _ in range(3)] + [torch.randn(13, Z)]
    s1 = torch.get_rng_state()
    torch.set_rng_state(s0)
    eps = torch.zeros(4, B, Z)
    assert HostReplay.run([HostReplay.op(DRAW_NORMAL, B * Z, eps, B * Z * 4)], 3)
    assert HostReplay.run([HostReplay.op(DRAW_NORMAL, 13 * Z, eps[3], 0)], 1)
    assert all(torch.equal(eps[i], ref[i]) for i in range(3))
    assert torch.equal(eps[3].view(-1)[:13 * Z].view(13, Z), ref[3])
    assert torch.equal(torch.get_rng_state(), s1)


def test_replay_threads_do_not_change_results():
    torch.manual_seed(21)
    s0 = torch.get_rng_state()
    a, b = torch.empty(8, 5120), torch.empty(8, 5120)
    HostReplay.run([HostReplay.op(DRAW_NORMAL, 5120, a, 5120 * 4)], 8)
    torch.set_rng_state(s0)
    _lib.call("gm_host_replay_threads", 4)
    try:
        HostReplay.run([HostReplay.op(DRAW_NORMAL, 5120, b, 5120 * 4)], 8)
    finally:
        _lib.call("gm_host_replay_threads", 1)
    assert torch.equal(a, b)


def test_fill_worker_matches_inline_replay_and_opens_gate():
    """gm_fill_submit: jobs run in order on the caller's state buffer, results bit-identical to
    gm_host_replay called inline, gate advanced after each job; an unsupported job is sticky and
    leaves later gates closed until gm_fill_reset."""
    import ctypes
    from generative_models_amd import _lib
    lib = _lib.load()
    assert HostReplay.available()
    torch.manual_seed(21)
    s0 = torch.get_rng_state()
    B, Z, N = 64, 20, 5000
    ref_idx, ref_z = torch.empty(6, B, dtype=torch.int64), torch.empty(6, B, Z)
    prog = lambda idx, z: [HostReplay.op(DRAW_SAMPLER, B, idx, B * 8, a=N),
                           HostReplay.op(DRAW_NORMAL, B * Z, z, B * Z * 4)]
    st_ref = s0.clone()
    assert HostReplay.call(st_ref, prog(ref_idx, ref_z), 6) == 0
    idx, z = torch.empty(6, B, dtype=torch.int64), torch.empty(6, B, Z)
    st = s0.clone()
    gate = torch.zeros(2, dtype=torch.int64)
    jobs = []
    for c0, n in ((0, 1), (1, 2), (3, 3)):                      # three sub-chunks, in order
        p = prog(idx[c0:], z[c0:])
        arr = (_lib.DrawOp * len(p))(*p)
        jobs.append(lib.gm_fill_submit(st.data_ptr(), st.numel(), arr, len(p), n, gate.data_ptr(), c0 + n))
    assert jobs == sorted(jobs) and jobs[0] > 0
    assert lib.gm_fill_wait(jobs[-1]) == 0
    assert lib.gm_fill_completed() >= jobs[-1]
    assert int(gate[0]) == 6
    assert torch.equal(idx, ref_idx) and torch.equal(z, ref_z) and torch.equal(st, st_ref)
    # sticky failure: normal_ on < 16 elements is outside the restated paths
    small = torch.empty(8)
    bad = [HostReplay.op(DRAW_NORMAL, 8, small, 0)]
    j1 = lib.gm_fill_submit(st.data_ptr(), st.numel(), (_lib.DrawOp * 1)(*bad), 1, 1, gate.data_ptr(), 100)
    p = prog(idx, z)
    j2 = lib.gm_fill_submit(st.data_ptr(), st.numel(), (_lib.DrawOp * 2)(*p), 2, 1, gate.data_ptr(), 200)
    assert lib.gm_fill_wait(j2) == _lib.GM_EUNSUPPORTED and lib.gm_fill_wait(j1) == _lib.GM_EUNSUPPORTED
    assert int(gate[0]) == 6 and torch.equal(st, st_ref)      # neither job ran / opened a gate
    assert lib.gm_fill_reset() == 0
    j3 = lib.gm_fill_submit(st.data_ptr(), st.numel(), (_lib.DrawOp * 2)(*p), 2, 1, gate.data_ptr(), 7)
    assert lib.gm_fill_wait(j3) == 0 and int(gate[0]) == 7


def _torch_sequence(gen_state, seq, B_info=None):
    """The draws of `seq` made by torch itself on a private generator positioned at gen_state."""
    g = torch.Generator()
    g.set_state(gen_state.clone())
    out = []
    for kind, a, b in seq:
        if kind == "normal":
            out.append(torch.empty(a).normal_(generator=g))
        elif kind == "uniform":
            out.append(torch.empty(a).uniform_(generator=g))
        elif kind == "sampler":                       # DataLoader(shuffle=True): base seed, sampler seed, randperm
            torch.empty((), dtype=torch.int64).random_(generator=g)
            seed = int(torch.empty((), dtype=torch.int64).random_(generator=g).item())
            out.append(torch.randperm(a, generator=torch.Generator().manual_seed(seed))[:b])
        else:                                         # info: randn(B, zd) | one_hot(randint(0, nd)) | randn(B, nc)
            zd, nd, nc = b
            zz = torch.empty(a, zd).normal_(generator=g)
            cat = torch.randint(0, nd, (a,), dtype=torch.long, generator=g)
            cc = torch.empty(a, nc).normal_(generator=g)
            t = torch.zeros(a, zd + nd + nc)
            t[:, :zd] = zz
            t[torch.arange(a), zd + cat] = 1
            t[:, zd + nd:] = cc
            out.append(t)
    return out, g.get_state()


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_draw_programmes_match_torch_bit_for_bit(seed):
    """Randomly composed programmes (normal_ on 16..6000 elements incl. lengths that are not multiples
    of 16, uniform_ on 1..5000, sampler prefixes, InfoGAN's composite noise), started from a generator
    left mid-block with a pending state twist at a random distance: every value and the final
    generator state must equal torch's own."""
    import random
    rnd = random.Random(1000 + seed)
    g0 = torch.Generator().manual_seed(seed * 7919 + 1)
    torch.empty(rnd.randint(0, 700)).uniform_(generator=g0)          # random position inside the 624-word block
    s0 = g0.get_state()
    seq = []
    for _ in range(rnd.randint(3, 9)):
        k = rnd.choice(["normal", "normal", "uniform", "sampler", "info"])
        if k == "normal":
            seq.append((k, rnd.choice([16, 17, 31, 32, 100, 640, 5120, rnd.randint(16, 6000)]), None))
        elif k == "uniform":
            seq.append((k, rnd.choice([1, 7, 256, 1003, rnd.randint(1, 5000)]), None))
        elif k == "sampler":
            n = rnd.choice([64, 1000, 50000])
            seq.append((k, n, rnd.randint(1, min(n, 300))))
        else:
            zd, nd, nc = rnd.choice([(4, 10, 6), (20, 10, 10), (8, 3, 5)])
            # (the composite path is restated for 16-multiples of B*zd and B*nc; other shapes return
            # GM_EUNSUPPORTED and the engine draws them through torch)
            bsz = rnd.choice([b for b in (4, 16, 24, 48, 256) if (b * zd) % 16 == 0 and (b * nc) % 16 == 0])
            seq.append((k, bsz, (zd, nd, nc)))
    ref, s1 = _torch_sequence(s0, seq)
    outs, ops = [], []
    for kind, a, b in seq:
        if kind == "normal":
            t = torch.empty(a); ops.append(HostReplay.op(DRAW_NORMAL, a, t, 0))
        elif kind == "uniform":
            t = torch.empty(a); ops.append(HostReplay.op(DRAW_UNIFORM, a, t, 0))
        elif kind == "sampler":
            t = torch.empty(b, dtype=torch.int64); ops.append(HostReplay.op(DRAW_SAMPLER, b, t, 0, a=a))
        else:
            zd, nd, nc = b
            t = torch.empty(a, zd + nd + nc); ops.append(HostReplay.op(DRAW_INFO, a, t, 0, a=zd, b=nd, c=nc))
        outs.append(t)
    st = s0.clone()
    assert HostReplay.call(st, ops, 1) == 0
    for i, (r, o) in enumerate(zip(ref, outs)):
        assert torch.equal(r, o), (seed, i, seq[i])
    assert torch.equal(st, s1), (seed, seq)


# ---- numpy's legacy global generator (BIR-VAE's noise, bir_vae.py:92-94) replayed in C ---------------------
def _np_state(rs):
    return engine.NumpyReplay._unpack(rs.get_state(legacy=True))


def test_numpy_replay_selfcheck_passes_on_this_host():
    assert engine.NumpyReplay.available()


@pytest.mark.parametrize("seed", [0, 7, 2024])
@pytest.mark.parametrize("threads", [1, 3])
def test_numpy_legacy_normal_bit_for_bit(seed, threads):
    """gm_numpy_legacy_normal_f32 == torch.from_numpy(RandomState.normal(loc, scale, n)).float() bit for bit, for
    odd and even counts (the cached second value of the polar method carries over between calls), positions
    anywhere in the 624-word block, and the generator ends in exactly numpy's state."""
    rs = np.random.RandomState(seed)
    rs.random_sample(seed % 5)                         # position not a multiple of 4
    state = _np_state(rs)
    rnd = np.random.RandomState(seed + 1)
    for _ in range(12):
        n = int(rnd.choice([1, 2, 3, 7, 156, 623, 624, 1248, 4097, 10240, 20480]))
        loc, scale = float(rnd.choice([0.0, 1.5])), float(rnd.choice([1.0, 0.37, 13.3]))
        ref = torch.from_numpy(rs.normal(loc, scale, n)).float()
        got = torch.empty(n)
        engine.NumpyReplay._call(state, loc, scale, n, got.data_ptr(), threads)
        assert torch.equal(ref.view(torch.int32), got.view(torch.int32)), (seed, n)
        end = rs.get_state(legacy=True)
        assert np.array_equal(end[1], state[0]) and (int(end[2]), int(end[3]), float(end[4])) == tuple(state[1:])
        if rnd.rand() < 0.3:                           # other draws in between advance both alike
            rs.random_sample(3)
            state = _np_state(rs)


def test_numpy_replay_fill_matches_the_reference_loop_on_the_global_generator():
    """NumpyReplay.fill == the reference's per-batch np.random.normal(0, set_var, (b, Z)).float() on numpy's GLOBAL
    generator (full batches as one call, the ragged one on its own), and leaves the generator where numpy does."""
    B, Z, sizes = 64, 20, [64, 64, 64, 37]
    saved = np.random.get_state()
    try:
        np.random.seed(99)
        ref = [torch.from_numpy(np.random.normal(0.0, 0.5, size=(b, Z))).float() for b in sizes]
        tail = np.random.normal(size=3)
        np.random.seed(99)
        dst = torch.zeros(6, B, Z)
        assert engine.NumpyReplay.fill(0.5, dst, B, Z, sizes)
        for k, b in enumerate(sizes):
            assert torch.equal(dst[k].view(-1)[:b * Z], ref[k].view(-1)), k
        assert not dst[3].view(-1)[37 * Z:].any() and not dst[4:].any()
        assert np.array_equal(np.random.normal(size=3), tail)
    finally:
        np.random.set_state(saved)


def test_numpy_replay_scalar_candidate_stage_in_a_fresh_process():
    """The candidate stage has an AVX-512 form and a scalar one (chosen once per process): the scalar one, forced
    through GM_NUMPY_SCALAR=1 in a child process, reproduces numpy bit for bit too."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, torch\n"
        "from generative_models_amd import engine\n"
        "rs = np.random.RandomState(11); rs.random_sample(2)\n"
        "st = engine.NumpyReplay._unpack(rs.get_state(legacy=True))\n"
        "for n in (10240, 5, 4097):\n"
        "    ref = torch.from_numpy(rs.normal(0.0, 0.5, n)).float(); got = torch.empty(n)\n"
        "    engine.NumpyReplay._call(st, 0.0, 0.5, n, got.data_ptr(), 1)\n"
        "    assert torch.equal(ref.view(torch.int32), got.view(torch.int32)), n\n"
        "end = rs.get_state(legacy=True)\n"
        "assert np.array_equal(end[1], st[0]) and (int(end[2]), int(end[3]), float(end[4])) == tuple(st[1:])\n"
        "print('scalar ok')\n")
    env = dict(os.environ, GM_NUMPY_SCALAR="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "scalar ok" in out.stdout, out.stderr[-2000:]
