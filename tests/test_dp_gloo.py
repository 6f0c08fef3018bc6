"""Data-parallel semantics on CPU: world_size=2 over gloo (the GPU path uses RCCL with the same
calls).  The compute here is the ORACLE's math (tests may use it); what is under test is the
product's sharding + flat-bucket all-reduce plumbing (generative_models_amd/dp.py, engine host
protocol): N ranks == 1 rank up to fp32 summation order, identical sampling on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from generative_models_amd import dp, engine
from oracle import port

B, I, H, Z, N = 32, 64, 48, 8, 400


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat(params):
    return torch.cat([p.grad.reshape(-1) for p in params])


def _d_loss_rows(model, images, noise, inv_b, variant):
    """Per-shard critic loss with GLOBAL 1/B scaling (what each rank's loss kernel computes)."""
    sx, sg = model.D(images), model.D(model.G(noise))
    if variant == "ns":
        return -(torch.log(sx + 1e-8) + torch.log(1 - sg + 1e-8)).sum() * inv_b
    return (0.5 * (sx - 1) ** 2 + 0.5 * (sg - 0) ** 2).sum() * inv_b        # ls


def _worker(rank, world, port_no, variant, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port_no)
    os.environ["WORLD_SIZE"], os.environ["RANK"], os.environ["LOCAL_RANK"] = \
        str(world), str(rank), str(rank)
    torch.set_num_threads(1)
    w, r, _ = dp.init_from_env(backend="gloo")
    assert (w, r) == (world, rank) and dp.current()[:2] == (world, rank)
    # identical host RNG protocol on every rank -> identical global index batch and noise
    torch.manual_seed(3435)
    data = torch.bernoulli(torch.full((N, I), 0.1307))
    model = port.build("ns" if variant == "ns" else "ls", I, H, Z)
    torch.manual_seed(99)
    idx = np.empty(B, dtype=np.int64)
    engine.draw_sampler_indices(N, B, idx)
    noise = torch.randn(B, Z)
    lo, hi = dp.shard_range(B, world, rank)
    images = data[torch.from_numpy(idx)]
    inv_b = 1.0 / B
    loss_local = _d_loss_rows(model, images[lo:hi], noise[lo:hi], inv_b, variant)
    model.zero_grad()
    loss_local.backward()
    dparams = list(model.D.parameters())
    bucket = _flat(dparams).clone()
    dp.allreduce_sum_(bucket)                       # ONE flat bucket per optimizer
    loss_t = loss_local.detach().clone().reshape(1)
    dp.allreduce_sum_(loss_t)
    # single-process reference on the full batch
    model.zero_grad()
    full = _d_loss_rows(model, images, noise, inv_b, variant)
    full.backward()
    ref = _flat(dparams)
    out_q.put((rank, idx.copy(), float((bucket - ref).abs().max()), float(ref.abs().max()),
               float(loss_t.item()), float(full.item()), lo, hi))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["ns", "ls"])
def test_two_rank_gradients_match_single_process(variant):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port_no, variant, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert np.array_equal(res[0][1], res[1][1]), "ranks must sample the same global batch"
    covered = sorted((r[6], r[7]) for r in res)
    assert covered == [(0, B // 2), (B // 2, B)]
    for r in res:
        assert r[2] <= 1e-6 * max(1.0, r[3]), "all-reduced shard grads != full-batch grads: %r" % (r,)
        assert abs(r[4] - r[5]) <= 1e-6 * max(1.0, abs(r[5]))


def test_shard_range_and_errors():
    assert dp.shard_range(1024, 8, 3) == (384, 512)
    assert [dp.shard_range(256, 4, r) for r in range(4)] == [(0, 64), (64, 128), (128, 192), (192, 256)]
    with pytest.raises(ValueError):
        dp.shard_range(100, 8, 0)
    assert dp.current() == (1, 0, None)


def _replay_worker(rank, world, port_no, out_q):
    """Each rank replays three iterations of the global draw protocol materialising only its own
    noise rows (engine.GANEngine._draw_D/_draw_G on a stand-in engine), then the ranks all-gather
    their rows: the assembled tensors must be the single-process draws, bit for bit."""
    from types import SimpleNamespace
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port_no)
    os.environ["WORLD_SIZE"], os.environ["RANK"], os.environ["LOCAL_RANK"] = \
        str(world), str(rank), str(rank)
    torch.set_num_threads(1)
    dp.init_from_env(backend="gloo")
    Bg, Zg, R = 64, 20, 3                                    # 32 rows x 20 = 640 elements per rank

    def replay(w, r):
        eng = SimpleNamespace(N=N, B=Bg, Bl=Bg // w, world=w, rank=r, variant="ns")
        eng._noise = lambda dst, kind: engine.GANEngine._noise(eng, dst, kind)
        torch.manual_seed(2024)
        s = dict(idx=torch.zeros(R, Bg, dtype=torch.int64), zD=torch.zeros(R, Bg, Zg),
                 zG=torch.zeros(R, Bg, Zg))
        s["idx_np"] = s["idx"].numpy()
        for k in range(R):
            engine.GANEngine._draw_D(eng, s, k)
            engine.GANEngine._draw_G(eng, s, k)
        return s, torch.get_rng_state()

    mine, state = replay(world, rank)
    lo, hi = dp.shard_range(Bg, world, rank)
    ok = True
    for key in ("zD", "zG"):
        parts = [torch.zeros(R, Bg // world, Zg) for _ in range(world)]
        dist.all_gather(parts, mine[key][:, lo:hi].contiguous())
        mine[key + "_all"] = torch.cat(parts, dim=1)
    full, full_state = replay(1, 0)                          # what ONE process would have drawn
    ok = ok and torch.equal(mine["zD_all"], full["zD"]) and torch.equal(mine["zG_all"], full["zG"])
    ok = ok and torch.equal(mine["idx"], full["idx"]) and torch.equal(state, full_state)
    out_q.put((rank, bool(ok), bool(engine._skip_supported())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_assemble_the_single_process_noise_from_their_own_rows():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()
    procs = [ctx.Process(target=_replay_worker, args=(r, world, port_no, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)], res


def test_ragged_batch_row_split_covers_every_row_once():
    """VAEEngine._rows: a batch of b rows (incl. the ragged last one, 50 000 mod 512 = 336) is split over
    the ranks without gaps or overlaps, sizes differing by at most one."""
    from generative_models_amd.engine import VAEEngine

    class E:
        pass
    for world in (1, 2, 3, 4, 8):
        for b in (1, 6, 7, 16, 336, 512, 1000):
            seen, sizes = [], []
            for rank in range(world):
                e = E()
                e.world, e.rank = world, rank
                lo, hi = VAEEngine._rows(e, b)
                assert 0 <= lo <= hi <= b
                seen += list(range(lo, hi))
                sizes.append(hi - lo)
            assert seen == list(range(b)), (world, b)
            assert max(sizes) - min(sizes) <= 1, (world, b, sizes)
