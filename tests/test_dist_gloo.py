"""CPU, world_size 2 (gloo): the sharding + all-gather path used for N > 1 GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import dimx  # noqa: F401
    from dimx import dist as dd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = dd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    B = 7                                    # odd on purpose: shards of 4 and 3 rows
    full = torch.arange(B * 5, dtype=torch.int32).view(B, 5)
    lo, hi = dd.shard_bounds(B, r, w)
    got = dd.all_gather_rows(full[lo:hi].clone())
    ok1 = torch.equal(got, full)
    even = torch.arange(8 * 3, dtype=torch.float32).view(8, 3)
    lo, hi = dd.shard_bounds(8, r, w)
    ok2 = torch.equal(dd.all_gather_rows(even[lo:hi].clone()), even)
    # round 5: shard sizes known on every rank -> exactly ONE collective, no count exchange (all_gather_counts must not run);
    # several per-row results packed into one buffer
    lo, hi = dd.shard_bounds(B, r, w)
    real_counts = dd.all_gather_counts
    calls = []
    dd.all_gather_counts = lambda *a, **k: calls.append(1) or real_counts(*a, **k)
    tok = full[lo:hi].clone()
    pred = (full[lo:hi].float() * 0.5).reshape(hi - lo, 5, 1).repeat(1, 1, 3)
    buf = dd.all_gather_rows(dd.pack_rows(tok, pred), dd.shard_counts(B))
    tok_all, pred_all = dd.unpack_rows(buf, [((5,), torch.int32), ((5, 3), torch.float32)])
    ok3 = (not calls) and torch.equal(tok_all, full) and torch.equal(pred_all, (full.float() * 0.5).reshape(B, 5, 1).repeat(1, 1, 3))
    dd.all_gather_counts = real_counts
    mx = dd.max_over_ranks(float(rank + 1))
    dd.barrier()
    q.put((rank, ok1, ok2 and ok3, mx))
    dist.destroy_process_group()


def test_shard_and_allgather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok1, ok2, mx in res:
        assert ok1 and ok2 and mx == 2.0


def test_shard_bounds_partition():
    sys.path.insert(0, ROOT)
    import dimx  # noqa: F401
    from dimx import dist as dd
    for n in (1, 7, 256, 2048):
        for w in (1, 2, 3, 8):
            b = [dd.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


# ---------------------------------------------------------------------------------------------------------------
# The sharded evaluation paths (x_engine_pt.generate_sharded / evaluate_test_epoch) with a CPU stub standing in for
# SLMFT: its outputs depend on the clip CONTENT, on the global batch row (batch_row_offset -- the VQ decoder's
# positional quirk) and on a counter-based noise stream indexed by the global row (shard=(lo, total)), so the
# world-size-2 result can only equal the single-process result if both offsets are plumbed through.
# ---------------------------------------------------------------------------------------------------------------
class _StubListener(torch.nn.Module):
    def forward(self, v_speaker, v_listener, v_audio, mask, mode="val", n_samples=1, batch_row_offset=0, shard=None,
                return_tokens=False, noise=None, seed=7, **kw):
        B, T, _ = v_speaker.shape
        off, total = shard if shard is not None else (0, B)
        S = int(n_samples)
        rows = torch.arange(B, dtype=torch.float32) + batch_row_offset                   # "positional row"
        base = v_speaker[:, 1:, :] * 0.5 + v_audio[:, 1:, :56] * 0.25 + rows[:, None, None] * 1e-3
        g_rows = (torch.arange(B) + off)[:, None] * S + torch.arange(S)[None, :]        # global sequence row
        if noise is not None:                                                           # injected [T-1, B*S, 56]
            nz = noise.permute(1, 0, 2).reshape(B, S, T - 1, -1)
        else:
            t = torch.arange(T - 1, dtype=torch.float32)[None, None, :, None]
            c = torch.arange(56, dtype=torch.float32)[None, None, None, :]
            nz = torch.sin(g_rows[:, :, None, None].float() * 12.9898 + t * 78.233 + c * 37.719 + seed) * 0.3
        pred = base[:, None] + nz
        tokens = (pred.abs().sum(-1) * 1000).long() % 512
        if S == 1:
            pred, tokens = pred[:, 0], tokens[:, 0]
        out = (torch.zeros(()), {}, pred)
        return out + (tokens,) if return_tokens else out


def _eval_inputs(B=7, T=12):
    g = torch.Generator().manual_seed(3)
    v_s, v_l, v_a = torch.randn(B, T, 56, generator=g), torch.randn(B, T, 56, generator=g), torch.randn(B, T, 768, generator=g)
    lens = [12, 9, 12, 5, 7, 12, 4][:B]
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    src = torch.cat([v_s, v_a], -1) * mask[..., None]
    loader = [(src, v_l * mask[..., None], lens, None, ["id%d" % j for j in range(B)])]
    return v_s, v_l, v_a, mask, loader


def _eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import dimx  # noqa: F401
    from dimx import dist as dd
    from dimx import x_engine_pt
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        dd.init_from_env("gloo")
    model = _StubListener()
    v_s, v_l, v_a, mask, loader = _eval_inputs()
    tok, pred = x_engine_pt.generate_sharded(model, v_s, v_l, v_a, mask)
    # pre-sharded inputs (every rank holds only its rows) must give the same gathered result
    lo, hi = dd.shard_bounds(v_s.shape[0], dd.rank(), dd.world_size())
    tok2, pred2 = x_engine_pt.generate_sharded(model, v_s[lo:hi], v_l[lo:hi], v_a[lo:hi], mask[lo:hi], pre_sharded=True)
    yt, yp, xs, ids = x_engine_pt.evaluate_test_epoch(model, loader, torch.device("cpu"), beam_size=4)
    yt3, yp3, _, _ = x_engine_pt.evaluate_test_epoch(model, loader, torch.device("cpu"), beam_size=3)   # looped samples
    q.put((rank, tok.numpy(), pred.numpy(), tok2.numpy(), pred2.numpy(), [np.asarray(a) for a in yp],
           [np.asarray(a) for a in yt], ids, [np.asarray(a) for a in yp3]))
    if world > 1:
        dist.destroy_process_group()


def _run_eval(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_sharded_generation_and_evaluation_equal_single_process_world2():
    import numpy as np
    single = _run_eval(1)[0]
    for r in _run_eval(2):
        assert np.array_equal(r[1], single[1]) and np.allclose(r[2], single[2], atol=0, rtol=0)      # generate_sharded
        assert np.array_equal(r[3], single[1]) and np.array_equal(r[4], single[2])                    # pre-sharded
        assert r[7] == single[7] and len(r[5]) == 7
        for a, b in zip(r[5], single[5]):                       # best-of-4 (batched samples): same selected prediction
            assert a.shape == b.shape and np.array_equal(a, b)
        for a, b in zip(r[6], single[6]):
            assert np.array_equal(a, b)
        for a, b in zip(r[8], single[8]):                       # best-of-3 (sample loop)
            assert np.array_equal(a, b)
    # the stub really depends on both offsets: dropping them changes the second shard
    m = _StubListener()
    v_s, v_l, v_a, mask, _ = _eval_inputs()
    full = m(v_s, v_l, v_a, mask)[2]
    part = m(v_s[4:], v_l[4:], v_a[4:], mask[4:])[2]
    good = m(v_s[4:], v_l[4:], v_a[4:], mask[4:], batch_row_offset=4, shard=(4, 7))[2]
    assert not torch.equal(part, full[4:]) and torch.equal(good, full[4:])


# ---------------------------------------------------------------------------------------------------------------
# evaluate_test_epoch over the loaders get_vico_dataloaders hands out.  The evaluation loop expects every
# rank to see the SAME full batch (it splits the rows itself and all-gathers the winners); round 3 wrapped 'valid' in a
# DistributedSampler, which paired gathered predictions with another rank's targets.  World size 2 must return exactly
# the single-process lists -- targets, ids and the selected prediction of every clip.
# ---------------------------------------------------------------------------------------------------------------
def _loader_eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    import dimx  # noqa: F401
    from dimx import dist as dd
    from dimx import x_engine_pt
    from dimx.dataset.data_loader import get_vico_dataloaders
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank))
        dd.init_from_env("gloo")
    loaders = get_vico_dataloaders(batch_size=3, synthetic={"n_clips": 7, "max_len": 20, "min_len": 6, "seed": 11})
    sharded = isinstance(loaders["train"].sampler, torch.utils.data.distributed.DistributedSampler)
    eval_sharded = isinstance(loaders["valid"].sampler, torch.utils.data.distributed.DistributedSampler)
    n_train = sum(b[0].shape[0] for b in loaders["train"])
    yt, yp, xs, ids = x_engine_pt.evaluate_test_epoch(_StubListener(), loaders["valid"], torch.device("cpu"), beam_size=4)
    q.put((rank, sharded, eval_sharded, n_train, [np.asarray(a) for a in yt], [np.asarray(a) for a in yp], list(ids)))
    if world > 1:
        dist.destroy_process_group()


def _run_loader_eval(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_loader_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_evaluate_test_epoch_over_the_valid_loader_is_rank_invariant_world2():
    import numpy as np
    single = _run_loader_eval(1)[0]
    assert not single[1] and not single[2] and single[3] == 7
    both = _run_loader_eval(2)
    assert sum(r[3] for r in both) == 8          # the TRAINING loader is sharded (7 clips -> 4 + 4 with one wrap-around)
    for r in both:
        assert r[1] and not r[2]                 # 'train' sharded, 'valid' not
        assert r[6] == single[6] and len(r[5]) == 7
        for a, b in zip(r[4], single[4]):
            assert np.array_equal(a, b)          # targets in the single-process order
        for a, b in zip(r[5], single[5]):
            assert a.shape == b.shape and np.array_equal(a, b)     # the same winner for every clip
