"""CPU, world_size 8 (gloo): the EXACT shardings of BASELINE configs C4 and C5 -- 2048 clips -> 8 x 256 with batch_row_offset
0, 256, ..., 1792 and the sampler window (lo, 2048); 512 clips -> 8 x 64 -- through `x_engine_pt.generate_sharded` and through
`bench.py --gpus 8 --stub` (the widest process group tested so far was 2, and the 8-way bounds were only
ever computed at N = 2 or by hand-set offsets).  A CPU stub stands in for SLMFT: its output depends on the clip content, on the
positional row (batch_row_offset) and on the global sequence row (shard), and it records what every rank was handed."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stub(torch.nn.Module):
    """tokens / coefficients depend on content + positional row + global row; `seen` = what the rank was asked for"""

    def __init__(self):
        super().__init__()
        self.seen = []

    def forward(self, v_speaker, v_listener, v_audio, mask, mode="val", batch_row_offset=0, shard=None, return_tokens=False, **kw):
        B, T, _ = v_speaker.shape
        off, total = shard if shard is not None else (0, B)
        self.seen.append((B, int(batch_row_offset), int(off), int(total)))
        pos = (torch.arange(B, dtype=torch.float32) + batch_row_offset)[:, None, None] * 1e-3      # VQ decoder's pe[b] quirk
        grow = (torch.arange(B) + off)[:, None, None].float()                                       # sampler's global row
        t = torch.arange(T - 1, dtype=torch.float32)[None, :, None]
        c = torch.arange(56, dtype=torch.float32)[None, None, :]
        pred = v_speaker[:, 1:, :] * 0.5 + v_audio[:, 1:, :56] * 0.25 + pos + torch.sin(grow * 12.9898 + t * 78.233 + c * 37.719) * 0.3
        tokens = (pred.abs().sum(-1) * 1000).long() % 512
        out = (torch.zeros(()), {}, pred)
        return out + (tokens,) if return_tokens else out


def _inputs(B, T=6):
    g = torch.Generator().manual_seed(B)
    v_s, v_l, v_a = (torch.randn(B, T, 56, generator=g) for _ in range(3))
    return v_s, v_l, v_a, torch.ones(B, T, dtype=torch.bool)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import dimx  # noqa: F401
    from dimx import dist as dd
    from dimx import x_engine_pt
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        dd.init_from_env("gloo")
    out = {}
    for name, B in (("C4", 2048), ("C5", 512)):
        m = _Stub()
        v_s, v_l, v_a, mask = _inputs(B)
        calls = []
        real = dd.all_gather_counts
        dd.all_gather_counts = lambda *a, **k: calls.append(1) or real(*a, **k)     # the count exchange must not run
        tok, pred = x_engine_pt.generate_sharded(m, v_s, v_l, v_a, mask)
        dd.all_gather_counts = real
        out[name] = (m.seen, tok.numpy(), pred.numpy(), len(calls), dd.shard_counts(B) if world > 1 else [B])
    q.put((rank, out))
    if world > 1:
        dist.destroy_process_group()


def _run(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_c4_and_c5_shardings_at_world_size_8_equal_the_single_process_batch():
    import numpy as np
    single = _run(1)[0][1]
    res = _run(WORLD)
    assert [r for r, _ in res] == list(range(WORLD))
    for name, B in (("C4", 2048), ("C5", 512)):
        per = B // WORLD                                        # 256 / 64 clips per GPU (BASELINE.json configs 4 and 5)
        for rank, out in res:
            seen, tok, pred, count_exchanges, counts = out[name]
            assert seen == [(per, rank * per, rank * per, B)], (name, rank, seen)     # rows, batch_row_offset, sampler window
            assert counts == [per] * WORLD and count_exchanges == 0
            assert tok.shape == (B, 5) and np.array_equal(tok, single[name][1])       # every rank holds the whole batch's result
            assert np.array_equal(pred, single[name][2])
    # the stub really depends on both offsets
    m = _Stub()
    v_s, v_l, v_a, mask = _inputs(512)
    full = m(v_s, v_l, v_a, mask)[2]
    assert not torch.equal(m(v_s[448:], v_l[448:], v_a[448:], mask[448:])[2], full[448:])
    assert torch.equal(m(v_s[448:], v_l[448:], v_a[448:], mask[448:], batch_row_offset=448, shard=(448, 512))[2], full[448:])


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(rows[-1]) if rows else None)


def test_bench_launcher_at_8_ranks_with_c4_and_c5_shard_sizes():
    """`python bench.py --gpus 8` as the driver launches it (plain python -> self-launch of 8 ranks), stub step on gloo: the C4
    layout (256 clips per rank, global batch 2048) and the C5 layout (64 per rank, 512), the default payload of the timed
    collective (code indices + decoded coefficients, what evaluate_test_epoch gathers) and the indices-only one."""
    r, d = _bench(["--gpus", "8", "--stub", "--steps", "2", "--warmup", "1", "--batch", "256", "--frames", "12"])
    assert r.returncode == 0 and d is not None, r.stdout[-500:] + r.stderr[-2000:]
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["config"]["global_batch"] == 2048 and d["scaling"] == "weak"
    assert d["config"]["gather"] == "tokens+coeffs" and d["config"]["gather_bytes_per_rank"] == 256 * 11 * (4 + 56 * 4)
    sc = d["shard_check"]
    assert sc["identical"] and sc["global_batch"] == 64 and sc["tokens_sha256_sharded"] == sc["tokens_sha256_one_rank"]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1
    r, d = _bench(["--gpus", "8", "--stub", "--steps", "1", "--warmup", "0", "--batch", "64", "--frames", "10", "--gather", "tokens"])
    assert r.returncode == 0 and d is not None, r.stdout[-500:] + r.stderr[-2000:]
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 512 and d["config"]["gather"] == "tokens"
    assert d["config"]["gather_bytes_per_rank"] == 64 * 9 * 4 and d["shard_check"]["identical"]
