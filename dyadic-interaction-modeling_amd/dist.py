"""Multi-GPU plumbing: one process per GPU, batch sharded by contiguous rows, weights replicated, and ONE
collective per evaluation batch -- an all-gather of the generated code indices (and optionally the decoded
coefficients, packed into the same buffer) so that any rank can run the host-side metrics.
Shard sizes follow from ``shard_bounds`` on every rank, so no count exchange and no host synchronisation precede the
payload collective (the round-4 form all-gathered the row counts and ``.item()``-ed them first);
``all_gather_rows`` without ``counts`` keeps that exchange for callers whose shards are genuinely unknown.

``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI inside a node); the same code runs on the
"gloo" backend for the CPU tests.  Payloads are 150 KB - 2.5 MB per rank: latency-bound, so a single
un-bucketed all_gather_into_tensor is the right shape (no ring tuning, no overlap machinery).
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def rank():
    return dist.get_rank() if is_initialized() else 0


def world_size():
    return dist.get_world_size() if is_initialized() else 1


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return 0, 1, local
    if not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size(), local


def shard_bounds(n, r, w):
    """Contiguous shard [lo, hi) of n rows for rank r of w (first n % w ranks get one extra row)."""
    base, extra = divmod(n, w)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def all_gather_counts(n_rows, device):
    """[rows held by rank 0, rank 1, ...]."""
    if world_size() == 1:
        return [int(n_rows)]
    n = torch.tensor([int(n_rows)], device=device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world_size())]
    dist.all_gather(counts, n)
    return [int(c.item()) for c in counts]


def _via_host(t):
    """gloo moves host memory: a GPU tensor on a gloo group (CPU tests with a GPU present, bench.py's one-GPU rehearsal) takes the
    collective on the host and comes back; on RCCL ("nccl") tensors stay where they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def shard_counts(n, w=None):
    """rows of every rank's contiguous shard of n rows (what shard_bounds gives each of them): known without communication."""
    w = world_size() if w is None else w
    return [shard_bounds(n, r, w)[1] - shard_bounds(n, r, w)[0] for r in range(w)]


def pack_rows(*tensors):
    """[n, ...] tensors of 4-byte element types -> one [n, sum of row sizes] int32 buffer (bit patterns), so that several
    per-row results travel in ONE collective; ``unpack_rows`` undoes it."""
    n = tensors[0].shape[0]
    parts = []
    for t in tensors:
        assert t.shape[0] == n and t.element_size() == 4, "pack_rows: 4-byte element types, equal row counts"
        k = 1
        for d in t.shape[1:]:
            k *= d
        parts.append(t.contiguous().view(torch.int32).reshape(n, k))   # explicit row size: an empty shard has no "-1"
    return torch.cat(parts, 1) if len(parts) > 1 else parts[0]


def unpack_rows(buf, like):
    """inverse of pack_rows: ``like`` = [(shape of one row, dtype), ...]"""
    out, c = [], 0
    for shape, dtype in like:
        k = 1
        for d in shape:
            k *= d
        out.append(buf[:, c:c + k].contiguous().view(dtype).reshape((buf.shape[0],) + tuple(shape)))
        c += k
    return out


def all_gather_rows(t, counts=None):
    """Concatenate every rank's rows along dim 0 (shards may differ in length, empty shards included).  ``counts``: the rows of
    every rank's shard when the caller knows them (``shard_counts``): then this is exactly one collective and no host sync."""
    if world_size() == 1:
        return t
    if _via_host(t):
        return all_gather_rows(t.cpu(), counts).to(t.device)
    t = t.contiguous()
    w = world_size()
    if counts is None:
        counts = all_gather_counts(t.shape[0], t.device)
    assert len(counts) == w and counts[rank()] == t.shape[0], "all_gather_rows: counts do not describe this rank's shard"
    if len(set(counts)) == 1:
        out = torch.empty((w * counts[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    out = torch.empty((w * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[i * mx:i * mx + c] for i, c in enumerate(counts)], 0)


def max_over_ranks(x: float, device=None) -> float:
    if world_size() == 1:
        return x
    if device is not None and torch.device(device).type == "cuda" and dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()


def assert_same_batch_count(n_batches, device=None):
    """A collective per batch needs the same number of batches on every rank."""
    if world_size() == 1:
        return
    t = torch.tensor([n_batches, -n_batches], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError("ranks see different batch counts (%d .. %d): shard the loader with a DistributedSampler "
                           "(dimx.dataset.data_loader.get_vico_dataloaders does)" % (-int(t[1]), int(t[0])))
