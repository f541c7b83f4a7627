#!/usr/bin/env python
"""bench.py -- DIM-Listener clips/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path over one batch of synthetic dyad clips already resident in HBM:
listener VQ encode -> speaker encoder stack + context + cross-K/V -> T-1 autoregressive decoder steps
(KV cache, top-k sampler) -> VQ decode -> continuous loss, i.e. ``SLMFT.forward(mode='val')``; with N > 1
every rank does that on its own 256 clips (weak scaling, weights replicated) and the generated code
indices are all-gathered (RCCL over xGMI) inside the timed region.

N = 1 workload = BASELINE config C3 (B=256, T=300, autoregressive decode, bf16 perf mode).
Extra objects on the line: ``roofline`` (dominant kernel, timed live with HIP events) and ``cpu_baseline``
(the CPU oracle on a bounded sample of the same workload, rank 0 at N=1 only); beside them ``parity_mode`` (the same
workload in the f32 mode), ``cross_attn_mfma`` (the K/V projection GEMM) and ``train_step`` (the HIP training step of
SURVEY 8 row f3, B=16) -- all measured outside the timed region of ``value``.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--frames 300] [--mode bf16|f32]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dimx  # noqa: E402,F401
from dimx import dist as ddist  # noqa: E402
from dimx import lib as L  # noqa: E402
from dimx import prng, weights  # noqa: E402
from dimx.seq2seq_pretrain import SLMFT, mark_prefix  # noqa: E402

SEED = 20260928
GFLOP_PER_CLIP_T300 = 73.5  # SURVEY.md section 8d, necessary work


def synth_batch(B, T, device, salt):
    v_s = torch.from_numpy(prng.normal(SEED + salt, "bench.v_speaker", (B, T, 56))).to(device)
    v_l = torch.from_numpy(prng.normal(SEED + salt, "bench.v_listener", (B, T, 56))).to(device)
    v_a = torch.from_numpy(prng.normal(SEED + salt, "bench.v_audio", (B, T, 768))).to(device)
    mask = mark_prefix(torch.ones(B, T, dtype=torch.bool, device=device))   # full clips: a prefix mask, like the engine protocol's
    return v_s, v_l, v_a, mask


def _cpu_worker(T, b):
    """(subprocess) BASELINE.md section 3 protocol on the GPU box's host cores: the CPU oracle on b clips, 2 warm-ups +
    median of 5 timed runs of (a) the necessary-work variant (one listener VQ encode per clip), 1 warm-up + median of 3 of
    (b) the reference-faithful variant (+3 redundant VQ encodes per clip, code/seq2seq_pretrain.py:497-500), at the fastest
    thread count of a short probe (on a 2x64-core host the tiny per-step matmuls of the AR loop get slower with every
    extra thread), plus one single-thread run on 2 clips.  Bounded to about half a minute of CPU work."""
    from oracle import ref_cpu
    torch.set_grad_enabled(False)
    sd = weights.synth_state_dict(weights.slmft_spec(), SEED)
    v_s = torch.from_numpy(prng.normal(SEED, "bench.v_speaker", (b, T, 56)))
    v_l = torch.from_numpy(prng.normal(SEED, "bench.v_listener", (b, T, 56)))
    v_a = torch.from_numpy(prng.normal(SEED, "bench.v_audio", (b, T, 768)))
    mask = torch.ones(b, T, dtype=torch.bool)
    noise = torch.from_numpy(prng.exponential(SEED, "bench.noise", (T - 1, b, 512)))
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (4, 8, 16, 32, 64, ncpu // 2, ncpu) if 1 <= c <= ncpu})
    probe_T = 24
    best_thr, best_t = cands[0], float("inf")
    for thr in cands:
        torch.set_num_threads(thr)
        t0 = time.perf_counter()
        ref_cpu.slmft_forward(sd, v_s[:, :probe_T], v_l[:, :probe_T], v_a[:, :probe_T], mask[:, :probe_T], "val",
                              noise=noise[:probe_T - 1])
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_thr, best_t = thr, dt
        if dt > 4 * best_t:
            break

    def necessary(n=b):
        ref_cpu.slmft_forward(sd, v_s[:n], v_l[:n], v_a[:n], mask[:n], "val", noise=noise[:, :n])

    def faithful():
        necessary()
        # the reference's forward_vq encodes speaker AND listener, and forward() calls it twice (:497-500)
        ref_cpu.forward_vq(sd, v_s, v_l, mask, with_speaker=True)
        ref_cpu.forward_vq(sd, v_s, v_l, mask, with_speaker=False)

    def median_of(fn, runs, warmups):
        for _ in range(warmups):
            fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], ts

    torch.set_num_threads(best_thr)
    t_a, runs_a = median_of(necessary, 5, 2)
    t_b, runs_b = median_of(faithful, 3, 1)
    torch.set_num_threads(1)
    n1 = min(2, b)
    t0 = time.perf_counter()
    necessary(n1)
    t_1 = time.perf_counter() - t0
    print("CPU_BASELINE " + json.dumps({
        "value": b / t_a, "unit": "clips/s", "cores": best_thr, "kind": "port",
        "sample": "%d clips x T=%d through oracle/ref_cpu.slmft_forward(mode='val') (torch CPU fp32; %d of %d host threads = "
                  "fastest in a T=%d probe); necessary-work variant, 2 warm-ups + median of 5 runs (%s s)"
                  % (b, T, best_thr, ncpu, probe_T, ", ".join("%.2f" % t for t in runs_a)),
        "reference_faithful": {"value": b / t_b, "unit": "clips/s",
                               "what": "+3 redundant VQ encodes per clip as code/seq2seq_pretrain.py:497-500, 1 warm-up + "
                                       "median of 3 runs (%s s)" % ", ".join("%.2f" % t for t in runs_b)},
        "single_thread": {"value": n1 / t_1, "unit": "clips/s", "cores": 1, "sample": "%d clips, one run of %.1f s" % (n1, t_1)}}))


def cpu_baseline(T, timeout_s=300):
    """Run the CPU oracle on a bounded sample in a subprocess (hard timeout: the bench never hangs on it)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(T), "8"],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("CPU_BASELINE "):
                return json.loads(line[len("CPU_BASELINE "):])
        return {"value": None, "unit": "clips/s", "cores": 0, "kind": "port",
                "sample": "cpu worker failed: " + (r.stderr or "")[-200:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "clips/s", "cores": 0, "kind": "port",
                "sample": "cpu worker exceeded %d s on 8 clips x T=%d" % (timeout_s, T)}


def train_step_line(device, T, batches=(16, 4, 64), warm=3, steps=8):
    """SURVEY 8 row f3 next to the headline: one optimisation step of SLMFT (forward + backward + clip 1.0 + AdamW, the
    reference's train_epoch body, code/x_engine_pt.py:9-60) on the hand-written HIP training step, bf16 operands with f32
    accumulation and f32 master weights; B clips of T frames (B = 16 is the line's own figure, B = 4 is the reference's ViCo
    batch, code/finetune_s2s_pretrain.py:121, B = 64 shows the MFMA-bound end), listener codes from the frozen VQ-VAE
    precomputed (they do not depend on the trained weights).  Reported beside the metric, never part of ``value``."""
    from dimx.train_hip import HipTrainer
    torch.cuda.empty_cache()
    m = SLMFT(synthetic_seed=SEED, numeric_mode=L.MODE_PERF_BF16).to(device)
    tr = HipTrainer(m, lr=1e-5, clip=1.0)
    out = None
    by_batch = {}
    for B in batches:
        v_s, v_l, v_a, mask = synth_batch(B, T, device, salt=7)
        _, z = m.forward_vq(v_s, v_l, mask, with_speaker=False)
        for _ in range(warm):
            tr.train_step(v_s, v_l, v_a, mask, kv_mask=False, z_l=z)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.train_step(v_s, v_l, v_a, mask, kv_mask=False, z_l=z)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        assert torch.isfinite(loss)
        g = tr.graph_stats()
        by_batch[str(B)] = {"ms_per_step": dt * 1e3, "clips_per_s": B / dt, "graph_nodes": g[2]}
        if out is None:
            out = {"value": B / dt, "unit": "clips/s", "ms_per_step": dt * 1e3, "batch": B, "frames": T, "dtype": "bf16",
                   "steps": steps, "warmup": warm,
                   "note": "SLMFT training step (forward + backward + clip + AdamW) on the HIP kernels of csrc/train*.hip; "
                           "PyTorch autograd on rocBLAS for the same step: tools/bench_train.py"}
    out["by_batch"] = by_batch
    g = tr.graph_stats()
    out["hip_graph"] = {"steps_replayed": g[0], "steps_kernel_by_kernel": g[1]}
    del tr, m
    torch.cuda.empty_cache()
    out["other_models"] = other_train_steps(device, T, warm=2, steps=4)
    return out


def other_train_steps(device, T, B=16, warm=2, steps=4):
    """The two other training loops of the reference on their HIP steps: SLM pre-training
    (code/train_s2s_pretrain.py:41-64 over SLM.forward, code/seq2seq_pretrain.py:300-323: three encoders, InfoNCE, the decoder for
    both streams, both trainable VQ-VAE decoders) and the legacy ListenerGenerator (code/x_engine.py:8-36; its step includes the
    frozen speaker VQ-VAE encoder on the inference engine).  bf16 operands, f32 accumulation and master weights, B clips of T frames."""
    from dimx import train as Tr
    from dimx.seq2seq import ListenerGenerator
    from dimx.seq2seq_pretrain import SLM
    from dimx.train_hip import LegacyHipTrainer, SlmHipTrainer
    res = {}

    def timed(fn):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            last = fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / steps, last
    v_s, v_l, v_a, mask = synth_batch(B, T, device, salt=9)
    m = SLM(synthetic_seed=SEED, numeric_mode=L.MODE_PERF_BF16).to(device)
    Tr.set_slm_trainable(m)
    m.train()
    with torch.no_grad():
        z_s, z_l = m.forward_vq(v_s, v_l, mask)
    ms_, ml_ = m.random_masking_unstructured(v_s, mask, 0.15), m.random_masking_unstructured(v_l, mask, 0.15)
    tr = SlmHipTrainer(m, lr=1e-5, clip=1.0)
    dt, last = timed(lambda: tr.train_step(v_s, v_l, v_a, mask, mask_speaker=ms_, mask_listener=ml_, z_s=z_s, z_l=z_l)[0])
    assert torch.isfinite(last)
    res["slm_pretraining"] = {"ms_per_step": dt * 1e3, "clips_per_s": B / dt, "batch": B, "frames": T,
                              "note": "SlmHipTrainer: dimx_train_slm_forward_backward + dimx_train_adamw"}
    del tr, m
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(5)
    v_s824 = torch.randn(B, T, 824, generator=g).to(device)
    lid = (torch.arange(B) % 100).to(device)
    m = ListenerGenerator(numeric_mode=L.MODE_PERF_BF16).to(device)
    Tr.set_legacy_trainable(m)
    m.train()
    tr = LegacyHipTrainer(m, lr=1e-5, clip=1.0)
    dt, last = timed(lambda: tr.train_step(v_s824, v_l, mask, listener_ids=lid)[0])
    assert torch.isfinite(last)
    res["legacy_generator"] = {"ms_per_step": dt * 1e3, "clips_per_s": B / dt, "batch": B, "frames": T,
                               "note": "LegacyHipTrainer: dimx_train_legacy_forward_backward + dimx_train_adamw; includes the frozen "
                                       "speaker VQ-VAE encoder + listener VQ encode of the step's inputs"}
    del tr, m
    torch.cuda.empty_cache()
    return res


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_ranks(args, argv):
    """``--gpus N`` must mean N ranks, one per GPU -- or a non-zero exit, never a line that says ``n_gpus: 1``.

    * under torchrun (WORLD_SIZE set): WORLD_SIZE must equal N, else exit 2;
    * plain ``python bench.py --gpus N`` with N > 1: this process becomes the launcher -- it re-executes itself as
      ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
      bench.py <same arguments>`` (one process per GPU, HSA_ENABLE_IPC_MODE_LEGACY=0), after checking that N GPUs are
      visible (exit 2 with a message otherwise)."""
    want = max(1, args.gpus)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != want:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d (refusing to report "
                             "a %s-rank run as %d GPUs)" % (want, env_world, want, env_world, want))
        return
    if want == 1:
        return
    if not args.stub and not args.rehearse_shared_gpu:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < want:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible on this node" % (want, have))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(want),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)      # the launcher's exit code is torchrun's


class _StubModel:
    """LAUNCHER SELF-TEST ONLY (``--stub``, tests/test_bench_launcher.py): a CPU stand-in with SLMFT's call shape whose
    tokens depend on the clip's content, the seed and the GLOBAL row (shard window), so that the rank plumbing -- launch,
    sharding, all-gather, digest, the JSON line -- runs on gloo without a GPU.  Its line says so and is not a measurement."""

    def __call__(self, v_s, v_l, v_a, mask, mode="val", seed=0, return_tokens=True, n_samples=1, shard=None,
                 batch_row_offset=0, **kw):
        B, T = mask.shape
        lo = shard[0] if shard else 0
        rows = torch.arange(lo, lo + B)[:, None]
        t = torch.arange(T - 1)[None, :]
        tokens = (rows * 7919 + t * 104729 + int(seed) * 31 + (v_s[:, :T - 1, 0] * 1000).long()
                  + (batch_row_offset - lo) * 17) % 512
        pred = (tokens[..., None].float() * 1e-3).expand(B, T - 1, 56).contiguous()
        return torch.zeros(()), {}, pred, tokens


def shard_check(make_model, device, rank, world, T=60, per_rank=8):
    """The SAME seeded global batch of G = per_rank * N clips generated by N ranks (rank r: rows [r G/N, (r+1) G/N) with
    its sampler window and batch_row_offset) and all-gathered must equal what ONE rank generates for all G
    clips, bit for bit (f32 parity mode): sha256 of both on the line (what tools/scale_check.py computes across
    separate runs, here inside the one run the driver launches)."""
    import hashlib
    G = per_rank * world
    lo = rank * per_rank
    def clips(name, c):
        return torch.from_numpy(prng.normal(SEED, "bench.shard." + name, (G, T, c)))
    v_s, v_l, v_a = clips("v_speaker", 56), clips("v_listener", 56), clips("v_audio", 768)
    model = make_model()
    def run(a, b):
        m = torch.ones(b - a, T, dtype=torch.bool, device=device)
        m = mark_prefix(m) if device.type == "cuda" else m
        _, _, pred, tok = model(v_s[a:b].to(device), v_l[a:b].to(device), v_a[a:b].to(device), m, mode="val",
                                seed=SEED + 5, return_tokens=True, shard=(a, G), batch_row_offset=a)
        return tok.reshape(b - a, T - 1).to(torch.int32), pred.reshape(b - a, -1).float().contiguous()
    tok, pred = run(lo, lo + per_rank)
    # one collective, shard sizes known on every rank (per_rank rows each): no count exchange, no host sync
    buf = ddist.all_gather_rows(ddist.pack_rows(tok, pred), [per_rank] * world)
    all_tok, all_pred = ddist.unpack_rows(buf, [((T - 1,), torch.int32), ((pred.shape[1],), torch.float32)])
    out = None
    if rank == 0:
        one_tok, one_pred = run(0, G)
        d_sh = hashlib.sha256(all_tok.cpu().numpy().tobytes()).hexdigest()
        d_one = hashlib.sha256(one_tok.cpu().numpy().tobytes()).hexdigest()
        out = {"global_batch": G, "frames": T, "dtype": "f32", "tokens_sha256_sharded": d_sh, "tokens_sha256_one_rank": d_one,
               "identical": d_sh == d_one, "max_abs_pred_diff": float((all_pred - one_pred).abs().max())}
    ddist.barrier()
    return out


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        _cpu_worker(int(sys.argv[2]), int(sys.argv[3]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--mode", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--samples", type=int, default=1, help="generations per clip in one pass (best-of-N protocol)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-mode-compare", action="store_true", help="skip the bf16-vs-f32 motion statistics (part of the parity-mode leg)")
    ap.add_argument("--no-parity-mode", action="store_true")
    ap.add_argument("--no-train-step", action="store_true")
    ap.add_argument("--no-shard-check", action="store_true")
    ap.add_argument("--no-best-of-n", action="store_true", help="skip the best-of-10 leg (10 samples per clip in one pass)")
    ap.add_argument("--gather", default="tokens+coeffs", choices=["tokens", "tokens+coeffs"],
                    help="payload of the timed step's ONE all-gather (N > 1): the generated code indices (north_star's wording), or the "
                         "indices + the decoded coefficients in one packed buffer -- what x_engine_pt.evaluate_test_epoch gathers for the "
                         "metrics rank (17 MB per rank at C4 against 0.3 MB); the default times the real payload")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # launcher self-test on CPU / gloo, no kernels
    # rehearsal of the N-rank path with the REAL model on a 1-GPU box: every rank on cuda:0, collectives on gloo (RCCL refuses
    # two ranks on one device).  Its line says so and is not a measurement (tools/scale_check.sh rehearse).
    ap.add_argument("--rehearse-shared-gpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    ensure_ranks(args, sys.argv[1:])          # N > 1 from plain python: re-executes under torch.distributed.run
    rank, world, local = ddist.init_from_env("gloo" if (args.stub or args.rehearse_shared_gpu) else None)
    if args.rehearse_shared_gpu:
        local = 0
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: %d rank(s) initialised for --gpus %d" % (world, args.gpus))
    if args.stub:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    torch.set_grad_enabled(False)
    B, T = args.batch, args.frames
    mode = L.MODE_PERF_BF16 if args.mode == "bf16" else L.MODE_PARITY_F32

    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    if args.stub:
        model, eng = _StubModel(), None
        g = torch.Generator().manual_seed(SEED + rank)
        v_s, v_l, v_a = (torch.randn(B, T, c, generator=g) for c in (56, 56, 8))
        mask = torch.ones(B, T, dtype=torch.bool)
    else:
        model = SLMFT(synthetic_seed=SEED, numeric_mode=mode).eval()
        v_s, v_l, v_a, mask = synth_batch(B, T, device, salt=rank)
        eng = model.engine(device)

    def step(i):
        _, _, pred, tokens = model(v_s, v_l, v_a, mask, mode="val", seed=SEED + i + 1, return_tokens=True,
                                   n_samples=args.samples)
        tok = tokens.reshape(-1, T - 1).to(torch.int32)
        # equal shards: ONE collective, nothing synchronises with the host first
        if args.gather == "tokens":
            return pred, ddist.all_gather_rows(tok, [tok.shape[0]] * world)
        # the payload evaluate_test_epoch gathers (code/x_engine_pt.py:259-270 accumulates exactly these on the host): indices +
        # decoded coefficients, packed into one buffer (dist.pack_rows)
        buf = ddist.all_gather_rows(ddist.pack_rows(tok, pred.reshape(tok.shape[0], -1).float()), [tok.shape[0]] * world)
        return pred, buf[:, :T - 1]      # the indices' columns of the gathered buffer (the metrics rank unpacks the rest)

    for i in range(args.warmup):
        step(i)
    ddist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        pred, gathered = step(args.warmup + i)
    sync()
    ddist.barrier()
    elapsed = ddist.max_over_ranks(time.perf_counter() - t0, device if (world > 1 and device.type == "cuda") else None)
    assert gathered.shape == (world * B * args.samples, T - 1) and torch.isfinite(pred).all()

    rccl_ranks = 1
    check = None
    if world > 1:
        import torch.distributed as tdist
        one = torch.ones(1, device="cpu" if tdist.get_backend() == "gloo" else device)
        tdist.all_reduce(one)                 # the collective library (RCCL on GPUs) saw this many ranks
        rccl_ranks = int(one.item())
        if not args.no_shard_check:
            check = shard_check((lambda: _StubModel()) if args.stub else
                                (lambda: SLMFT(synthetic_seed=SEED, numeric_mode=L.MODE_PARITY_F32).eval()),
                                device, rank, world)
        tdist.barrier()
        tdist.destroy_process_group()
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    clips_s = world * B * args.samples * args.steps / elapsed
    names = {(256, 300): "C3", (64, 1500): "C5 per-GPU shard"}
    if world > 1 and (B, T) == (256, 300):
        tag = "C4 layout (256 clips per GPU x %d GPUs)" % world
    else:
        tag = names.get((B, T), "custom")
    out = {
        "metric": "listener clips/sec (T=%d, EMOCA-56)" % T + ("" if args.samples == 1 else " x %d samples per clip in one pass" % args.samples),
        "value": clips_s, "unit": "clips/s" if args.samples == 1 else "generated sequences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": "%s: B=%d/GPU synthetic dyad clips, T=%d, SLMFT.forward(mode='val'): listener VQ "
                               "encode + encoder + %d-step AR decode (top-k 52 sampling) + VQ decode"
                               % (tag, B, T, T - 1),
                   "global_batch": world * B, "seq_len": T, "parallelism": "dp%d (clips sharded, ONE all-gather of %s per batch)"
                   % (world, "code indices" if args.gather == "tokens" else "code indices + decoded coefficients"),
                   "gather": args.gather,
                   "gather_bytes_per_rank": B * args.samples * (T - 1) * (4 + (56 * 4 if args.gather != "tokens" else 0))},
        "achieved_tflops_necessary_work": clips_s * GFLOP_PER_CLIP_T300 * (T / 300.0) / 1e3,
        "rccl_ranks": rccl_ranks,
    }
    if args.stub:
        out["metric"] = "LAUNCHER SELF-TEST (--stub: CPU stand-in on gloo, no kernels) -- not a measurement"
        out["data"] = "stub"
    if args.rehearse_shared_gpu:
        out["metric"] = "REHEARSAL (%d ranks sharing ONE GPU, collectives on gloo) -- not a measurement" % world
        out["data"] = "synthetic (rehearsal)"
    if check is not None:
        out["shard_check"] = check
    if rccl_ranks != world or (check is not None and not check["identical"]):
        print(json.dumps(out))
        raise SystemExit("bench.py: collective saw %d of %d ranks / sharded generation %s the one-rank result"
                         % (rccl_ranks, world, "equals" if (check is None or check["identical"]) else "DIFFERS from"))
    if args.stub:
        print(json.dumps(out))
        return
    # batches dimx_generate had to regenerate because its XCD-local chain kernels / deferred LayerNorm reported a fault (0 expected)
    out["chain_faults"] = int(eng.chain_faults())
    if world == 1 and args.mode == "bf16" and args.samples == 1 and not args.no_best_of_n and B * 10 <= 4096:
        # the reference's real test-time protocol draws 10 generations per clip (code/x_engine_pt.py:257): the same clips as ONE pass
        # of B x 10 sequences (shared context K/V per clip), 1 warm-up + 2 timed passes
        model(v_s, v_l, v_a, mask, mode="val", seed=SEED + 900, n_samples=10)
        sync()
        t1 = time.perf_counter()
        for i in range(2):
            model(v_s, v_l, v_a, mask, mode="val", seed=SEED + 901 + i, n_samples=10)
        sync()
        dt10 = (time.perf_counter() - t1) / 2
        out["best_of_10"] = {"value": B * 10 / dt10, "unit": "generated sequences/s", "ms_per_pass": dt10 * 1e3, "clips": B, "samples_per_clip": 10,
                             "note": "SLMFT.forward(mode='val', n_samples=10): the reference's best-of-10 evaluation draws in one pass "
                                     "(python bench.py --samples 10 times the same with the --steps / --warmup of the line)"}
    if world == 1 and args.mode == "bf16" and not args.no_parity_mode and args.samples == 1:
        # the same workload in the mode that meets north_star's tolerance (f32 operands, exact-f32 MFMA; VQ indices
        # and generated tokens bit-identical to the oracle): 2 warm-ups + 5 timed steps
        pm = SLMFT(synthetic_seed=SEED, numeric_mode=L.MODE_PARITY_F32).eval()
        if not args.no_mode_compare:
            # what the bf16 mode costs in the quantities the reference reports (per-clip FD / MSE / variance / STS of the generated
            # motion): same clips, same sampler seed, both modes -- against the spread two sampler seeds give in the f32 mode
            from dimx.mode_compare import compare_modes
            out["bf16_vs_f32"] = compare_modes(model, pm, v_s, v_l, v_a, mask, seed=SEED + 777)
        del model, eng
        torch.cuda.empty_cache()
        PM_WARM, PM_STEPS = 2, 5
        for i in range(PM_WARM):
            pm(v_s, v_l, v_a, mask, mode="val", seed=SEED + i)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(PM_STEPS):
            pm(v_s, v_l, v_a, mask, mode="val", seed=SEED + 100 + i)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / PM_STEPS
        out["parity_mode"] = {"value": B / dt, "unit": "clips/s", "ms_per_step": dt * 1e3, "dtype": "f32", "steps": PM_STEPS,
                              "warmup": PM_WARM, "note": "same workload in DIMX_MODE_PARITY_F32 (the mode the oracle parity "
                              "tests run in: indices bit-exact, coefficients <= 1e-4); round 6: its decode-step GEMMs run on the bf16 "
                              "matrix cores as exact three-plane splits of both operands with f32 accumulation (csrc/gemm_x3.hip, "
                              "f32-equivalent; DIMX_NO_X3=1 = the exact-f32 MFMA kernel)",
                              "decode_gemm": "gemm_x3_kernel" if os.environ.get("DIMX_NO_X3") is None else "gemm_ws_kernel<float>"}
        eng = pm.engine(device)
    if not args.no_roofline:   # per-GPU kernel measurement on rank 0's device (for N > 1 the other ranks have left by now)
        from dimx import roofline
        out["roofline"] = roofline.dominant_kernel(eng, B, T, args.mode)
        if args.mode == "bf16" and "parity_mode" in out and B <= 256:
            # the parity mode's decode GEMMs (round 6: split-bf16 on the bf16 matrix cores, f32-equivalent) next to the bf16 mode's kernels
            out["roofline"]["others"] += roofline.decode_gemm_x3(B, device)
        out["cross_attn_mfma"] = roofline.cross_kv_gemm(B, T, args.mode, device)
        if args.mode == "bf16":
            out["cross_attn_bundle"] = roofline.cross_attn_bundle(B, T, args.mode, device, out["roofline"], out["cross_attn_mfma"])
    if world == 1 and args.mode == "bf16" and not args.no_train_step and args.samples == 1:
        out["train_step"] = train_step_line(device, T)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(T)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
