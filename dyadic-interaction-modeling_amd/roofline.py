"""Live roofline measurement of the dominant kernels of the autoregressive decode loop, for bench.py.

The two kernels that dominate the rocprofv3 kernel trace of the C3 workload (profiles/) are launched here
exactly as dimx_generate launches them, on the current stream, bracketed by HIP events:

  * decode attention (cross-attention form): pure HBM stream of the [B,12,T,64] K and V caches.
    Algorithmic bytes per launch = B*12*T*64 * 2 (K,V) * elem_size (+ q and out, B*768*elem_size each);
    bound = HBM (8 TB/s peak).  The four decoder layers' caches are visited round-robin so the working set
    (4 x 236 MB at B=256,T=300, bf16) exceeds the 256 MB Infinity Cache like it does in the real loop.
  * decode GEMM (the 1152 <- 4608 feed-forward down projection, M = B rows, f32 residual epilogue).
    Algorithmic flops per launch = 2*B*1152*4608; bound = MFMA (2.5 PFLOP/s dense bf16, 157.3 TFLOP/s f32).

`traffic` (HBM bytes per launch from the TCC PMC counters) comes from profiles/pmc_decode_attn_<hash>.json, where
<hash> is the first 12 hex digits of the sha256 of csrc/decode_attn.hip: a PMC pass only counts for the kernel source
it was collected on (tools/pmc_record.py writes the file; .git does not travel to the GPU box, so the kernel source
hash -- not the commit id -- is what ties the two together; the commit is recorded inside the file).  No matching
file -> null.
"""
import hashlib
import json
import os

import torch

from . import engine as E

HERE = os.path.dirname(os.path.abspath(__file__))
PROFILES = os.path.join(os.path.dirname(HERE), "profiles")


DECODE_ATTN_SOURCES = ("decode_attn.hip", "decode_attn_body.hpp")   # the kernel body lives in the header since round 3
PREFILL_MFMA_SOURCES = ("mlp_fused.hip", "attention_tr.hip")         # tools/pmc_prefill_record.py
LAYER_CHAIN_SOURCES = ("chain.hip", "decode_attn_body.hpp")          # xcd_layer_kernel (round 5): the attention bodies + the chain phases


def kernel_source_hash(name=DECODE_ATTN_SOURCES):
    """sha256[:12] over the source file(s) a kernel is built from (a PMC record only counts for the sources it was taken on)."""
    names = (name,) if isinstance(name, str) else tuple(name)
    hsh = hashlib.sha256()
    for n in names:
        with open(os.path.join(HERE, "csrc", n), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:12]


def pmc_traffic(B, T, mode):
    """HBM bytes per launch of the cross-attention decode kernel from the PMC pass recorded for THIS kernel source."""
    path = os.path.join(PROFILES, "pmc_decode_attn_%s.json" % kernel_source_hash())
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if (rec.get("B"), rec.get("T"), rec.get("mode")) != (B, T, mode):
        return None
    return float(rec["traffic_bytes"])

def pmc_traffic_layer(B, T, mode):
    """HBM bytes per launch of the layer kernel (xcd_layer_kernel) from the PMC pass recorded for THIS source (tools/pmc_record.py)."""
    path = os.path.join(PROFILES, "pmc_layer_chain_%s.json" % kernel_source_hash(LAYER_CHAIN_SOURCES))
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    if (rec.get("B"), rec.get("T"), rec.get("mode")) != (B, T, mode):
        return None
    return float(rec["traffic_bytes"])


HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}


def _time_launches(fn, n_warm, n_iter):
    for i in range(n_warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_iter):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n_iter


def decode_attention(B, T, mode, device, iters=200):
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    es = 2 if mode == "bf16" else 4
    H, Tp = 12, (T + 7) // 8 * 8
    layers = 4
    kc = [torch.randn(B, H, Tp, 64, device=device).to(dt) for _ in range(layers)]
    vc = [torch.randn(B, H, Tp, 64, device=device).to(dt) for _ in range(layers)]
    q = torch.randn(B, H * 64, device=device)   # f32 projection slab, exactly what dimx_generate hands the kernel
    km = torch.ones(B, T, dtype=torch.uint8, device=device)   # the context mask dimx_generate passes
    # 40 warm-up launches: bench.py comes here straight from the f32 parity run, and the first launches after a change of
    # workload run at another clock (round 3: 39.7 us live with 8 warm-ups against 37.75 us in the kernel trace of the loop)
    sec = _time_launches(lambda i: E.op_decode_attn(q, kc[i % layers], vc[i % layers], T, 0.125, km), 40, iters)
    alg_bytes = B * H * T * 64 * 2 * es + 2 * B * H * 64 * es
    gbs = alg_bytes / sec / 1e9
    # HBM bytes per launch from the PMC pass recorded for this kernel source (FETCH_SIZE doubled for 16-B/lane
    # streaming reads on gfx950 per MI355X_MICROARCH.md + WRITE_SIZE), else null
    traffic = pmc_traffic(B, T, mode)
    pairs = B * H   # waves per (clip, head): launch_decode_attn's rule (csrc/decode_attn.hip)
    nsplit = (4 if pairs * 4 <= 3072 else 2 if pairs * 2 <= 3072 else 1) if mode == "bf16" else (4 if Tp >= 1024 else 2 if Tp >= 512 else 1)
    return {"kernel": "decode_attn_kernel<%s, false, true, %d> (cross-attention form, %d keys)" % ("dimx::bf16" if mode == "bf16" else "float", nsplit, T), "bound": "hbm",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": sec * 1e6}


def layer_chain(B, T, device, iters=120, n_self=None, prof=None):
    """The attention half of a decoder layer as dimx_generate launches it for 128 < B <= 256 clips in the bf16 mode (round 5:
    chain.hip xcd_layer_kernel = self attention -> out-projection -> cross-q -> cross attention -> out-projection, one XCD-local
    launch), at the MEAN self-attention cache fill of the loop (T / 2 keys), rotating over four layers' worth of caches and
    weights (4 x {236 MB cross + 236 MB self} exceed the 256 MB Infinity Cache like they do in the loop).
    Algorithmic bytes per launch: cross K/V B*12*T*64*2*2 + self K/V B*12*n*64*2*2 (+ the appended row) + the q/k/v slabs read
    + o / qc / x / y rows + the three weight matrices ONCE (each XCD really reads its own copy: 8 x; `traffic` shows it)."""
    from . import lib as L
    lib = L.load()
    H, D, C = 12, 64, 1152
    inner = H * D
    Tp = (T + 7) // 8 * 8
    n = T // 2 if n_self is None else n_self
    layers, nslab = 4, 2
    bf = torch.bfloat16
    ck = [torch.randn(B, H, Tp, D, device=device).to(bf) for _ in range(layers)]
    cv = [torch.randn(B, H, Tp, D, device=device).to(bf) for _ in range(layers)]
    sk = [torch.randn(B, H, T, D, device=device).to(bf) for _ in range(layers)]
    sv = [torch.randn(B, H, T, D, device=device).to(bf) for _ in range(layers)]
    wso = [(torch.randn(C, inner, device=device) / inner ** 0.5).to(bf) for _ in range(layers)]
    wcq = [(torch.randn(inner, C, device=device) / C ** 0.5).to(bf) for _ in range(layers)]
    wco = [(torch.randn(C, inner, device=device) / inner ** 0.5).to(bf) for _ in range(layers)]
    cs = [w.float().sum(1).contiguous() for w in wcq]
    qkv = torch.randn(nslab, B, 3 * inner, device=device) * 0.5
    x0 = torch.randn(B, C, device=device)
    x = x0.clone()
    y = torch.empty(B, C, device=device, dtype=bf)
    o = torch.empty(B, inner, device=device, dtype=bf)
    qc = torch.empty(B, inner, device=device)
    stats = torch.zeros(8, 32, 32, 2, device=device)
    km = torch.ones(B, T, dtype=torch.uint8, device=device)
    step = torch.tensor([n], dtype=torch.int32, device=device)
    scratch = torch.zeros(1024, dtype=torch.int32, device=device)
    calls = [0]

    def run(i):
        j = i % layers
        if calls[0] % 64 == 0:
            x.copy_(x0)          # the residual stream grows by two projections per call: keep it in range (outside nothing: a copy per 64 launches)
        L.check(lib.dimx_op_layer_chain(L.ptr(qkv), nslab, B * 3 * inner, L.ptr(sk[j]), L.ptr(sv[j]), T, L.ptr(ck[j]), L.ptr(cv[j]), Tp, T,
                                        L.ptr(km), L.ptr(wso[j]), L.ptr(wcq[j]), L.ptr(cs[j]), L.ptr(wco[j]), L.ptr(x), L.ptr(y), L.ptr(o),
                                        L.ptr(qc), L.ptr(stats), B, L.ptr(step), calls[0], 0.125, L.ptr(scratch), L.ptr(prof),
                                        L.stream_ptr(device)), "dimx_op_layer_chain")
        calls[0] += 1
    sec = _time_launches(run, 40, iters)
    flags = int(scratch[768].item())
    kv = B * H * 64 * 2 * 2
    alg = kv * T + kv * n + kv                      # cross K/V, cached self K/V, the appended row
    alg += nslab * B * 3 * inner * 4                # q / k / v slabs
    alg += 2 * (B * inner * 2 * 2)                  # o written + read, twice
    alg += B * inner * 4 * 2                        # qc written + read
    alg += 2 * (B * C * 4 * 2 + B * C * 2)          # x read + written, y written, twice
    alg += B * C * 2                                # y read by the q projection
    alg += 3 * C * inner * 2                        # the three weight matrices, once
    gbs = alg / sec / 1e9
    return {"kernel": "xcd_layer_kernel<2, 1, 2> (one decoder layer's attention half: self attention at %d cached keys -> out-projection -> cross-q -> "
                      "cross attention over %d keys -> out-projection; B = %d)" % (n, T, B),
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": pmc_traffic_layer(B, T, "bf16"), "algorithmic_bytes_per_launch": alg, "avg_launch_us": sec * 1e6,
            "error_flags": flags,
            "note": "the launch contains two HBM streams (attention, ~0.85 of the launch) and three latency-bound projections with four "
                    "XCD-local group barriers; `phases` (bench.py) splits it with in-kernel stamps"}


def layer_chain_phases(B, T, device, n_self=None):
    """in-kernel wall-clock stamps (100 MHz) of one xcd_layer_kernel launch: phase ends in us after the first block's start, mean and
    max over the 256 blocks, and the HBM rate of the two attention phases including the group barrier that closes them."""
    prof = torch.zeros(256 * 16, dtype=torch.int64, device=device)
    layer_chain(B, T, device, iters=24, n_self=n_self, prof=prof)
    p = prof.view(256, 16).cpu().double()
    t0 = p[:, 0][p[:, 0] > 0].min()
    us = (p[:, :13] - t0) / 100.0
    names = ["start", "self attention done", "barrier 1", "rows + W1 landed", "out-proj stored", "barrier 2", "y rows + W2 landed",
             "cross-q stored", "barrier 3", "cross attention done", "barrier 4", "rows + W3 landed", "out-proj stored"]
    out = {"stamps_us_mean_max": {"%02d %s" % (i, nm): [float(us[:, i].mean()), float(us[:, i].max())] for i, nm in enumerate(names)}}
    n = T // 2 if n_self is None else n_self
    kv = B * 12 * 64 * 2 * 2
    self_us = float(us[:, 2].mean() - us[:, 0].mean())
    cross_us = float(us[:, 10].mean() - us[:, 8].mean())
    out["self_attention_phase"] = {"us_incl_barrier": self_us, "GBps": kv * n / self_us / 1e3, "frac_of_hbm_peak": kv * n / self_us / 1e3 / HBM_PEAK_GBS}
    out["cross_attention_phase"] = {"us_incl_barrier": cross_us, "GBps": kv * T / cross_us / 1e3, "frac_of_hbm_peak": kv * T / cross_us / 1e3 / HBM_PEAK_GBS}
    out["projections_and_barriers_us"] = float(us[:, 12].mean() - us[:, 0].mean()) - self_us - cross_us
    return out


def decode_self_attention(B, T, mode, device, iters=40):
    """Self-attention form of the step kernel at the AVERAGE cache fill of the loop (T/2 keys): appends the step's
    k/v and streams the [B,12,n,64] K and V caches.  Algorithmic bytes = B*12*n*64*2*es + q/k/v in (f32) + out."""
    from . import lib as L
    lib = L.load()
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    es = 2 if mode == "bf16" else 4
    H, n = 12, T // 2
    layers = 4
    kc = [torch.randn(B, H, T, 64, device=device).to(dt) for _ in range(layers)]
    vc = [torch.randn(B, H, T, 64, device=device).to(dt) for _ in range(layers)]
    qkv = torch.randn(B, 3 * H * 64, device=device)
    out = torch.empty(B, H * 64, device=device, dtype=dt)
    step = torch.tensor([n], dtype=torch.int32, device=device)

    def run(i):
        L.check(lib.dimx_op_decode_attn_self(L.BF16 if mode == "bf16" else L.F32, L.ptr(qkv), 3 * H * 64,
                                             L.ptr(kc[i % layers]), L.ptr(vc[i % layers]), L.ptr(out), B, H, T,
                                             L.ptr(step), 0.125, 1, L.stream_ptr(device)), "decode_attn_self")
    sec = _time_launches(run, 8, iters)
    alg_bytes = B * H * n * 64 * 2 * es + B * 3 * H * 64 * 4 + B * H * 64 * es + 2 * B * H * 64 * es
    gbs = alg_bytes / sec / 1e9
    return {"kernel": "decode_attn_kernel<%s, true, true, 1> (self-attention form, %d cached keys = mean fill)" % (
                "dimx::bf16" if mode == "bf16" else "float", n), "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_us": sec * 1e6}


def decode_layernorm(B, mode, device, iters=60, C=1152, nslab=4):
    """Residual + pre-norm of the decode step (x += 4 split-K slabs; y = LN(x)): B rows of 1152.  Algorithmic bytes =
    B*C*(4 read x + 4*nslab read slabs + 4 write x + es write y); latency-bound at B = 256."""
    from . import lib as L
    lib = L.load()
    es = 2 if mode == "bf16" else 4
    x = torch.randn(B, C, device=device)
    slabs = torch.randn(nslab, B, C, device=device) * 0.01
    y = torch.empty(B, C, device=device, dtype=torch.bfloat16 if mode == "bf16" else torch.float32)
    g = torch.ones(C, device=device)

    def run(i):
        L.check(lib.dimx_op_add_slabs_layernorm(L.BF16 if mode == "bf16" else L.F32, L.ptr(x), L.ptr(slabs), nslab, B * C,
                                                L.ptr(y), L.ptr(g), B, C, L.stream_ptr(device)), "add_slabs_layernorm")
    sec = _time_launches(run, 8, iters)
    alg_bytes = B * C * (4 + 4 * nslab + 4 + es)
    gbs = alg_bytes / sec / 1e9
    return {"kernel": "add_slabs_layernorm_kernel<%s, %d> (%d slabs)" % (mode, C, nslab), "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": sec * 1e6}


def decode_gemm(B, mode, device, iters=40):
    M, N, K = B, 1152, 4608
    a = torch.randn(M, K, device=device)
    w = [torch.randn(N, K, device=device) / 68.0 for _ in range(4)]
    bias = torch.randn(N, device=device)
    res = torch.randn(M, N, device=device)
    bf = mode == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    a_ = a.to(dt).contiguous()
    w_ = [x.to(dt).contiguous() for x in w]
    from . import lib as L
    lib = L.load()
    out = torch.empty(8, M, N, device=device)   # split-K slabs, as the decode step launches this projection

    def run(i):
        L.check(lib.dimx_op_gemm(L.BF16 if bf else L.F32, L.F32, L.ptr(a_), K, L.ptr(w_[i % 4]), K, L.ptr(out), N, M,
                                 N, K, L.ptr(bias), 0, None, 0, 0, None, 5 if bf else 4, L.stream_ptr(device)), "gemm")
    sec = _time_launches(run, 8, iters)
    flops = 2.0 * M * N * K
    tf = flops / sec / 1e12
    peak = MFMA_PEAK_TFLOPS[mode]
    name = "gemm_ws72_kernel<float> (64 x 72 tiles, one block per CU)" if (bf and M <= 256) else "gemm_ws_kernel<%s,float>" % mode
    return {"kernel": "%s M=%d N=%d K=%d (decode FF down-projection, split-K slabs; latency-bound: "
                      "4.7 MB of weights per launch)" % (name, M, N, K),
            "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
            "algorithmic_flops_per_launch": flops, "avg_launch_us": sec * 1e6}


def decode_gemm_x3(B, device, iters=60):
    """Round 6: the f32 parity mode's decode GEMMs on the bf16 matrix cores (csrc/gemm_x3.hip: three bf16 planes per operand, six
    products, f32 accumulation = an f32 GEMM) -- the three chip-wide shapes of a decoder layer, launched as dimx_generate launches
    them (pre-split weight planes, split-K slabs where the step uses them), rotating over four layers' weights.  `achieved` counts
    the f32 GEMM's flops (2 M N K) against the f32 MFMA peak the mode would otherwise be bound by; `bf16_mfma_tflops` the six bf16
    products actually issued against the bf16 peak."""
    from . import lib as L
    lib = L.load()
    M = B
    shapes = [("fused q/k/v", 2304, 1152, True, 0), ("ff1 + erf-GELU", 4608, 1152, False, 3), ("ff2", 1152, 4608, True, 0)]
    out = []
    for name, N, K, slabs, act in shapes:
        a = torch.randn(M, K, device=device)
        planes = [E.op_split_x3(torch.randn(N, K, device=device) / K ** 0.5) for _ in range(4)]
        bias = torch.randn(N, device=device)
        flags = 4 if slabs else 0
        ns = lib.dimx_op_gemm_slabs(L.F32, M, N, K, 16 | 5) if slabs else 1
        c = torch.empty(ns, M, N, device=device)

        def run(i):
            L.check(lib.dimx_op_gemm_x3(L.ptr(a), K, L.ptr(planes[i % 4]), L.ptr(c), N, M, N, K, None if slabs else L.ptr(bias), act, None, 0,
                                        flags, L.stream_ptr(device)), "gemm_x3")
        sec = _time_launches(run, 10, iters)
        flops = 2.0 * M * N * K
        out.append({"kernel": "gemm_x3_kernel M=%d N=%d K=%d (%s; f32 parity mode, %s)" % (M, N, K, name, "%d split-K slabs" % ns if slabs else "one pass over K"),
                    "bound": "mfma", "achieved": flops / sec / 1e12, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                    "frac": flops / sec / 1e12 / MFMA_PEAK_TFLOPS["f32"], "traffic": None, "algorithmic_flops_per_launch": flops,
                    "bf16_mfma_tflops": 6 * flops / sec / 1e12, "bf16_mfma_frac": 6 * flops / sec / 1e12 / MFMA_PEAK_TFLOPS["bf16"],
                    "avg_launch_us": sec * 1e6})
    return out


def cross_kv_gemm(B, T, mode, device, iters=12):
    """The MFMA GEMM of the cross-attention bundle as dimx_encode_ctx launches it: the K/V projection of the speaker
    context, [B*T, 1152] x [N, 1152]^T into head-major [B,12,Tp,64] caches.  bf16: all four decoder layers in one launch
    of the 256 x 256 phase-pipelined kernel (N = 6144); f32: one layer per launch (N = 1536).  BASELINE.json's
    'cross-attn MFMA util %' is quoted on it: achieved TFLOP/s / dense MFMA peak of the operand type (2.5 PFLOP/s is
    the 2.4 GHz figure; the clock under this load is lower, see profiles/)."""
    from . import lib as L
    lib = L.load()
    bf = mode == "bf16"
    nlayers = 4 if bf else 1
    M, N, K = B * T, nlayers * 1536, 1152
    Tp = (T + 7) // 8 * 8
    dt = torch.bfloat16 if bf else torch.float32
    a = torch.randn(M, K, device=device).to(dt)
    w = [(torch.randn(N, K, device=device) / 34.0).to(dt) for _ in range(2)]
    out = torch.empty(2 * nlayers, B, 12, Tp, 64, device=device, dtype=dt)

    def run(i):
        L.check(lib.dimx_op_gemm_headmajor(L.BF16 if bf else L.F32, L.ptr(a), K, L.ptr(w[i % 2]), K, L.ptr(out), M, N, K, T,
                                           Tp, nlayers, L.stream_ptr(device)), "gemm_headmajor")
    runs = sorted(_time_launches(run, 3, iters) for _ in range(5))      # 5 back-to-back measurements: the spread is the box's DVFS
    sec = runs[len(runs) // 2]
    flops = 2.0 * M * N * K
    tf = flops / sec / 1e12
    peak = MFMA_PEAK_TFLOPS[mode]
    out = {"kernel": "%s (cross-attention K/V projection, %d layer%s per launch) M=%d N=%d K=%d" % (
               "gemm256p2_kernel<bf16>" if bf else "gemm_glds_kernel<float>", nlayers, "s" if nlayers > 1 else "", M, N, K),
           "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "util_pct": 100.0 * tf / peak,
           "avg_launch_us": sec * 1e6, "pmc_mfma_busy_pct": None,
           "util_pct_runs": {"n": len(runs), "median": 100.0 * tf / peak, "best": 100.0 * flops / runs[0] / 1e12 / peak,
                             "worst": 100.0 * flops / runs[-1] / 1e12 / peak},
           "note": "achieved / util_pct are the MEDIAN of %d measurements of %d launches each" % (len(runs), iters)}
    # MFMA busy cycles / kernel cycles from the PMC pass recorded for THIS kernel source (same rule as `traffic`): the
    # flop-rate figure above is against the 2.4 GHz peak, the counter figure is against the clock the kernel really ran at
    path = os.path.join(PROFILES, "pmc_gemm256_%s.json" % kernel_source_hash("gemm256.hip"))
    if bf and (B, T) == (256, 300) and os.path.exists(path):
        with open(path) as fh:
            rec = json.load(fh)
        out["pmc_mfma_busy_pct"] = rec.get("mfma_busy_pct")
        out["pmc_shader_clock_GHz"] = rec.get("shader_clock_GHz_under_pmc")
    return out


def _chain_launch_us(B, device, with_q, iters=60):
    """One deferred-LayerNorm chain launch of the decode step as dimx_generate issues it (bf16): the attention
    out-projection 768 -> 1152 + residual (+ the cross-attention q projection 1152 -> 768 when with_q)."""
    from . import lib as L
    lib = L.load()
    C, K1, N2 = 1152, 768, 768
    a1 = torch.randn(B, K1, device=device).to(torch.bfloat16)
    w1 = [(torch.randn(C, K1, device=device) / 28.0).to(torch.bfloat16) for _ in range(4)]
    w2 = [(torch.randn(N2, C, device=device) / 34.0).to(torch.bfloat16) for _ in range(4)]
    cs = [w.float().sum(1).contiguous() for w in w2]
    x = torch.randn(B, C, device=device)
    y = torch.empty(B, C, dtype=torch.bfloat16, device=device)
    stats = torch.zeros(8, 32, 32, 2, device=device)
    out2 = torch.empty(B, N2, device=device)
    scratch = torch.zeros(512 + B * C, dtype=torch.int32, device=device)

    def run(i):
        L.check(lib.dimx_op_chain_ln(L.ptr(a1), K1, L.ptr(w1[i % 4]), L.ptr(x), L.ptr(y), L.ptr(stats),
                                     L.ptr(w2[i % 4]) if with_q else None, L.ptr(cs[i % 4]) if with_q else None,
                                     N2 if with_q else 0, L.ptr(out2) if with_q else None, B, C, L.ptr(scratch),
                                     L.stream_ptr(device)), "chain_ln")
    sec = _time_launches(run, 8, iters)
    assert int(scratch[129].item()) & 3 == 0
    return sec * 1e6


def pmc_prefill_busy(kernel_prefix):
    """MFMA busy % of a prefill kernel from the PMC pass recorded for the current sources (tools/pmc_prefill_record.py), else None"""
    path = os.path.join(PROFILES, "pmc_prefill_mfma_%s.json" % kernel_source_hash(PREFILL_MFMA_SOURCES))
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        rec = json.load(fh)
    for k, v in rec.get("kernels", {}).items():
        if k.startswith(kernel_prefix):
            return v.get("mfma_busy_pct")
    return None


def prefill_cross_attention(B, T, device, iters=10):
    """The teacher-forced cross attention (mode='train': 299 queries x 300 context keys, 12 heads x 64, context mask) on the
    prefill attention kernel: MFMA utilisation = 4 B H Lq Lk 64 flops / time / 2.5 PFLOP/s."""
    from . import lib as L
    lib = L.load()
    H, D, Lq, Lk = 12, 64, T - 1, T
    C = H * D
    bufs = [tuple(torch.randn(B, n, C, device=device).to(torch.bfloat16) for n in (Lq, Lk, Lk)) for _ in range(3)]
    out = torch.empty(B, Lq, C, device=device, dtype=torch.bfloat16)
    km = torch.ones(B, Lk, dtype=torch.uint8, device=device)

    def run(i):
        q, k, v = bufs[i % 3]
        L.check(lib.dimx_op_attention_rowv(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), B, H, Lq, Lk, D, C, C, C, C, 0.125, 0, None,
                                           L.ptr(km), L.stream_ptr(device)), "attention_rowv")
    sec = _time_launches(run, 3, iters)
    flops = 4.0 * B * H * Lq * Lk * D
    tf = flops / sec / 1e12
    return {"kernel": "attn_tr_kernel<64, 2> (persistent, row-major q / k / v, Lq %d x Lk %d, 12 heads x 64, context mask)" % (Lq, Lk),
            "avg_launch_us": sec * 1e6, "achieved": tf, "unit": "TFLOP/s", "util_pct": 100.0 * tf / MFMA_PEAK_TFLOPS["bf16"],
            "pmc_mfma_busy_pct": pmc_prefill_busy("attn_tr_kernel<64")}


def prefill_mlp_fused(B, T, device, iters=8):
    """The fused feed-forward sublayer of the prefill (csrc/mlp_fused.hip; 20 launches per forward: 12 in the VQ-VAE stacks with
    tanh-GELU, 8 in the encoders with erf-GELU) at the headline row count M = B T: 4 M C F flops (C = 384, F = 1536) against the
    dense bf16 MFMA peak, and its algorithmic bytes (x read once for the LayerNorm, read again for the residual, written once)."""
    import ctypes
    from . import lib as L
    lib = L.load()
    M, C, F = B * T, 384, 1536
    g = torch.Generator().manual_seed(11)
    w1, b1, w2 = torch.randn(F, C, generator=g) * C ** -0.5, torch.randn(F, generator=g) * 0.05, torch.randn(C, F, generator=g) * F ** -0.5
    nbytes = int(lib.dimx_mlp_fused_packed_bytes(C, F))
    host = torch.empty(nbytes, dtype=torch.uint8)
    hp = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(lib.dimx_mlp_fused_pack(hp(w1), hp(b1), hp(w2), C, F, hp(host), nbytes), "mlp_fused_pack")
    packed = host.to(device)
    b2, ln_g, ln_b = torch.randn(C, device=device) * 0.02, torch.rand(C, device=device) + 0.5, torch.randn(C, device=device) * 0.1
    xs = [torch.randn(M, C, device=device) for _ in range(2)]
    out = []
    for act, beta, name in ((2, ln_b, "tanh-GELU, LayerNorm with bias (VQ-VAE blocks)"), (3, None, "erf-GELU (encoders)")):
        def run(i):
            L.check(lib.dimx_op_mlp_fused_packed(L.ptr(xs[i % 2]), L.ptr(packed), L.ptr(b2), L.ptr(ln_g), L.ptr(beta), M, C, F, act,
                                                 L.stream_ptr(device)), "mlp_fused")
        sec = _time_launches(run, 2, iters)
        for x in xs:                      # the sublayer is applied in place: keep the rows in a sane range between launches
            x.normal_()
        tf = 4.0 * M * C * F / sec / 1e12
        out.append({"kernel": "mlp_fused_kernel (x += W2 gelu(W1 LN(x) + b1) + b2 in one launch, M %d x 384 -> 1536 -> 384; %s)" % (M, name),
                    "bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS["bf16"],
                    "traffic": None, "algorithmic_flops_per_launch": 4.0 * M * C * F, "algorithmic_bytes_per_launch": 3 * M * C * 4,
                    "avg_launch_us": sec * 1e6, "pmc_mfma_busy_pct": pmc_prefill_busy("mlp_fused_kernel<%d" % act)})
    return out


def cross_attn_bundle(B, T, mode, device, roof, kv):
    """SURVEY 8(d)'s 'cross-attention GEMM' bundle -- K/V projection of the context (4.25 GFLOP per clip at T = 300) + the
    cross-attention q and out projections of the T - 1 decode steps (4.23) + the decode steps' scores / AV (1.10) = 9.58
    GFLOP per clip -- as achieved TFLOP/s over the summed time of the kernels that execute it in SLMFT.forward(mode='val'):
    the fused 4-layer K/V projection launch (measured: `cross_attn_mfma`), the decode cross-attention launches (measured:
    `roofline`), and the chain launches that hold the two projections (measured here; the launch that also carries the
    self-attention out-projection is charged half its time, its two projections have equal flops).  The bundle is dominated
    by the HBM-bound one-query attention, so its MFMA utilisation is small by construction; the K/V projection alone is the
    MFMA-bound member (`cross_attn_mfma`).  `teacher_forced_cross_attention` is the same attention in mode='train'."""
    assert mode == "bf16"
    depth, steps = 4, T - 1
    layer = roof.get("phases") if "xcd_layer_kernel" in roof.get("kernel", "") else None
    if layer is not None:
        # round 5: the decode step's cross attention and its two projections are phases of xcd_layer_kernel -- their time comes
        # from its in-kernel stamps (means over the 256 blocks): cross-q = barrier 2 -> barrier 3, the attention = barrier 3 ->
        # barrier 4, the out-projection = barrier 4 -> end
        st = {k[:2]: v[0] for k, v in layer["stamps_us_mean_max"].items()}
        t_att = st["10"] - st["08"]
        t_a = 2.0 * (st["08"] - st["05"])       # charged half below, like the chain launch that also held the self out-projection
        t_b = st["12"] - st["10"]
    else:
        att = roof if "decode_attn" in roof.get("kernel", "") else next(
            (o for o in [roof.get("secondary", {})] + roof.get("others", []) if "decode_attn" in o.get("kernel", "")), {})
        t_att = att["avg_launch_us"]
        t_a = _chain_launch_us(B, device, True)
        t_b = _chain_launch_us(B, device, False)
    f_kv = 2.0 * B * T * (depth * 1536) * 1152
    f_proj = 2.0 * B * 1152 * 768                    # one projection of one layer-step
    f_att = 4.0 * B * 768 * T                        # scores + AV of one layer-step
    t_total = kv["avg_launch_us"] + depth * steps * (t_att + 0.5 * t_a + t_b)
    f_total = f_kv + depth * steps * (2 * f_proj + f_att)
    tf = f_total / t_total / 1e6
    return {"gflop_per_clip": f_total / B / 1e9, "achieved": tf, "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS["bf16"],
            "util_pct": 100.0 * tf / MFMA_PEAK_TFLOPS["bf16"], "time_ms_per_batch": t_total / 1e3,
            "parts_us": {"kv_projection (1 launch, 4 layers)": kv["avg_launch_us"], "decode_cross_attention (per launch)": t_att,
                         "chain: self out-proj + residual + cross q-proj (per launch, half charged)": t_a,
                         "chain: cross out-proj + residual (per launch)": t_b, "launches_per_batch": depth * steps,
                         "source": "phases of xcd_layer_kernel (in-kernel stamps)" if layer is not None else "separate launches (HIP events)"},
            "parts_gflop_per_clip": {"kv_projection": f_kv / B / 1e9, "q_and_out_projections": depth * steps * 2 * f_proj / B / 1e9,
                                     "scores_and_av": depth * steps * f_att / B / 1e9},
            "teacher_forced_cross_attention": prefill_cross_attention(B, T, device)}


def dominant_kernel(eng, B, T, mode):
    """Roofline object for bench.py: the kernel with the largest share of the decode loop, with the other
    candidate attached under 'secondary'."""
    dev = eng.device
    att = decode_attention(B, T, mode, dev)
    gem = decode_gemm(B, mode, dev)
    if mode == "bf16" and 128 < B <= 256:
        # round 5: for this batch range the attentions run inside xcd_layer_kernel (4 launches per decode step, ~70 % of it): THAT
        # is the dominant kernel of the trace; the standalone attention kernel (smaller batches, several samples per clip, the f32
        # mode) stays on the line under `others` with its own PMC traffic record
        first = layer_chain(B, T, dev)
        first["phases"] = layer_chain_phases(B, T, dev)
        first["secondary"] = gem
        first["others"] = [att, decode_self_attention(B, T, mode, dev), decode_layernorm(B, mode, dev)]
        first["others"] += prefill_mlp_fused(B, T, dev)
        pa = prefill_cross_attention(B, T, dev)
        pa.update(bound="mfma", peak=MFMA_PEAK_TFLOPS["bf16"], frac=pa["util_pct"] / 100.0, traffic=None)
        first["others"].append(pa)
        return first
    # per decode step: 8 attention launches vs 16 small-M GEMM launches of comparable size
    att_share = 8 * att["avg_launch_us"]
    gem_share = 16 * gem["avg_launch_us"]
    first, second = (att, gem) if att_share >= gem_share else (gem, att)
    first = dict(first)
    first["secondary"] = second
    first["others"] = [decode_self_attention(B, T, mode, dev), decode_layernorm(B, mode, dev)]
    if mode == "bf16":
        first["others"] += prefill_mlp_fused(B, T, dev)
    elif B <= 256:
        first["others"] += decode_gemm_x3(B, dev)
    return first
