"""GPU: the seq2seq stages (encode_ctx, decode_tf, generate) through the C-ABI against the CPU oracle on the
same seeded inputs (f32 parity mode), plus perf-mode (bf16) agreement reports."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LOGIT_TOL = 1e-4   # SURVEY Appendix C; measured on MI355X (round 3): 3.6e-6 .. 3.7e-6 at T = 24 .. 300


@pytest.fixture(scope="module")
def eng(full_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict(full_sd)
    return e


@pytest.fixture(scope="module")
def eng_bf16(full_sd):
    from dimx import engine, lib
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    e.load_state_dict(full_sd)
    return e


def _case(B, T, lens, seed=3):
    from dimx import prng
    v_s = torch.from_numpy(prng.normal(seed, "s2s.vs", (B, T, 56)))
    v_a = torch.from_numpy(prng.normal(seed, "s2s.va", (B, T, 768)))
    z = torch.from_numpy(prng.integers(seed, "s2s.z", (B, T), 0, 512))
    mask = torch.zeros(B, T, dtype=torch.bool)
    for j, n in enumerate(lens):
        mask[j, :n] = True
    z = torch.where(mask, z, torch.full_like(z, -100))
    return v_s, v_a, z, mask


@pytest.mark.parametrize("B,T,lens", [(3, 40, [40, 33, 7]), (2, 300, [300, 212])])
def test_encode_ctx_matches_oracle(eng, full_sd, B, T, lens):
    from oracle import ref_cpu
    v_s, v_a, z, mask = _case(B, T, lens)
    ref = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    x_s = eng.encode_ctx(v_s.cuda(), v_a.cuda(), mask.to(torch.uint8).cuda(), False, return_x_s=True).cpu()
    for b, n in enumerate(lens):
        err = (x_s[b, :n] - ref[b, :n]).abs().max().item()
        assert err < 1e-4, "x_s err %g clip %d" % (err, b)


@pytest.mark.parametrize("B,T,lens,use_kv", [(3, 40, [40, 33, 7], True), (2, 300, [300, 212], True),
                                             (2, 24, [24, 24], False)])
def test_decode_tf_matches_oracle(eng, full_sd, B, T, lens, use_kv):
    from oracle import ref_cpu
    v_s, v_a, z, mask = _case(B, T, lens)
    kv = ref_cpu.ar_kv_mask(B, T, 0.15, torch.Generator().manual_seed(1)) if use_kv else None
    x_s = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    ctx = ref_cpu.slmft_context(full_sd, x_s, v_a)
    loss, ref = ref_cpu.ar_forward(full_sd, z, ctx, mask, kv)
    m8 = mask.to(torch.uint8).cuda()
    eng.encode_ctx(v_s.cuda(), v_a.cuda(), m8, False)
    logits, row_loss, amax = eng.decode_tf(z.cuda(), m8, kv.to(torch.uint8).cuda() if use_kv else None)
    logits = logits.cpu()
    err = (logits - ref).abs().max().item()
    print("teacher-forced logits max |gpu - oracle| = %.2e (B=%d T=%d)" % (err, B, T))
    # measured on MI355X (round 3): 3.6e-6 .. 3.7e-6 (f32 MFMA, other summation order than the CPU's blocked GEMMs);
    # SURVEY Appendix C asks for 1e-4, asserted as such (round 2 had 1e-3 here)
    assert err < LOGIT_TOL, "logit err %g" % err
    # argmax identical where the top-2 margin is comfortable
    top2 = ref.topk(2, -1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(amax.cpu().long()[safe], ref.argmax(-1)[safe])
    tgt = z[:, 1:]
    n_valid = (tgt != -100).sum()
    my_loss = row_loss.cpu().sum() / n_valid
    assert abs(my_loss.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))


@pytest.mark.parametrize("B,T,lens,noisy", [(3, 40, [40, 33, 7], False), (3, 40, [40, 33, 7], True),
                                            (4, 300, [300, 300, 251, 190], False),
                                            (4, 300, [300, 300, 251, 190], True)])
def test_generate_matches_oracle(eng, full_sd, B, T, lens, noisy):
    from dimx import prng
    from oracle import ref_cpu
    v_s, v_a, z, mask = _case(B, T, lens, seed=9)
    noise = torch.from_numpy(prng.exponential(11, "s2s.noise", (T - 1, B, 512))) if noisy else None
    x_s = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    ctx = ref_cpu.slmft_context(full_sd, x_s, v_a)
    start = z[:, 0]
    ref_tok, ref_lg = ref_cpu.ar_generate(full_sd, start, T - 1, ctx, mask, noise, return_logits=True)
    m8 = mask.to(torch.uint8).cuda()
    eng.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    tok, lg = eng.generate(start.cuda(), m8, T, 1.0 if noisy else 0.0, 52, noise.cuda() if noisy else None,
                           return_logits=True)
    tok, lg = tok.cpu().long(), lg.cpu()
    same = (tok == ref_tok)
    if not same.all():
        # report the first divergence and how close the decision was
        b, t = [int(v[0]) for v in torch.nonzero(~same, as_tuple=True)]
        raise AssertionError("token mismatch: %d/%d differ, first at clip %d step %d (logit err before: %g)" % (
            (~same).sum(), same.numel(), b, t, (lg[b, :t + 1] - ref_lg[b, :t + 1]).abs().max()))
    assert (lg - ref_lg).abs().max() < 2e-3
    # KV-cached generation == teacher-forced logits at the sampled prefix (GPU self-consistency)
    seq = torch.cat([start[:, None], tok], 1)
    eng.encode_ctx(v_s.cuda(), v_a.cuda(), m8, False)
    tf_logits, _, _ = eng.decode_tf(seq.cuda(), m8, None)
    assert (tf_logits.cpu() - lg).abs().max() < 2e-3


def test_generate_graph_equals_eager(full_sd, monkeypatch):
    import os
    from dimx import engine, lib
    v_s, v_a, z, mask = _case(2, 30, [30, 21], seed=4)
    m8 = mask.to(torch.uint8).cuda()
    outs = []
    for flag in ("0", "1"):
        os.environ["DIMX_NO_GRAPH"] = flag
        e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
        e.load_state_dict(full_sd)
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        outs.append(e.generate(z[:, 0].cuda(), m8, 30, 0.0).cpu())
        # replaying the captured graph a second time must give the same tokens
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        assert torch.equal(e.generate(z[:, 0].cuda(), m8, 30, 0.0).cpu(), outs[-1])
        e.close()
    os.environ.pop("DIMX_NO_GRAPH", None)
    assert torch.equal(outs[0], outs[1])


def test_generate_is_independent_of_clip_grouping(full_sd):
    """dimx_generate decodes independent clip groups concurrently on separate streams; the tokens must not
    depend on the number of groups (sampler noise is indexed by the global clip row)."""
    import os
    from dimx import engine, lib, prng
    B, T = 5, 36
    v_s, v_a, z, mask = _case(B, T, [36, 30, 36, 12, 25], seed=6)
    noise = torch.from_numpy(prng.exponential(2, "grp.noise", (T - 1, B, 512))).cuda()
    m8 = mask.to(torch.uint8).cuda()
    outs = []
    for groups in ("1", "2", "3"):
        os.environ["DIMX_GEN_GROUPS"] = groups
        e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
        e.load_state_dict(full_sd)
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        a = e.generate(z[:, 0].cuda(), m8, T, 1.0, 52, noise).cpu()
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        b = e.generate(z[:, 0].cuda(), m8, T, 1.0, 52, None, seed=77).cpu()
        outs.append((a, b))
        e.close()
    os.environ.pop("DIMX_GEN_GROUPS", None)
    for a, b in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(b, outs[0][1])


def test_bf16_mode_agreement_report(eng_bf16, full_sd):
    """perf mode: report (not assert bit-exactness) token agreement and logit error vs the f32 oracle."""
    from oracle import ref_cpu
    B, T, lens = 4, 120, [120, 120, 90, 64]
    v_s, v_a, z, mask = _case(B, T, lens, seed=21)
    x_s = ref_cpu.slmft_forward_encoder(full_sd, v_s, mask)
    ctx = ref_cpu.slmft_context(full_sd, x_s, v_a)
    _, ref = ref_cpu.ar_forward(full_sd, z, ctx, mask, None)
    m8 = mask.to(torch.uint8).cuda()
    xs_g = eng_bf16.encode_ctx(v_s.cuda(), v_a.cuda(), m8, False, return_x_s=True).cpu()
    logits, _, amax = eng_bf16.decode_tf(z.cuda(), m8, None)
    valid = mask[:, 1:]
    lerr = (logits.cpu() - ref).abs()[valid].max().item()
    agree = (amax.cpu().long() == ref.argmax(-1))[valid].float().mean().item()
    xerr = max((xs_g[b, :n] - x_s[b, :n]).abs().max().item() for b, n in enumerate(lens))
    print("bf16 perf mode: x_s max err %.4f, logit max err %.4f, argmax agreement %.3f" % (xerr, lerr, agree))
    # measured on MI355X (round 2): x_s 0.028, logits 0.0085, argmax agreement 0.995 -- asserted with a small margin
    assert agree >= 0.99 and lerr <= 2e-2 and xerr <= 6e-2


def test_chain_fault_is_reported_and_repaired_by_the_call_that_suffered_it(full_sd):
    """bf16 mode: a generate() whose XCD-local chain kernels run on a non-bijective (XCD, CU slot) placement (forced through
    dimx_debug_chain_fault: the blocks of every odd XCD claim their even neighbour's slots) must not hand back the tokens
    those kernels produced: the SAME call detects the claim collision, regenerates the batch on the one-kernel-per-op
    step and counts the event; the tokens equal those of a handle that never used the chain kernels."""
    import os
    from dimx import engine, lib
    B, T = 6, 40
    v_s, v_a, z, mask = _case(B, T, [40, 33, 40, 12, 40, 25], seed=9)
    m8 = mask.to(torch.uint8).cuda()

    def make(no_chain):
        if no_chain:
            os.environ["DIMX_NO_CHAIN"] = "1"
        try:
            e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
        finally:
            os.environ.pop("DIMX_NO_CHAIN", None)
        e.load_state_dict(full_sd)
        return e

    ref_eng = make(True)
    ref_eng.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    ref_tok = ref_eng.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()
    assert ref_eng.chain_faults() == 0
    ref_eng.close()

    e = make(False)
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    ok_tok = e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()       # healthy chain path first
    assert e.chain_faults() == 0
    e.debug_chain_fault(1)
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    tok = e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()
    assert e.chain_faults() == 1, "the faulted call itself must notice"
    assert torch.equal(tok, ref_tok), "the faulted batch must be regenerated without the chain kernels"
    # the handle keeps working (chain path off from now on) and stays deterministic
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    assert torch.equal(e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu(), ref_tok)
    assert e.chain_faults() == 1
    # healthy chain tokens agree with the one-kernel-per-op step until the first bf16 rounding tie flips a greedy choice
    # (deferred LayerNorm rounds differently; after a flip an autoregressive sequence is a different sequence)
    assert torch.equal(ok_tok[:, :8], ref_tok[:, :8])
    e.close()


def test_layer_kernel_reproduces_the_four_launches_bit_for_bit(full_sd):
    """round 5, bf16 mode, > 128 clips: self attention -> out-projection -> cross-q -> cross attention -> out-projection as ONE
    XCD-local launch per layer (chain.hip xcd_layer_kernel) runs the same arithmetic in the same order as the four launches it
    replaces (DIMX_NO_LAYER_CHAIN=1): identical tokens, greedy and sampled, ragged context masks, a batch that leaves the last
    clip group partly empty; no chain fault."""
    import os
    from dimx import engine, lib
    B, T = 200, 48
    lens = [T - (i * 7) % 20 for i in range(B)]
    v_s, v_a, z, mask = _case(B, T, lens, seed=31)
    m8 = mask.to(torch.uint8).cuda()
    noise = torch.empty(T - 1, B, 512).exponential_(generator=torch.Generator().manual_seed(5)).cuda()

    def run(off):
        if off:
            os.environ["DIMX_NO_LAYER_CHAIN"] = "1"
        try:
            e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
        finally:
            os.environ.pop("DIMX_NO_LAYER_CHAIN", None)
        e.load_state_dict(full_sd)
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        greedy = e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        sampled = e.generate(z[:, 0].cuda(), m8, T, 1.0, noise=noise).cpu()
        faults = e.chain_faults()
        e.close()
        return greedy, sampled, faults

    g1, s1, f1 = run(False)
    g0, s0, f0 = run(True)
    assert f0 == 0 and f1 == 0
    assert torch.equal(g1, g0) and torch.equal(s1, s0)


def test_layer_kernel_fault_is_reported_and_repaired(full_sd):
    """the layer kernel answers for its placement like the chain kernels do: with a forced non-bijective (XCD, CU slot) claim the
    SAME generate() call notices, regenerates the batch on the one-kernel-per-op step and counts the event."""
    import os
    from dimx import engine, lib
    B, T = 160, 24
    v_s, v_a, z, mask = _case(B, T, [T] * B, seed=41)
    m8 = mask.to(torch.uint8).cuda()
    os.environ["DIMX_NO_CHAIN"] = "1"
    try:
        ref_eng = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    finally:
        os.environ.pop("DIMX_NO_CHAIN", None)
    ref_eng.load_state_dict(full_sd)
    ref_eng.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    ref_tok = ref_eng.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()
    ref_eng.close()
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    e.load_state_dict(full_sd)
    e.debug_chain_fault(1)
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    tok = e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu()
    assert e.chain_faults() == 1
    assert torch.equal(tok, ref_tok)
    e.close()


def test_optional_bias_tensors_of_other_xtransformers_releases(full_sd):
    """SURVEY A.2 [XT?]: a state dict that carries project_in.bias / to_logits.bias is computed WITH them (encoder, teacher-forced
    logits, cached generation against the oracle reading the same dict; the bf16 decode step leaves its last chain kernel, which
    has no bias input, for two launches); a zero LayerNorm bias is accepted, a non-zero one refused; a later state dict without
    the tensors restores the bias-free results bit for bit; the training step refuses the tensors."""
    from dimx import engine, lib, prng
    from oracle import ref_cpu
    B, T, lens = 3, 40, [40, 33, 7]
    v_s, v_a, z, mask = _case(B, T, lens, seed=17)
    sd = dict(full_sd)
    sd["encoder_s.project_in.bias"] = torch.from_numpy(prng.normal(5, "xt.b0", (384,))) * 0.3
    sd["encoder_joint.project_in.bias"] = torch.from_numpy(prng.normal(5, "xt.b1", (384,))) * 0.3
    sd["decoder_joint.net.to_logits.bias"] = torch.from_numpy(prng.normal(5, "xt.b2", (512,))) * 0.5
    sd["decoder_joint.net.attn_layers.final_norm.bias"] = torch.zeros(1152)
    m8 = mask.to(torch.uint8).cuda()

    def oracle(d):
        x_s = ref_cpu.slmft_forward_encoder(d, v_s, mask)
        ctx = ref_cpu.slmft_context(d, x_s, v_a)
        _, lg = ref_cpu.ar_forward(d, z, ctx, mask, None)
        tok = ref_cpu.ar_generate(d, z[:, 0], T - 1, ctx, mask, None)
        return x_s, lg, tok

    def run(e):
        x_s = e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, False, return_x_s=True).cpu()
        lg = e.decode_tf(z.cuda(), m8, None)[0].cpu()
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
        tok = e.generate(z[:, 0].cuda(), m8, T, 0.0).cpu().long()
        return x_s, lg, tok

    e = engine.Engine("cuda:0", lib.MODE_PARITY_F32)
    e.load_state_dict(full_sd)
    x0, lg0, tok0 = run(e)
    e.load_state_dict(sd)
    x1, lg1, tok1 = run(e)
    rx, rlg, rtok = oracle(sd)
    for b, n in enumerate(lens):
        assert (x1[b, :n] - rx[b, :n]).abs().max() < 1e-4
    assert (lg1 - rlg).abs().max() < LOGIT_TOL and torch.equal(tok1, rtok)
    assert (lg1 - lg0).abs().max() > 0.1          # the tensors are not decoration
    bad = dict(full_sd)
    bad["encoder_s.attn_layers.layers.0.0.0.bias"] = torch.full((384,), 1e-3)
    with pytest.raises(lib.DimxError):
        e.load_state_dict(bad)
    e.load_state_dict(full_sd)                      # names the weights, not the biases: they go
    x2, lg2, tok2 = run(e)
    assert torch.equal(x2, x0) and torch.equal(lg2, lg0) and torch.equal(tok2, tok0)
    # round 6: a checkpoint loaded in pieces gives the same result in ANY order -- dimx_load_weights never drops a
    # tensor, only dimx_begin_checkpoint (a new checkpoint) forgets the optional ones.  Biases first, weights afterwards:
    opt = {k: v for k, v in sd.items() if k.endswith("project_in.bias") or k.endswith("to_logits.bias")}
    rest = {k: v for k, v in sd.items() if k not in opt}
    e.load_state_dict(opt, new_checkpoint=True)          # begins the checkpoint with the optional tensors only
    e.load_state_dict(rest, new_checkpoint=False)        # ... the weights arrive later and name no bias
    x3, lg3, tok3 = run(e)
    assert torch.equal(x3, x1) and torch.equal(lg3, lg1) and torch.equal(tok3, tok1)
    e.load_state_dict(rest, new_checkpoint=False)        # the same weights again: still nothing is dropped
    assert torch.equal(run(e)[1], lg1)
    e.load_state_dict(full_sd)                           # a NEW checkpoint without the tensors: gone
    assert torch.equal(run(e)[1], lg0)
    e.close()

    # bf16 decode step: the logits of generate() move by the bias (the rest of the step is the chain path it always was)
    only = dict(full_sd)
    only["decoder_joint.net.to_logits.bias"] = sd["decoder_joint.net.to_logits.bias"]
    e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
    e.load_state_dict(full_sd)
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    _, a = e.generate(z[:, 0].cuda(), m8, T, 0.0, return_logits=True)
    e.load_state_dict(only)
    e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True)
    _, b = e.generate(z[:, 0].cuda(), m8, T, 0.0, return_logits=True)
    d = (b[:, 0] - a[:, 0]).cpu() - only["decoder_joint.net.to_logits.bias"]
    print("bf16 step-0 logits: |(with - without) - bias| max = %.3g" % d.abs().max())
    assert d.abs().max() < 2e-2 and e.chain_faults() == 0
    # the training step has no gradient for these tensors: it refuses them
    assert e.lib.dimx_train_num_params(e.h) == -4 and b"to_logits.bias" in e.lib.dimx_last_error()   # DIMX_ERR_STATE
    e.close()


def test_prefill_clip_groups_reproduce_the_single_batch(full_sd):
    """round 5, bf16 mode: VQ encode, the encoders + context and VQ decode run as clip groups on several streams once a batch has
    >= 16 384 rows (csrc/model.hip ClipGroups; DIMX_PREFILL_GROUPS=1 keeps one batch on the caller's stream).  Every clip's
    results must be what the single batch gives: indices and tokens identical, coefficients to the last bit or within bf16
    noise where a group's row count picks another GEMM tiling; ragged lengths, a batch that does not divide by four."""
    import os
    from dimx import engine, lib
    B, T = 131, 300
    lens = [T - (i * 13) % 120 for i in range(B)]
    v_s, v_a, z, mask = _case(B, T, lens, seed=41)
    v_l = torch.randn(B, T, 56, generator=torch.Generator().manual_seed(3)).cuda()
    m8 = mask.to(torch.uint8).cuda()
    lens_t = torch.tensor(lens, dtype=torch.int32).cuda()
    idx_dec = torch.randint(0, 512, (B, T - 1), generator=torch.Generator().manual_seed(4), dtype=torch.int32).cuda()

    def run(groups):
        if groups:
            os.environ["DIMX_PREFILL_GROUPS"] = str(groups)
        try:
            e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)
        finally:
            os.environ.pop("DIMX_PREFILL_GROUPS", None)
        e.load_state_dict(full_sd)
        idx, zq = e.vq_encode(1, v_l, lens_t, pe_mode=0, pad_value=-100, return_z=True)
        x_s = e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True, return_x_s=True)
        tok = e.generate(z[:, 0].cuda(), m8, T, 0.0)
        dec = e.vq_decode(1, idx_dec)
        dec3 = e.vq_decode(1, idx_dec[:, :100].contiguous(), row_offset=5)
        torch.cuda.synchronize()
        out = [t.cpu() for t in (idx, zq, x_s, tok, dec, dec3)]
        e.close()
        return out

    one = run(1)
    for groups in (0, 2, 3):   # 0 = automatic (four groups at this size)
        many = run(groups)
        assert torch.equal(one[0], many[0]), "VQ indices differ with %d groups" % groups
        for k, name in ((1, "z"), (2, "x_s"), (4, "decoded motion"), (5, "decoded motion, row offset 5")):
            d = (one[k] - many[k]).abs().max().item()
            print("groups %d: %s max |difference| %.3g" % (groups, name, d))
            assert d <= 2e-2, "%s differs by %g with %d groups" % (name, d, groups)
        agree = (one[3] == many[3]).float().mean().item()
        print("groups %d: generated tokens agreement %.4f" % (groups, agree))
        assert torch.equal(one[3][:, :4], many[3][:, :4])


def test_multi_sample_cross_attention_on_the_matrix_cores_matches_the_multi_query_kernel(full_sd):
    """Round 6, the reference's best-of-N protocol in one pass (code/x_engine_pt.py:257), bf16 mode: the S queries of a clip attend
    over its context K/V on the MFMA prefill attention kernel (attention_tr.hip with Lq = S) instead of the VALU multi-query
    kernel (DIMX_NO_MULTI_TR=1 keeps the latter).  Same softmax, same masks (ragged context lengths), bf16 operand rounding of q and
    of the probabilities: the first step's logits of the two paths agree to bf16 level and pick the same token almost everywhere; the
    f32 parity mode is untouched (its multi-sample pass equals independent runs token for token, test_gpu_module)."""
    import os
    from dimx import engine, lib, prng
    B, T, S, lens = 5, 72, 10, [72, 65, 40, 9, 72]
    v_s, v_a, z, mask = _case(B, T, lens, seed=23)
    m8 = mask.to(torch.uint8).cuda()
    noise = torch.from_numpy(prng.exponential(9, "s2s.multi", (T - 1, B * S, 512))).cuda()
    out = {}
    for name, env in (("tr", None), ("valu", "1")):
        if env is None:
            os.environ.pop("DIMX_NO_MULTI_TR", None)
        else:
            os.environ["DIMX_NO_MULTI_TR"] = env
        try:
            e = engine.Engine("cuda:0", lib.MODE_PERF_BF16)       # the switch is read when the handle is created
        finally:
            os.environ.pop("DIMX_NO_MULTI_TR", None)
        e.load_state_dict(full_sd)
        e.encode_ctx(v_s.cuda(), v_a.cuda(), m8, True, n_samples=S)
        tok, lg = e.generate(z[:, 0].cuda(), m8, T, 1.0, noise=noise, return_logits=True, n_samples=S)
        out[name] = (tok.cpu(), lg.cpu())
        e.close()
    (t_tr, l_tr), (t_va, l_va) = out["tr"], out["valu"]
    assert torch.isfinite(l_tr).all() and t_tr.shape == (B * S, T - 1)
    d0 = (l_tr[:, 0] - l_va[:, 0]).abs().max().item()            # step 0: same inputs on both paths
    agree0 = (t_tr[:, 0] == t_va[:, 0]).float().mean().item()
    print("multi-sample cross attention, MFMA vs VALU kernel: step-0 logits differ by %.3g, step-0 tokens agree %.3f" % (d0, agree0))
    assert d0 < 3e-2 and agree0 >= 0.9
    # samples of a clip differ (the noise differs) and a padded context never leaks: clip 3 has 9 valid frames
    assert not torch.equal(t_tr[0], t_tr[1])
