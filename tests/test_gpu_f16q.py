"""GPU: the f16q operand format (csrc/f16q.h, gemm_f16q.hip) and the opt-in PARITY-GRADE prefill mode built on it
(ChatTSForCausalLM(precision="f16q"), chatts_decoder_set_prefill_f16q): f16 high part on v_mfma_f32_16x16x32_f16 + e4m3 residual x e4m3
weights on the CDNA4 block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE.json configs[4] names that pipe; the reference forwards
quant_config, chatts/vllm/chatts_vllm.py:475,481).  No reference twin for the format (NOT IN REFERENCE): the producers are compared
BIT-exactly with their torch restatement (tests/f16q_ref.py), the GEMM with the float64 product of the dequantised operands (its own
arithmetic: 2e-6) and with the true float32-activation product (the format's price: < 3e-5 per GEMM against bf16x2's 4e-6), the whole
model with the CPU float32 oracle (bar 1e-3, north_star)."""
import numpy as np
import pytest
import torch

from chatts_amd import _lib
from tests import f16q_ref
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _operands(lib, m, n, k, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    w = (torch.randn((n, k), generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    a = torch.randn((m, k), generator=g).to(DEV)
    a[:, ::97] *= 8.0                                            # outlier columns, as real activations have
    hi = torch.empty((m, k), dtype=torch.float16, device=DEV)
    lo = torch.empty((m, k), dtype=torch.uint8, device=DEV)
    sc = torch.empty((m, k // 128), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_split_f16q(a.data_ptr(), m, k, k, hi.data_ptr(), lo.data_ptr(), sc.data_ptr(), k, k // 128, 0, st()))
    w16 = torch.empty((n, k), dtype=torch.float16, device=DEV)
    w8 = torch.empty((n, k), dtype=torch.uint8, device=DEV)
    w8e = torch.empty((n,), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_weights_f16q(w.data_ptr(), n, k, k, w16.data_ptr(), w8.data_ptr(), w8e.data_ptr(), k, st()))
    torch.cuda.synchronize()
    return a, w, (hi, lo, sc), (w16, w8, w8e)


def test_f16q_producers_are_bit_exact(lib):
    a, w, (hi, lo, sc), (w16, w8, w8e) = _operands(lib, 130, 272, 640, seed=1)
    rh, rq, rs = f16q_ref.split(a)
    assert torch.equal(rh.view(torch.int16), hi.view(torch.int16)) and torch.equal(rq, lo) and torch.equal(rs, sc)
    r16, r8, r8e = f16q_ref.weights(w)
    assert torch.equal(r16.view(torch.int16), w16.view(torch.int16)) and torch.equal(r8, w8) and torch.equal(r8e, w8e)
    big = w.float().abs() >= 2.0 ** -17                          # the f16 copy of a bf16 weight is exact down to 2^-17, RNE to 2^-24 steps below
    assert torch.equal(w16.float()[big], w.float()[big]) and float((w16.float() - w.float()).abs().max()) <= 2.0 ** -25
    # RMSNorm written as planes == the split of the float32 RMSNorm
    nw = torch.rand(640, device=DEV) + 0.5
    nh, nl, ns = (torch.empty_like(hi), torch.empty_like(lo), torch.empty_like(sc))
    _lib.check(lib.chatts_rmsnorm_f16q(a.data_ptr(), nw.data_ptr(), nh.data_ptr(), nl.data_ptr(), ns.data_ptr(), 640, 5, 130, 640, 1e-6, 0, st()))
    y = torch.empty_like(a)
    _lib.check(lib.chatts_rmsnorm(a.data_ptr(), nw.data_ptr(), y.data_ptr(), 130, 640, 1e-6, st()))
    torch.cuda.synchronize()
    eh, eq, es = f16q_ref.split(y)
    assert torch.equal(eh.view(torch.int16), nh.view(torch.int16)) and torch.equal(eq, nl) and torch.equal(es, ns)


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(96, 256, 128), (130, 416, 1024), (798, 2080, 640), (360, 7168, 5120), (200, 1024, 13824)])
def test_linear_f16q_parity(lib, epi, m, n, k):
    if epi == _lib.EPI_SWIGLU:
        n = (n + 255) // 256 * 256                   # (the SwiGLU form needs whole 256-row panels: 2 x inter with inter % 128 == 0)
    a, w, (hi, lo, sc), (w16, w8, w8e) = _operands(lib, m, n, k, seed=m + n + k + epi)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    g = torch.Generator(device="cpu").manual_seed(5)
    bias = torch.randn(n, generator=g).to(DEV)
    resid = torch.randn((m, ncols), generator=g).to(DEV)
    out = torch.full((m, ncols), float("nan"), device=DEV)
    wsb = max(int(lib.chatts_linear_f16q_workspace(m, n, k)), 16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    qa = _lib.LinearF16qArgs(a_hi=hi.data_ptr(), a_lo8=lo.data_ptr(), a_scale=sc.data_ptr(), ld_a=k, ld_scale=k // 128, w16=w16.data_ptr(),
                             w8=w8.data_ptr(), w8_exp=w8e.data_ptr(), ldw=k, bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), m=m, n=n, k=k, ldc=ncols, epilogue=epi, workspace=ws.data_ptr(), workspace_bytes=wsb)
    _lib.check(lib.chatts_linear_f16q(qa, st()))
    torch.cuda.synchronize()
    ref = f16q_ref.gemm(hi, lo, sc, w16, w8, w8e) + bias.double()
    true = a.double() @ w.double().t() + bias.double()
    if epi == _lib.EPI_RESID:
        ref, true = ref + resid.double(), true + resid.double()
    if epi == _lib.EPI_SWIGLU:
        def sw(t):
            v = t.view(m, n // 32, 2, 16)
            return (torch.nn.functional.silu(v[:, :, 0]) * v[:, :, 1]).reshape(m, n // 2)
        ref, true = sw(ref), sw(true)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 2e-6          # the kernel against its own arithmetic
    assert rel_err(out.cpu().numpy(), true.cpu().numpy()) < 3e-5         # the format against the float32-activation product
    if epi == _lib.EPI_SWIGLU and n % 256 == 0:                             # plane output == the split of the float32 output
        chi = torch.empty((m, ncols), dtype=torch.float16, device=DEV)
        clo = torch.empty((m, ncols), dtype=torch.uint8, device=DEV)
        csc = torch.empty((m, ncols // 128), dtype=torch.uint8, device=DEV)
        qa.c = None
        qa.c_hi, qa.c_lo8, qa.c_scale, qa.ld_cplanes, qa.ld_cscale = chi.data_ptr(), clo.data_ptr(), csc.data_ptr(), ncols, ncols // 128
        _lib.check(lib.chatts_linear_f16q(qa, st()))
        torch.cuda.synchronize()
        eh, eq, es = f16q_ref.split(out)
        assert torch.equal(eh.view(torch.int16), chi.view(torch.int16)) and torch.equal(eq, clo) and torch.equal(es, csc)


@pytest.mark.parametrize("preset,lengths", [("tiny-qwen2", [256] * 8), ("tiny-qwen3", [256, 100, 256, 64, 256, 256, 17, 256])])
def test_f16q_prefill_mode_matches_oracle(preset, lengths):
    """precision='f16q': a prompt of > 96 rows takes the f16q projections in every layer (chained post-norm planes included); logits of
    sampled prompt rows and the greedy tokens against the CPU float32 oracle - inside the 1e-3 bar, and measurably not the default's bits."""
    from chatts_amd import config as cfgmod, synth
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.processing import ChatTSProcessor
    from oracle import pipeline, synth as osynth
    from oracle.qwen_decoder import QwenOracle
    from tests.util import chat_prompt, random_walk_series
    cfg = cfgmod.preset(preset)
    proc = ChatTSProcessor.from_pretrained(cfg)
    rng = np.random.default_rng(1234)
    series = [random_walk_series(rng, L) for L in lengths]
    inputs = proc(text=[chat_prompt(lengths)], timeseries=series, padding=True, return_tensors="pt")
    ids = inputs["input_ids"][0].tolist()
    sd = osynth.state_dict(synth.all_specs(cfg), 3)
    new = 8
    want = pipeline.generate(cfg, sd, ids, inputs["timeseries"].numpy(), new)
    _, dec = pipeline.split_state_dict(sd)
    ref_logits = QwenOracle(cfg.oracle_dict(), dec).forward_embeds(torch.from_numpy(want["embeds"])).numpy()
    worst = {}
    hid = {}
    for prec in ("f16q", None):
        model = ChatTSForCausalLM.from_synthetic(cfg, seed=3, max_ctx=1024, max_prefill_tokens=1024, precision=prec)
        mm = model.get_multimodal_embeddings(timeseries=inputs["timeseries"], valid_lengths=proc.last_lengths)
        full = model.expand_input_ids(ids, [(L + 15) // 16 for L in lengths])
        assert len(full) >= 96, len(full)
        emb = model.get_input_embeddings(torch.tensor(full), mm)
        hidden = model.forward(inputs_embeds=emb)
        hid[prec] = hidden.clone()
        w = 0.0
        for row in range(0, len(full), max(1, len(full) // 8)):
            w = max(w, rel_err(model.compute_logits(hidden, row=row).cpu().numpy(), ref_logits[row]))
        worst[prec] = w
        out = model.generate(**inputs.to("cuda"), max_new_tokens=new, eos_token_id=[], valid_lengths=proc.last_lengths)
        assert out[0, len(ids):].tolist() == want["tokens"], prec
        del model
    assert worst["f16q"] < 1e-3, worst
    assert worst["f16q"] < 3e-4, worst                                       # what the split delivers (few layers: far below)
    assert not torch.equal(hid["f16q"], hid[None])                           # the mode really ran other arithmetic than the default


def test_f16q_mode_with_packed_multi_prompt_prefill_matches_oracle():
    """precision='f16q' under continuous batching: several prompts prefilled in ONE packed pass of > 96 rows - the attention half of a layer
    runs the default kernels per segment (layer_part0_packed), the MLP half the f16q projections, and the plane buffers change their type in
    between: every request must still produce the oracle's tokens (a stale 'planes already normed' flag across that boundary would not)."""
    from chatts_amd import config as cfgmod, synth
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.processing import ChatTSProcessor
    from oracle import pipeline, synth as osynth
    from tests.util import chat_prompt, random_walk_series
    cfg = cfgmod.preset("tiny-qwen2")
    proc = ChatTSProcessor.from_pretrained(cfg)
    sd = osynth.state_dict(synth.all_specs(cfg), 5)
    rng = np.random.default_rng(21)
    specs = [[256, 256], [100, 256], [256], [64, 64, 64]]
    reqs, wants = [], []
    for lengths in specs:
        series = [random_walk_series(rng, L) for L in lengths]
        inp = proc(text=[chat_prompt(lengths)], timeseries=series, return_tensors="pt")
        ids = inp["input_ids"][0].tolist()
        reqs.append((ids, inp["timeseries"], list(lengths)))
        wants.append(pipeline.generate(cfg, sd, ids, inp["timeseries"].numpy(), 6)["tokens"])
    m = ChatTSForCausalLM.from_synthetic(cfg, seed=5, max_ctx=512, max_prefill_tokens=512, max_batch=4, precision="f16q")
    outs = m.generate_batch(reqs, max_new_tokens=6, eos_token_id=None, sync_every=3)
    assert outs == wants
    assert m.prefix_stats.get("packed_prefills", 0) >= 1


# ---- tiled operands: every LDS-DMA piece of gemm_f16q_kernel is consecutive memory (csrc/f16q.h) ------------------------------------------
def _tile16(t):
    """chatts_tile_bf16's layout for any 16-bit matrix [rows, k] (rows padded to 16 by repeating the last row): block (row / 16, k / 32) =
    1 KB; the 16-byte chunk at position l holds row 16 b + (l >> 2), 8 values from 32 t + 8 ((l & 3) ^ ((l >> 5) << 1))"""
    rows, k = t.shape
    rb = (rows + 15) // 16
    idx = torch.clamp(torch.arange(rb * 16, device=t.device), max=rows - 1)
    x = t[idx].view(rb, 16, k // 32, 4, 8)
    l = torch.arange(64, device=t.device)
    return x.permute(0, 2, 1, 3, 4)[:, :, l >> 2, (l & 3) ^ ((l >> 5) << 1), :].reshape(-1)


def _tile8_rows16(t, rows_valid):
    """the e4m3 ACTIVATION planes: blocks of 16 rows x 32 bytes = 512 B, position 16 h + r holds bytes 16 h .. 16 h + 15 of row r; only
    the first rows_valid rows are defined"""
    rows, k = t.shape
    rb = (rows + 15) // 16
    x = torch.zeros((rb * 16, k), dtype=t.dtype, device=t.device)
    x[:rows] = t
    return x.view(rb, 16, k // 32, 2, 16).permute(0, 2, 3, 1, 4).reshape(-1), rows_valid


def _tile8_rows32(t):
    """chatts_tile_e4m3 (weights): blocks of 32 rows x 32 bytes = 1 KB, position l holds row 32 b + (l >> 5) * 16 + (l & 15), bytes
    32 t + ((l >> 4) & 1) * 16 ..; rows padded by repeating the last"""
    rows, k = t.shape
    rb = (rows + 31) // 32
    idx = torch.clamp(torch.arange(rb * 32, device=t.device), max=rows - 1)
    x = t[idx].view(rb, 2, 16, k // 32, 2, 16)                   # [b, frag, r, t, h, 16]
    return x.permute(0, 3, 1, 4, 2, 5).reshape(-1)                # [b, t, frag, h, r, 16]


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(96, 256, 128), (130, 512, 1024), (798, 2304, 640), (200, 1024, 13824)])
def test_linear_f16q_on_tiled_operands_is_bitwise_the_row_major_result(lib, epi, m, n, k):
    """planes_tiled / w_tiled: the tiled layouts equal their restatements (producers with tiled = 1, chatts_tile_bf16 on the f16 weights,
    chatts_tile_e4m3 on the e4m3 weights); the GEMM's float32 output, its SwiGLU plane output and its post-norm planes are bit-identical to the
    row-major call's (plane outputs compared block by block over the valid rows)."""
    a, w, (hi, lo, sc), (w16, w8, w8e) = _operands(lib, m, n, k, seed=m + n + k + epi + 7)
    m16 = (m + 15) // 16 * 16
    thi = torch.zeros((m16, k), dtype=torch.float16, device=DEV)
    tlo = torch.zeros((m16, k), dtype=torch.uint8, device=DEV)
    tsc = torch.empty_like(sc)
    _lib.check(lib.chatts_split_f16q(a.data_ptr(), m, k, k, thi.data_ptr(), tlo.data_ptr(), tsc.data_ptr(), k, k // 128, 1, st()))
    w16t = torch.empty(int(lib.chatts_tile_bf16_elems(n, k)), dtype=torch.float16, device=DEV)
    w8t = torch.empty(int(lib.chatts_tile_e4m3_bytes(n, k)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_tile_bf16(w16.data_ptr(), n, k, k, w16t.data_ptr(), st()))
    _lib.check(lib.chatts_tile_e4m3(w8.data_ptr(), n, k, k, w8t.data_ptr(), st()))
    torch.cuda.synchronize()
    assert torch.equal(tsc, sc)
    assert torch.equal(_tile16(w16).view(torch.int16), w16t.view(torch.int16)) and torch.equal(_tile8_rows32(w8), w8t)
    # producer planes: compare the blocks of the valid rows (rows >= m of the last block are never written: zeros here)
    ref_hi = _tile16(torch.cat([hi, torch.zeros((m16 - m, k), dtype=hi.dtype, device=DEV)])) if m16 > m else _tile16(hi)
    assert torch.equal(ref_hi.view(torch.int16), thi.view(-1).view(torch.int16))
    assert torch.equal(_tile8_rows16(lo, m)[0], tlo.view(-1))
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    g = torch.Generator(device="cpu").manual_seed(5)
    bias = torch.randn(n, generator=g).to(DEV)
    resid = torch.randn((m, ncols), generator=g).to(DEV)
    nw = (torch.rand(n, generator=g) + 0.5).to(DEV)
    wsb = max(int(lib.chatts_linear_f16q_workspace(m, n, k)), 16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    res = {}
    for tiled in (0, 1):
        out = torch.full((m, ncols), float("nan"), device=DEV)
        rows = m16 if tiled else m
        chi = torch.zeros((rows, ncols), dtype=torch.float16, device=DEV)
        clo = torch.zeros((rows, ncols), dtype=torch.uint8, device=DEV)
        csc = torch.zeros((m, max(1, ncols // 128)), dtype=torch.uint8, device=DEV)
        phi = torch.zeros((rows, n), dtype=torch.float16, device=DEV)
        plo = torch.zeros((rows, n), dtype=torch.uint8, device=DEV)
        psc = torch.zeros((m, max(1, n // 128)), dtype=torch.uint8, device=DEV)
        qa = _lib.LinearF16qArgs(a_hi=(thi if tiled else hi).data_ptr(), a_lo8=(tlo if tiled else lo).data_ptr(), a_scale=sc.data_ptr(), ld_a=k, ld_scale=k // 128,
                                 w16=(w16t if tiled else w16).data_ptr(), w8=(w8t if tiled else w8).data_ptr(), w8_exp=w8e.data_ptr(), ldw=k, bias=bias.data_ptr(),
                                 resid=resid.data_ptr() if epi == _lib.EPI_RESID else None, c=out.data_ptr(), m=m, n=n, k=k, ldc=ncols, epilogue=epi,
                                 workspace=ws.data_ptr(), workspace_bytes=wsb, planes_tiled=tiled, w_tiled=tiled)
        if epi == _lib.EPI_SWIGLU:
            qa.c = None
            qa.c_hi, qa.c_lo8, qa.c_scale, qa.ld_cplanes, qa.ld_cscale = chi.data_ptr(), clo.data_ptr(), csc.data_ptr(), ncols, ncols // 128
        if epi == _lib.EPI_RESID:
            qa.post_norm_w, qa.post_norm_eps = nw.data_ptr(), 1e-6
            qa.post_hi, qa.post_lo8, qa.post_scale, qa.ld_post, qa.ld_pscale = phi.data_ptr(), plo.data_ptr(), psc.data_ptr(), n, n // 128
        _lib.check(lib.chatts_linear_f16q(qa, st()))
        torch.cuda.synchronize()
        res[tiled] = (out, chi, clo, csc, phi, plo, psc)
    (o0, h0, l0, s0, ph0, pl0, ps0), (o1, h1, l1, s1, ph1, pl1, ps1) = res[0], res[1]
    if epi == _lib.EPI_SWIGLU:
        pad = lambda t: torch.cat([t, torch.zeros((m16 - m, t.shape[1]), dtype=t.dtype, device=DEV)]) if m16 > m else t
        assert torch.equal(_tile16(pad(h0)).view(torch.int16), h1.view(-1).view(torch.int16))
        assert torch.equal(_tile8_rows16(l0, m)[0], l1.view(-1)) and torch.equal(s0, s1)
    else:
        assert not torch.isnan(o0).any() and torch.equal(o0.view(torch.int32), o1.view(torch.int32))
    if epi == _lib.EPI_RESID:
        pad = lambda t: torch.cat([t, torch.zeros((m16 - m, t.shape[1]), dtype=t.dtype, device=DEV)]) if m16 > m else t
        assert torch.equal(_tile16(pad(ph0)).view(torch.int16), ph1.view(-1).view(torch.int16))
        assert torch.equal(_tile8_rows16(pl0, m)[0], pl1.view(-1)) and torch.equal(ps0, ps1)
