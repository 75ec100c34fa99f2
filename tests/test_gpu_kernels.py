"""GPU parity tests, kernel by kernel, through the C-ABI (ctypes) against the CPU oracle / float64 torch.

Tolerances: integer / gather work is bit-exact; float32 reductions 2e-5 relative (summation order only:
every weight x activation product is exact in these kernels)."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

from chatts_amd import _lib, synth
from oracle import synth as osynth, ts_embedding as ots
from tests.util import bf16_round, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _lib.load()


def st():
    return _lib.stream_ptr()


# ------------------------------------------------------------------------------------------ fill
@pytest.mark.parametrize("f32", [False, True])
def test_fill_hash_bit_exact(lib, f32):
    spec = synth.TensorSpec("model.layers.3.mlp.up_proj.weight", 300, 1000, 0.0, 13)
    key = osynth.tensor_key(11, spec.name)
    dst = torch.empty((37, 264), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    synth.fill_device(dst, spec, 11, row0=100, col0=40, rows=37, cols=260, ld=264)
    torch.cuda.synchronize()
    bits = osynth.bf16_bits(key, 0.0, 13, 37, 260, row0=100, col0=40, full_cols=1000)
    got = dst[:, :260].float().cpu().numpy()
    want = (bits.astype(np.uint32) << 16).view(np.float32)
    assert np.array_equal(got, want)
    norm = synth.TensorSpec("model.norm.weight", 1, 512, 1.0, 12, True)
    t = synth.fill_device(torch.empty(512, dtype=torch.float32, device=DEV), norm, 5)
    assert np.array_equal(t.cpu().numpy(), osynth.materialize(norm, 5))


def test_fill_hash_64bit_index(lib):
    # rows far enough that i = row*cols + col exceeds 2^32 -> exercises the hi32 term
    spec = synth.TensorSpec("big", 1 << 22, 2048, 0.0, 13)
    dst = torch.empty((4, 2048), dtype=torch.bfloat16, device=DEV)
    synth.fill_device(dst, spec, 1, row0=(1 << 21) + 5, rows=4)
    bits = osynth.bf16_bits(osynth.tensor_key(1, "big"), 0.0, 13, 4, 2048, row0=(1 << 21) + 5, full_cols=2048)
    assert np.array_equal(dst.view(torch.int16).cpu().numpy().view(np.uint16), bits)


# ------------------------------------------------------------------------------------------ TS front end
@pytest.mark.parametrize("name", ["posemb", "posidx", "raw", "single"])
def test_ts_frontend_bit_exact(lib, golden, name):
    from chatts_amd.ts_encoder import TimeSeriesEmbedding
    g = golden("ts_embedding_" + name)
    cfg = json.loads(str(g["config"]))
    enc = TimeSeriesEmbedding(cfg, device=DEV)
    x = torch.from_numpy(g["x"]).to(DEV)
    vl, pc = enc.get_patch_cnt(x)
    assert np.array_equal(pc.cpu().numpy(), g["patch_cnt"]) and pc.dtype == torch.int64
    assert np.array_equal(vl.cpu().numpy(), g["lengths"])
    table = g["w:position_embedding.weight"] if "w:position_embedding.weight" in g.files else None
    want, _ = ots.patch_features(g["x"], cfg, table)
    # run patchify alone through the C-ABI
    lens = g["lengths"].tolist()
    pcs = [(v + 15) // 16 for v in lens]
    off = np.concatenate([[0], np.cumsum(pcs)]).astype(np.int32)
    P = int(off[-1])
    feat = torch.full((P, enc.k0), 7.0, dtype=torch.float32, device=DEV)
    tab = torch.from_numpy(table).to(DEV) if table is not None else None
    pa = _lib.PatchifyArgs(series=x.data_ptr(), row_off=torch.from_numpy(off).to(DEV).data_ptr(),
                           valid_len=vl.data_ptr(), pos_table=_lib.ptr(tab), out=feat.data_ptr(), n_series=len(lens),
                           lmax=x.shape[1] // 2, patch_size=16, mode=enc.mode, emb_dim=enc.embedding_dim,
                           max_seq_len=cfg["max_sequence_length"], max_valid_len=max(lens), total_patches=P,
                           ld_out=enc.k0)
    off_dev = torch.from_numpy(off).to(DEV)
    pa.row_off = off_dev.data_ptr()
    _lib.check(lib.chatts_ts_patchify(pa, st()))
    got = feat.cpu().numpy()
    assert np.array_equal(got[:, :want.shape[1]], want)
    assert np.all(got[:, want.shape[1]:] == 0)


@pytest.mark.parametrize("name", ["posemb", "posidx", "raw", "single"])
def test_ts_encoder_vs_reference_golden(lib, golden, name):
    """Full encoder vs (a) the oracle run with the same bf16-rounded weights (tight) and (b) the reference's own
    float32 output (loose: the only difference is the bf16 rounding of the checkpoint weights)."""
    from chatts_amd.ts_encoder import TimeSeriesEmbedding
    g = golden("ts_embedding_" + name)
    cfg = json.loads(str(g["config"]))
    enc = TimeSeriesEmbedding(cfg, device=DEV)
    w = {k[2:]: g[k] for k in g.files if k.startswith("w:")}
    wr = {k: (bf16_round(v) if k.endswith("weight") and k.startswith("mlp") else v) for k, v in w.items()}
    for k, v in w.items():
        enc.load_tensor(k, torch.from_numpy(v))
    x = torch.from_numpy(g["x"]).to(DEV)
    for lens in (None, g["lengths"].tolist()):
        feats, pc = enc(x, valid_lengths=lens)
        torch.cuda.synchronize()
        want, wpc = ots.ts_embedding_forward(g["x"], cfg, wr)
        assert np.array_equal(pc.cpu().numpy(), wpc)
        assert feats.shape == want.shape
        assert rel_err(feats.cpu().numpy(), want) < 2e-5
        assert rel_err(feats.cpu().numpy(), g["features"]) < 2e-2


@pytest.mark.parametrize("hidden,lengths", [(1024, [17]), (1024, [256]), (1024, [100, 256, 1]), (1024, [256] * 8),
                                            (5120, [256] * 8), (4096, [1024, 64, 1000] * 6)])
def test_ts_encoder_plane_path_equals_float32_path(lib, monkeypatch, hidden, lengths):
    """chatts_ts_encode: P > 1 runs on bf16 hi / lo planes (patchify writes them, every GELU epilogue writes the next
    operand, LDS-DMA GEMM kernels: stream kernel for P <= 16, DMA kernel above); CHATTS_TS_F32_PATH=1 keeps the float32
    register-staged path.  Same products, different tilings -> equal to float32 summation noise; the patchify planes are
    exactly the split of the float32 rows."""
    from chatts_amd.ts_encoder import TimeSeriesEmbedding
    cfg = dict(patch_size=16, num_layers=5, hidden_size=hidden, num_features=2, max_sequence_length=2048,
               use_position_embedding=True, embedding_dim=16)
    enc = TimeSeriesEmbedding(cfg, device=DEV)
    enc.load_synthetic([s for s in synth.ts_encoder_specs(type("C", (), {"ts": cfg})())], 3)
    rng = np.random.default_rng(len(lengths) + hidden)
    lmax = max(lengths)
    x = np.zeros((len(lengths), 2 * lmax, 1), dtype=np.float32)
    for i, L in enumerate(lengths):
        x[i, 0:2 * L:2, 0] = rng.standard_normal(L)
        x[i, 1:2 * L:2, 0] = 1.0
    xd = torch.from_numpy(x).to(DEV)
    a, _ = enc(xd, valid_lengths=lengths)
    monkeypatch.setenv("CHATTS_TS_F32_PATH", "1")
    b, _ = enc(xd, valid_lengths=lengths)
    torch.cuda.synchronize()
    assert a.shape == b.shape == (sum((L + 15) // 16 for L in lengths), hidden)
    assert torch.isfinite(a).all()
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 5e-6
    # patchify alone: planes == split of the float32 rows
    pcs = [(v + 15) // 16 for v in lengths]
    off = torch.tensor(np.concatenate([[0], np.cumsum(pcs)]).astype(np.int32), device=DEV)
    vl = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    P = int(sum(pcs))
    f32 = torch.empty((P, enc.k0), dtype=torch.float32, device=DEV)
    hi = torch.empty((P, enc.k0), dtype=torch.bfloat16, device=DEV)
    lo = torch.empty((P, enc.k0), dtype=torch.bfloat16, device=DEV)
    for planes in (False, True):
        pa = _lib.PatchifyArgs(series=xd.data_ptr(), row_off=off.data_ptr(), valid_len=vl.data_ptr(),
                               pos_table=enc.position_embedding.data_ptr(), out=None if planes else f32.data_ptr(),
                               n_series=len(lengths), lmax=lmax, patch_size=16, mode=1, emb_dim=16, max_seq_len=2048,
                               max_valid_len=lmax, total_patches=P, ld_out=enc.k0, out_hi=hi.data_ptr() if planes else None,
                               out_lo=lo.data_ptr() if planes else None)
        _lib.check(lib.chatts_ts_patchify(pa, st()))
    torch.cuda.synchronize()
    want_hi = f32.to(torch.bfloat16)
    assert torch.equal(hi, want_hi) and torch.equal(lo, (f32 - want_hi.float()).to(torch.bfloat16))


@pytest.mark.parametrize("mode", ["pos_emb", "pos_idx", "raw"])
@pytest.mark.parametrize("hidden,lengths", [(1024, [17]), (1024, [100, 256, 1]), (5120, [256] * 8), (1024, [1000, 64, 1024] * 3), (4096, [33] * 5)])
def test_ts_encoder_fused_layer0_is_bitwise_the_two_launch_form(lib, monkeypatch, mode, hidden, lengths):
    """ts_layer0_kernel (patchify + first MLP layer in one launch: activation fragments built in registers from the series / position
    table, W0 fragments straight from L2, the prefill kernel's MFMA order) against CHATTS_TS_L0_FUSED=0 (feature planes written, layer 0
    through chatts_linear): the encoder output is bit-identical - 2 .. 200 patch rows (16-row block tails, several 128-row tiles), the three
    feature modes of TimeSeriesEmbedding (chatts_vllm.py:61-91)."""
    from chatts_amd.ts_encoder import TimeSeriesEmbedding
    cfg = dict(patch_size=16, num_layers=3, hidden_size=hidden, num_features=2, max_sequence_length=2048,
               use_position_embedding=mode == "pos_emb", use_position_idx=mode == "pos_idx", embedding_dim=16)
    enc = TimeSeriesEmbedding(cfg, device=DEV)
    enc.load_synthetic([s for s in synth.ts_encoder_specs(type("C", (), {"ts": cfg})())], 5)
    rng = np.random.default_rng(len(lengths) + hidden)
    lmax = max(lengths)
    x = np.zeros((len(lengths), 2 * lmax, 1), dtype=np.float32)
    for i, L in enumerate(lengths):
        x[i, 0:2 * L:2, 0] = rng.standard_normal(L)
        x[i, 1:2 * L:2, 0] = 1.0
    xd = torch.from_numpy(x).to(DEV)
    a, _ = enc(xd, valid_lengths=lengths)
    a = a.clone()
    monkeypatch.setenv("CHATTS_TS_L0_FUSED", "0")
    b, _ = enc(xd, valid_lengths=lengths)
    torch.cuda.synchronize()
    assert a.shape == b.shape == (sum((L + 15) // 16 for L in lengths), hidden) and torch.isfinite(a).all()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_ts_encoder_empty_and_errors(lib):
    from chatts_amd.ts_encoder import TimeSeriesEmbedding
    cfg = dict(patch_size=16, num_layers=2, hidden_size=64, num_features=2, max_sequence_length=64,
               use_position_embedding=True, embedding_dim=16)
    enc = TimeSeriesEmbedding(cfg, device=DEV)
    enc.load_synthetic([s for s in synth.ts_encoder_specs(type("C", (), {"ts": cfg})())], 0)
    feats, pc = enc(torch.zeros((2, 64, 1), device=DEV))                 # all-zero mask -> no patches
    assert feats.shape == (0, 64) and pc.tolist() == [0, 0]
    x = torch.ones((1, 2 * 80, 1), device=DEV)                           # 80 valid points > max_sequence_length
    with pytest.raises(IndexError):
        enc(x)
    with pytest.raises(RuntimeError):
        enc(torch.zeros((1, 32, 1)))                                     # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------ linear
def _linear(lib, a, w, bias=None, resid=None, epi=_lib.EPI_NONE, norm_w=None, eps=1e-6):
    m, k = a.shape
    n = w.shape[0]
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((m, ncols), float("nan"), dtype=torch.float32, device=DEV)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=_lib.ptr(bias), resid=_lib.ptr(resid), c=out.data_ptr(),
                         norm_w=_lib.ptr(norm_w), norm_eps=eps, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                         workspace=ws.data_ptr(), workspace_bytes=wsb)
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    return out


def _ref_linear(a, w, bias, resid, epi, norm_w=None, eps=1e-6):
    a64, w64 = a.double().cpu(), w.float().double().cpu()
    if norm_w is not None:
        a64 = norm_w.double().cpu() * (a64 * torch.rsqrt(a64.pow(2).mean(-1, keepdim=True) + eps))
    y = a64 @ w64.T
    if bias is not None:
        y = y + bias.double().cpu()
    if epi == _lib.EPI_GELU:
        y = torch.nn.functional.gelu(y)
    elif epi == _lib.EPI_RESID:
        y = y + resid.double().cpu()
    elif epi == _lib.EPI_SWIGLU:
        n = y.shape[1]
        v = y.view(y.shape[0], n // 32, 2, 16)
        y = (torch.nn.functional.silu(v[:, :, 0]) * v[:, :, 1]).reshape(y.shape[0], n // 2)
    return y.numpy()


def _rand_problem(m, n, k, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((m, k), generator=g) * scale).to(DEV)
    w = (torch.randn((n, k), generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    # asymmetric structure so a transposed / permuted tile cannot pass (guide rule 16)
    w[:, 0] = torch.linspace(-1, 1, n).to(torch.bfloat16)
    bias = torch.randn(n, generator=g).to(DEV)
    resid = torch.randn((m, n), generator=g).to(DEV)
    norm = (1 + 0.1 * torch.randn(k, generator=g)).to(DEV)
    return a, w, bias, resid, norm


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("n,k", [(1024, 512), (5120, 5120), (96, 1024), (5120, 13824), (2048, 8 * 32)])
def test_gemv_parity(lib, epi, norm, n, k):
    a, w, bias, resid, nw = _rand_problem(1, n, k, seed=n + k + epi)
    out = _linear(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi, nw if norm else None)
    want = _ref_linear(a, w, bias, resid, epi, nw if norm else None)
    assert rel_err(out.cpu().numpy(), want) < 2e-5
    assert not torch.isnan(out).any()


# launch geometries the kernel really reads (gemv.hip launch_gemv): rows per wave, chunks in flight, waves per workgroup
# (any 1..16, incl. the non-power-of-two counts of the balanced layouts), grid size, LDS padding
GEMV_GEOMS = [dict(ROWS=2, UNR=2, NW=4), dict(ROWS=4, UNR=2, NW=8), dict(ROWS=2, UNR=4, NW=16), dict(ROWS=4, UNR=4, NW=4),
              dict(ROWS=2, UNR=2, NW=7, BLOCKS=512, LDSPAD=2), dict(ROWS=2, UNR=2, NW=9, BLOCKS=768, LDSPAD=3),
              dict(ROWS=2, UNR=4, NW=5, BLOCKS=3), dict(ROWS=4, UNR=2, NW=11, BLOCKS=256, LDSPAD=1)]


@pytest.mark.parametrize("geom", range(len(GEMV_GEOMS)))
@pytest.mark.parametrize("k", [256, 1600, 3072, 5120, 8192, 10240, 13824])   # chunks per wave: 1,4,6,10,16,20,27
def test_gemv_all_geometries(lib, geom, k, monkeypatch):
    for key, v in GEMV_GEOMS[geom].items():
        monkeypatch.setenv("CHATTS_GEMV_" + key, str(v))
    for epi in (_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU):
        for n in (1504, 96):
            a, w, bias, resid, nw = _rand_problem(1, n, k, seed=geom * 10 + k)
            nob = epi == _lib.EPI_RESID
            out = _linear(lib, a, w, None if nob else bias, resid if nob else None, epi, nw)
            want = _ref_linear(a, w, None if nob else bias, resid, epi, nw)
            assert rel_err(out.cpu().numpy(), want) < 2e-5


def test_gemv_is_deterministic(lib):
    a, w, bias, resid, nw = _rand_problem(1, 5120, 13824, seed=5)
    outs = [_linear(lib, a, w, bias, resid, _lib.EPI_RESID) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(2, 256, 64), (16, 128, 288), (17, 1024, 512), (33, 5120, 288), (70, 384, 1024),
                                   (128, 5120, 5120), (360, 7168, 5120), (200, 1024, 13824), (1, 256, 512)])
def test_gemm_bf16x2_parity(lib, epi, m, n, k):
    if m == 1 and epi != _lib.EPI_GELU:
        pytest.skip("M=1 non-GELU goes to the GEMV kernel (covered above)")
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    out = _linear(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi)
    want = _ref_linear(a, w, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5       # bf16x2: |eps| <= 2^-17 per product, f32 accumulate


def _split_planes(lib, a, ld=None):
    m, k = a.shape
    ld = ld or k
    hi = torch.full((m, ld), float("nan"), dtype=torch.bfloat16, device=DEV)
    lo = torch.full((m, ld), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.chatts_split_bf16x2(a.data_ptr(), m, k, k, hi.data_ptr(), lo.data_ptr(), ld, st()))
    torch.cuda.synchronize()
    return hi, lo


def test_split_bf16x2_bit_exact(lib):
    g = torch.Generator().manual_seed(3)
    a = (torch.randn((37, 320), generator=g) * torch.logspace(-20, 20, 320)).to(DEV)    # wide exponent range
    a[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, 65504.0])
    hi, lo = _split_planes(lib, a, ld=328)
    want_hi = a.to(torch.bfloat16)
    want_lo = (a - want_hi.float()).to(torch.bfloat16)
    assert torch.equal(hi[:, :320], want_hi) and torch.equal(lo[:, :320], want_lo)
    # hi + lo carries 16 mantissa bits of a
    rec = hi[:, :320].float().double() + lo[:, :320].float().double()
    ok = a.abs() > 1e-30
    assert ((rec - a.double()).abs()[ok] <= a.double().abs()[ok] * 2.0 ** -16).all()


def _linear_planes(lib, a, w, bias, resid, epi, with_a=True, ld=None):
    m, k = a.shape
    n = w.shape[0]
    hi, lo = _split_planes(lib, a, ld)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((m, ncols), float("nan"), dtype=torch.float32, device=DEV)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr() if with_a else None, w=w.data_ptr(), bias=_lib.ptr(bias), resid=_lib.ptr(resid),
                         c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                         workspace=ws.data_ptr(), workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(),
                         ld_planes=ld or k)
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(96, 256, 64), (128, 5120, 5120), (130, 384, 1024), (360, 7168, 5120), (200, 1024, 13824),
                                   (798, 2080, 640), (513, 96, 192)])
def test_gemm_dma_parity(lib, epi, m, n, k):
    """LDS-DMA GEMM on pre-split planes (ragged M and N tiles, K = 1..216 steps of 64, split-K) vs float64."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    out = _linear_planes(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi, with_a=False, ld=k + 64)
    want = _ref_linear(a, w, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(96, 256, 64), (130, 384, 1024), (798, 2080, 640), (360, 7168, 5120)])
def test_gemm_dma_single_pass_speed_mode(lib, monkeypatch, epi, m, n, k):
    """CHATTS_GEMM_PRECISION=bf16 (the optional speed mode): the LDS-DMA GEMM multiplies the hi plane only - exactly the product of
    the bf16-ROUNDED activations with the weights (float64 reference on the rounded operand: 2e-5), and measurably not the
    float32-activation product the default mode delivers."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi + 5, scale=3.0)
    want_exact = _ref_linear(a, w, bias, resid, epi)
    want_rounded = _ref_linear(a.to(torch.bfloat16).float(), w, bias, resid, epi)
    monkeypatch.setenv("CHATTS_GEMM_PRECISION", "bf16")
    out = _linear_planes(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi, with_a=False, ld=k + 64)
    monkeypatch.delenv("CHATTS_GEMM_PRECISION")
    ref = _linear_planes(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi, with_a=False, ld=k + 64)
    assert rel_err(out.cpu().numpy(), want_rounded) < 2e-5
    assert rel_err(ref.cpu().numpy(), want_exact) < 2e-5
    assert 1e-4 < rel_err(out.cpu().numpy(), want_exact) < 2e-2          # the price of the mode: ~2^-9 per operand


def _tile_ref(t):
    """chatts_tile_bf16's layout restated with torch indexing (include/chatts_amd.h): block (b, t) = 1 KB, chunk position l holds row
    16 b + (l >> 2), K-chunk (l & 3) ^ ((l >> 5) << 1); rows beyond the matrix repeat the last one."""
    rows, k = t.shape
    rb = (rows + 15) // 16
    idx = torch.clamp(torch.arange(rb * 16, device=t.device), max=rows - 1)
    x = t[idx].view(rb, 16, k // 32, 4, 8)
    l = torch.arange(64, device=t.device)
    return x.permute(0, 2, 1, 3, 4)[:, :, l >> 2, (l & 3) ^ ((l >> 5) << 1), :].reshape(-1)


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(96, 256, 64), (130, 416, 1024), (798, 2080, 640), (360, 7168, 5120), (200, 1024, 13824)])
def test_gemm_ring_on_tiled_operands_is_bitwise_the_row_major_result(lib, epi, m, n, k):
    """chatts_tile_bf16 + ChattsLinearArgs.w_tiled / planes_tiled: the prefill kernel's LDS-DMA pieces read 1 KB of consecutive memory
    instead of 16 rows x 64 bytes.  The tiled layout equals its restatement; the products and their order are unchanged, so every output
    is bit-identical to the row-major call (ragged M and N: the last 16-row block repeats the last row, its outputs are never stored)."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi + 11, scale=3.0)
    hi, lo = _split_planes(lib, a, None)

    def tile(t):
        out = torch.empty(int(lib.chatts_tile_bf16_elems(t.shape[0], t.shape[1])), dtype=torch.bfloat16, device=DEV)
        _lib.check(lib.chatts_tile_bf16(t.data_ptr(), t.shape[0], t.shape[1], t.shape[1], out.data_ptr(), st()))
        return out
    wt, hit, lot = tile(w), tile(hi), tile(lo)
    for src, got in ((w, wt), (hi, hit)):
        assert torch.equal(_tile_ref(src).view(torch.int16), got.view(torch.int16))
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    outs = []
    for mode in range(3):                       # row-major | W tiled | W and planes tiled
        out = torch.full((m, ncols), float("nan"), dtype=torch.float32, device=DEV)
        la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=_lib.ptr(bias), resid=_lib.ptr(resid if epi == _lib.EPI_RESID else None),
                             c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                             workspace=ws.data_ptr(), workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
        if mode >= 1:
            la.w_tiled = wt.data_ptr()
        if mode == 2:
            la.a_hi, la.a_lo, la.planes_tiled = hit.data_ptr(), lot.data_ptr(), 1
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        outs.append(out)
    assert not torch.isnan(outs[0]).any()
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
    assert torch.equal(outs[0].view(torch.int32), outs[2].view(torch.int32))
    assert rel_err(outs[1].cpu().numpy(), _ref_linear(a, w, bias, resid, epi)) < 2e-5


def test_tiled_planes_are_refused_outside_the_prefill_kernel(lib):
    a, w, bias, resid, _ = _rand_problem(8, 256, 256, seed=5, scale=1.0)
    hi, lo = _split_planes(lib, a, None)
    out = torch.zeros((8, 256), dtype=torch.float32, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), c=out.data_ptr(), m=8, n=256, k=256, lda=256, ldw=256, ldc=256, epilogue=_lib.EPI_NONE,
                         workspace=ws.data_ptr(), workspace_bytes=ws.numel(), a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=256, planes_tiled=1)
    assert lib.chatts_linear(la, st()) == _lib.E_SHAPE and b"tiled planes" in lib.chatts_last_error()
    assert lib.chatts_tile_bf16(w.data_ptr(), 256, 250, 256, out.data_ptr(), st()) == _lib.E_SHAPE


@pytest.mark.parametrize("sk", [1, 2, 3])
@pytest.mark.parametrize("tiles", [0, 1, 3])
def test_gemm_ring_equals_register_staged_bitwise(lib, sk, tiles, monkeypatch):
    """Same products, same accumulation order: with the split-K factor pinned the register-staged kernel and the prefill kernel
    (gemm_ring_kernel: swapped MFMA operands, K walked in 32-deep half-stages, any number of M-tiles) agree bit for bit."""
    monkeypatch.setenv("CHATTS_GEMM_SK", str(sk))
    monkeypatch.setenv("CHATTS_GEMM_BM", "128")
    if tiles:
        monkeypatch.setenv("CHATTS_GEMM_T", str(tiles))
    for epi, (m, n, k) in ((_lib.EPI_RESID, (300, 640, 1536)), (_lib.EPI_SWIGLU, (257, 1024, 768)), (_lib.EPI_NONE, (144, 272, 384))):
        a, w, bias, resid, _ = _rand_problem(m, n, k, seed=sk + m, scale=2.0)
        r = resid if epi == _lib.EPI_RESID else None
        ref = _linear(lib, a, w, bias, r, epi)
        assert torch.equal(_linear_planes(lib, a, w, bias, r, epi), ref)


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k,planes_in", [(300, 640, 1536, True), (130, 1024, 768, True), (40, 512, 256, False),
                                             (17, 256, 2048, False)])
def test_linear_plane_output_equals_split_of_f32_output(lib, epi, m, n, k, planes_in):
    """c_hi / c_lo: the epilogue writes what chatts_split_bf16x2 would make of the float32 result (direct and split-K)."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + k + epi, scale=2.0)
    r = resid if epi == _lib.EPI_RESID else None
    want = _linear_planes(lib, a, w, bias, r, epi) if planes_in else _linear(lib, a, w, bias, r, epi)
    want_hi, want_lo = _split_planes(lib, want)
    ncols = want.shape[1]
    hi = torch.full((m, ncols + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    lo = torch.full((m, ncols + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    ahi, alo = _split_planes(lib, a)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=_lib.ptr(bias), resid=_lib.ptr(r), c=None, norm_w=None,
                         norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi, workspace=ws.data_ptr(),
                         workspace_bytes=wsb, c_hi=hi.data_ptr(), c_lo=lo.data_ptr(), ld_cplanes=ncols + 8)
    if planes_in:
        la.a_hi, la.a_lo, la.ld_planes = ahi.data_ptr(), alo.data_ptr(), k
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    assert torch.equal(hi[:, :ncols], want_hi) and torch.equal(lo[:, :ncols], want_lo)
    assert torch.isnan(hi[:, ncols:].float()).all()                      # padding columns untouched


def test_rmsnorm_planes_equals_split_of_rmsnorm(lib):
    g = torch.Generator().manual_seed(11)
    t, h = 37, 5120
    x = (torch.randn((t, h), generator=g) * 3).to(DEV)
    w = (1 + 0.1 * torch.randn(h, generator=g)).to(DEV)
    y = torch.empty_like(x)
    _lib.check(lib.chatts_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), t, h, 1e-6, st()))
    hi = torch.full((t, h + 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    lo = torch.full((t, h + 64), float("nan"), dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.chatts_rmsnorm_planes(x.data_ptr(), w.data_ptr(), hi.data_ptr(), lo.data_ptr(), h + 64, t, h, 1e-6, st()))
    torch.cuda.synchronize()
    want_hi, want_lo = _split_planes(lib, y)
    assert torch.equal(hi[:, :h], want_hi) and torch.equal(lo[:, :h], want_lo)


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(2, 128, 64), (3, 160, 320), (7, 5120, 5120), (16, 7168, 5120), (16, 1024, 13824),
                                   (5, 152064 // 8, 512), (16, 96, 128)])
@pytest.mark.parametrize("stages", [3, 4, 5])
def test_gemm_stream_parity(lib, epi, m, n, k, stages, monkeypatch):
    """2 <= M <= 16 on planes (batched decode): whole-line LDS-DMA streaming kernel; every epilogue, ragged N tiles,
    splits shorter than the ring, all ring depths."""
    monkeypatch.setenv("CHATTS_GEMM_STREAM_STAGES", str(stages))
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    out = _linear_planes(lib, a, w, bias, resid if epi == _lib.EPI_RESID else None, epi, with_a=False, ld=k + 64)
    want = _ref_linear(a, w, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


def test_gemm_stream_rows_do_not_depend_on_the_batch(lib):
    """Continuous batching invariant: a sequence's projection is bit-identical whichever other rows share the launch."""
    a, w, bias, resid, _ = _rand_problem(16, 5120, 5120, seed=77, scale=2.0)
    full = _linear_planes(lib, a, w, bias, None, _lib.EPI_NONE, with_a=False)
    assert torch.equal(full, _linear_planes(lib, a, w, bias, None, _lib.EPI_NONE, with_a=False))
    for rows in ([0, 1], [3, 9, 15], list(range(5))):
        sub = _linear_planes(lib, a[rows].contiguous(), w, bias, None, _lib.EPI_NONE, with_a=False)
        assert torch.equal(sub, full[rows])


def test_gemm_dma_small_m_and_odd_k_fall_back(lib):
    a, w, bias, resid, _ = _rand_problem(40, 256, 512, seed=9)
    assert torch.equal(_linear_planes(lib, a, w, bias, None, _lib.EPI_NONE), _linear(lib, a, w, bias, None, _lib.EPI_NONE))
    a, w, bias, resid, _ = _rand_problem(200, 256, 96, seed=10)          # K % 64 != 0: the planes are ignored
    assert torch.equal(_linear_planes(lib, a, w, bias, None, _lib.EPI_NONE), _linear(lib, a, w, bias, None, _lib.EPI_NONE))
    with pytest.raises(_lib.ChattsError):                                 # ... and without `a` there is nothing to fall back on
        _linear_planes(lib, a, w, bias, None, _lib.EPI_NONE, with_a=False)


def test_linear_argument_errors(lib):
    a = torch.zeros((2, 48), device=DEV)
    w = torch.zeros((32, 48), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.ChattsError) as e:
        _linear(lib, a, w)
    assert e.value.code == _lib.E_SHAPE and "multiple of 32" in e.value.msg


# ------------------------------------------------------------------------------------------ elementwise
def test_rmsnorm(lib):
    x = torch.randn((37, 5120), device=DEV) * 3
    w = 1 + 0.1 * torch.randn(5120, device=DEV)
    y = torch.empty_like(x)
    _lib.check(lib.chatts_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), 37, 5120, 1e-6, st()))
    x64 = x.double().cpu()
    want = w.double().cpu() * (x64 * torch.rsqrt(x64.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert rel_err(y.cpu().numpy(), want.numpy()) < 1e-6


def _rope_tables(max_pos, theta=1e6):
    from oracle.qwen_decoder import rope_cos_sin
    cos, sin = rope_cos_sin(torch.arange(max_pos), 128, theta)
    return cos[:, :64].contiguous().to(DEV), sin[:, :64].contiguous().to(DEV)


@pytest.mark.parametrize("qk_norm", [False, True])
def test_rope_kv_write(lib, qk_norm):
    from oracle.qwen_decoder import rms_norm, rope_cos_sin, rotate_half
    T, nq, nkv, pos0, max_ctx = 9, 4, 2, 5, 32
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), device=DEV)
    orig = qkv.clone().cpu()
    qn = (1 + 0.1 * torch.randn(128)).to(DEV) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(128)).to(DEV) if qk_norm else None
    cos, sin = _rope_tables(64)
    kc = torch.zeros((nkv, max_ctx, 128), device=DEV)
    vc = torch.zeros((nkv, max_ctx, 128), device=DEV)
    cache = _lib.KvCache(k=kc.data_ptr(), v=vc.data_ptr(), max_ctx=max_ctx)
    _lib.check(lib.chatts_rope_kv_write(qkv.data_ptr(), T, nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6, cos.data_ptr(),
                                        sin.data_ptr(), pos0, None, C.byref(cache), st()))
    torch.cuda.synchronize()
    o = orig.view(T, nq + 2 * nkv, 128)
    q, k, v = o[:, :nq], o[:, nq:nq + nkv], o[:, nq + nkv:]
    if qk_norm:
        q, k = rms_norm(q, qn.cpu(), 1e-6), rms_norm(k, kn.cpu(), 1e-6)
    c, s = rope_cos_sin(torch.arange(pos0, pos0 + T), 128, 1e6)
    q = q * c[:, None] + rotate_half(q) * s[:, None]
    k = k * c[:, None] + rotate_half(k) * s[:, None]
    got = qkv.cpu().view(T, nq + 2 * nkv, 128)
    assert rel_err(got[:, :nq].numpy(), q.numpy()) < 1e-6
    assert rel_err(kc[:, pos0:pos0 + T].cpu().numpy(), k.transpose(0, 1).numpy()) < 1e-6
    assert torch.equal(vc[:, pos0:pos0 + T].cpu(), v.transpose(0, 1))
    assert torch.all(kc[:, :pos0] == 0) and torch.all(kc[:, pos0 + T:] == 0)
    # device-resident position (graph replay path)
    pos_dev = torch.tensor([pos0], dtype=torch.int32, device=DEV)
    qkv2 = orig.clone().to(DEV)
    kc2, vc2 = torch.zeros_like(kc), torch.zeros_like(vc)
    cache2 = _lib.KvCache(k=kc2.data_ptr(), v=vc2.data_ptr(), max_ctx=max_ctx)
    _lib.check(lib.chatts_rope_kv_write(qkv2.data_ptr(), T, nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6, cos.data_ptr(),
                                        sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(cache2), st()))
    assert torch.equal(kc2, kc) and torch.equal(qkv2, qkv)


def _ref_attention(q, kc, vc, pos0):
    """q [T,nq,128] (rotated), caches [nkv, ctx, 128]; float64 eager attention with causal mask (one masked matrix product per head)."""
    T, nq, _ = q.shape
    nkv = kc.shape[0]
    g = nq // nkv
    n_keys = pos0 + T
    out = torch.zeros((T, nq, 128), dtype=torch.float64)
    future = torch.arange(n_keys)[None, :] > (pos0 + torch.arange(T))[:, None]          # key j is visible to row t iff j <= pos0 + t
    for h in range(nq):
        k = kc[h // g, :n_keys].double()
        v = vc[h // g, :n_keys].double()
        s = (q[:, h].double() @ k.t()) / np.sqrt(128.0)
        s = s.masked_fill(future, float("-inf"))
        out[:, h] = torch.softmax(s, dim=1) @ v
    return out


@pytest.mark.parametrize("nq,nkv", [(4, 2), (5, 1), (8, 1), (8, 8)])
@pytest.mark.parametrize("T,pos0,splits", [(1, 0, 1), (1, 63, 4), (1, 200, 4), (1, 333, 16), (7, 60, 1), (70, 0, 1),
                                           (16, 0, 1), (64, 0, 1), (65, 31, 1), (130, 100, 1), (200, 0, 1)])
def test_attention_parity(lib, nq, nkv, T, pos0, splits):
    max_ctx = 512
    g = torch.Generator().manual_seed(nq * 100 + T + pos0)
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), generator=g)
    kc = torch.randn((nkv, max_ctx, 128), generator=g)
    vc = torch.randn((nkv, max_ctx, 128), generator=g)
    kc[:, 17] *= 4.0     # a spiky key so the online-softmax rescale path is exercised
    want = _ref_attention(qkv.view(T, nq + 2 * nkv, 128)[:, :nq], kc, vc, pos0)
    qd, kd, vd = qkv.to(DEV), kc.to(DEV), vc.to(DEV)
    out = torch.full((T, nq * 128), float("nan"), device=DEV)
    cache = _lib.KvCache(k=kd.data_ptr(), v=vd.data_ptr(), max_ctx=max_ctx)
    wsb = int(lib.chatts_attn_workspace(T, nq, splits))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    _lib.check(lib.chatts_attention(qd.data_ptr(), T, nq, nkv, pos0, None, C.byref(cache), out.data_ptr(), splits,
                                    ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy().reshape(T, nq, 128), want.numpy()) < 1e-5


@pytest.mark.parametrize("qk_norm", [False, True])
@pytest.mark.parametrize("pos,splits", [(0, 1), (5, 4), (15, 2), (16, 2), (63, 4), (64, 4), (200, 3), (333, 16), (130, 32), (500, 64), (511, 7)])
def test_attention_decode_fused_equals_unfused(lib, qk_norm, pos, splits):
    """fused kernel (norm + RoPE + cache write + attention) == rope_kv_write followed by attention"""
    nq, nkv, max_ctx = 10, 2, 512
    g = torch.Generator().manual_seed(pos * 7 + splits)
    raw = torch.randn((1, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    kc0 = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    vc0 = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    kc0[:, pos:] = float("nan")      # rows >= pos are not part of the context yet: must never leak into the result
    vc0[:, pos:] = float("nan")
    qn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    cos, sin = _rope_tables(max_ctx)
    wsb = int(lib.chatts_attn_workspace(1, nq, splits))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    # reference: two-kernel path
    qkv_a, kc_a, vc_a = raw.clone(), kc0.clone(), vc0.clone()
    ca = _lib.KvCache(k=kc_a.data_ptr(), v=vc_a.data_ptr(), max_ctx=max_ctx)
    out_a = torch.empty((1, nq * 128), device=DEV)
    _lib.check(lib.chatts_rope_kv_write(qkv_a.data_ptr(), 1, nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6, cos.data_ptr(),
                                        sin.data_ptr(), pos, None, C.byref(ca), st()))
    _lib.check(lib.chatts_attention(qkv_a.data_ptr(), 1, nq, nkv, pos, None, C.byref(ca), out_a.data_ptr(), splits,
                                    ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    # fused, position read from the device
    kc_b, vc_b = kc0.clone(), vc0.clone()
    cb = _lib.KvCache(k=kc_b.data_ptr(), v=vc_b.data_ptr(), max_ctx=max_ctx)
    out_b = torch.full((1, nq * 128), float("nan"), device=DEV)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    _lib.check(lib.chatts_attention_decode_fused(raw.data_ptr(), nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6,
                                                 cos.data_ptr(), sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(cb),
                                                 out_b.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    assert not torch.isnan(out_b).any()
    assert rel_err(kc_b[:, pos].cpu().numpy(), kc_a[:, pos].cpu().numpy()) < 1e-6 and torch.equal(vc_b[:, pos], vc_a[:, pos])
    assert torch.equal(kc_b[:, :pos], kc0[:, :pos])
    assert rel_err(out_b.cpu().numpy(), out_a.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("vocab,stride", [(152064, 152064), (151936, 152064), (19008, 19010), (4099, 4100)])
def test_argmax_two_launch_form_picks_the_same_token(lib, vocab, stride):
    """chatts_argmax_batched_ws (64 partial workgroups per row + a merging wave) == chatts_argmax_batched == torch.argmax: maxima at
    block edges, exact ties (first index wins), a row of equal values, unaligned rows (odd stride), parked / saturating positions."""
    B = 5
    g = torch.Generator().manual_seed(vocab)
    logits = torch.randn((B, stride), generator=g).to(DEV)
    chunk = ((vocab + 63) // 64 + 3) // 4 * 4
    logits[0, chunk - 1] = 50.0; logits[0, chunk] = 50.0; logits[0, vocab - 1] = 50.0       # tie across a block edge and at the end
    logits[1, vocab - 1] = 60.0                                                            # last element
    logits[2, :vocab] = 1.25                                                               # all equal -> index 0
    logits[3, 7 * chunk + 3] = 70.0; logits[3, vocab:] = 99.0                               # padding beyond vocab must not win
    ws = torch.empty(int(lib.chatts_argmax_workspace(B)), dtype=torch.uint8, device=DEV)
    outs = []
    for use_ws in (False, True):
        tok = torch.zeros(B, dtype=torch.int64, device=DEV)
        val = torch.zeros(B, device=DEV)
        outt = torch.full((B, 4), -1, dtype=torch.int64, device=DEV)
        step = torch.tensor([0, 1, 2, 3, 0], dtype=torch.int32, device=DEV)
        pos = torch.tensor([5, -1, 9, 10, 3], dtype=torch.int32, device=DEV)
        _lib.check(lib.chatts_argmax_batched_ws(logits.data_ptr(), B, stride, vocab, 1000, tok.data_ptr(), val.data_ptr(), outt.data_ptr(), 4,
                                                step.data_ptr(), pos.data_ptr(), 10, ws.data_ptr() if use_ws else None,
                                                ws.numel() if use_ws else 0, st()))
        torch.cuda.synchronize()
        outs.append((tok.cpu(), val.cpu(), outt.cpu(), step.cpu(), pos.cpu()))
    want = torch.argmax(logits[:, :vocab], dim=1).cpu() + 1000
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][0], want)
    assert outs[1][0][0] == 1000 + chunk - 1 and outs[1][0][2] == 1000
    assert outs[1][4].tolist() == [6, -1, 10, 10, 4] and outs[1][3].tolist() == [1, 2, 3, 4, 1]


def test_embed_merge_and_argmax(lib):
    V, H, T = 1000, 256, 50
    table = torch.randn((V, H)).to(torch.bfloat16).to(DEV)
    ts_id = 777
    ids = torch.randint(0, 700, (T,), dtype=torch.int64)
    where = [3, 4, 5, 20, 49]
    ids[where] = ts_id
    rows = torch.randn((len(where), H), device=DEV)
    out = torch.empty((T, H), device=DEV)
    scan = torch.empty(T + 1, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ids_dev = ids.to(DEV)
    _lib.check(lib.chatts_embed_merge(ids_dev.data_ptr(), ids.data_ptr(), T, table.data_ptr(), V, H, rows.data_ptr(),
                                      len(where), ts_id, out.data_ptr(), scan.data_ptr(), status.data_ptr(), st()))
    torch.cuda.synchronize()
    want = table.float()[ids_dev]
    want[where] = rows
    assert torch.equal(out, want) and status.item() == 0
    # device-side mismatch flag when the host copy of the ids is not supplied
    _lib.check(lib.chatts_embed_merge(ids_dev.data_ptr(), None, T, table.data_ptr(), V, H, rows.data_ptr(), 4, ts_id,
                                      out.data_ptr(), scan.data_ptr(), status.data_ptr(), st()))
    assert status.item() == 1
    with pytest.raises(ValueError):
        _lib.check(lib.chatts_embed_merge(ids_dev.data_ptr(), ids.data_ptr(), T, table.data_ptr(), V, H,
                                          rows.data_ptr(), 4, ts_id, out.data_ptr(), scan.data_ptr(), None, st()))
    # a long sequence crosses the 1024-wide scan chunk
    T2 = 3000
    ids2 = torch.randint(0, 700, (T2,), dtype=torch.int64)
    sel = torch.rand(T2) < 0.3
    ids2[sel] = ts_id
    rows2 = torch.randn((int(sel.sum()), H), device=DEV)
    out2 = torch.empty((T2, H), device=DEV)
    scan2 = torch.empty(T2 + 1, dtype=torch.int32, device=DEV)
    _lib.check(lib.chatts_embed_merge(ids2.to(DEV).data_ptr(), ids2.data_ptr(), T2, table.data_ptr(), V, H,
                                      rows2.data_ptr(), rows2.shape[0], ts_id, out2.data_ptr(), scan2.data_ptr(), None, st()))
    want2 = table.float()[ids2.to(DEV)]
    want2[sel.to(DEV)] = rows2
    assert torch.equal(out2, want2)
    # argmax: first index of the maximum
    logits = torch.randn(152064, device=DEV)
    logits[[5000, 90000]] = 50.0
    tok = torch.zeros(1, dtype=torch.int64, device=DEV)
    val = torch.zeros(1, device=DEV)
    outt = torch.zeros(8, dtype=torch.int64, device=DEV)
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    pos = torch.tensor([10], dtype=torch.int32, device=DEV)
    _lib.check(lib.chatts_argmax(logits.data_ptr(), 152064, 1000, tok.data_ptr(), val.data_ptr(), outt.data_ptr(),
                                 step.data_ptr(), pos.data_ptr(), st()))
    assert tok.item() == 6000 and val.item() == 50.0 and outt[2].item() == 6000 and step.item() == 3 and pos.item() == 11


def test_attention_decode_batched_equals_per_sequence(lib):
    """batched decode attention (one position / cache slot per sequence) == the single-sequence call per slot"""
    nq, nkv, max_ctx, L, B, splits = 10, 2, 256, 3, 4, 16
    g = torch.Generator().manual_seed(42)
    raw = torch.randn((B, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    kc0 = torch.randn((B, L, nkv, max_ctx, 128), generator=g).to(DEV)
    vc0 = torch.randn((B, L, nkv, max_ctx, 128), generator=g).to(DEV)
    pos = [0, 37, 200, 16]
    layer = 1
    cos, sin = _rope_tables(max_ctx)
    wsb = int(lib.chatts_attn_workspace(B, nq, splits))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    seq_stride = L * nkv * max_ctx * 128
    kc_b, vc_b = kc0.clone(), vc0.clone()
    cache = _lib.KvCache(k=kc_b[0, layer].data_ptr(), v=vc_b[0, layer].data_ptr(), max_ctx=max_ctx)
    out_b = torch.full((B, nq * 128), float("nan"), device=DEV)
    pos_dev = torch.tensor(pos, dtype=torch.int32, device=DEV)
    _lib.check(lib.chatts_attention_decode_batched(raw.data_ptr(), B, nq, nkv, None, None, 1e-6, cos.data_ptr(),
                                                   sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(cache), seq_stride,
                                                   out_b.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    for b in range(B):
        kc_a, vc_a = kc0[b, layer].clone(), vc0[b, layer].clone()
        ca = _lib.KvCache(k=kc_a.data_ptr(), v=vc_a.data_ptr(), max_ctx=max_ctx)
        out_a = torch.empty((1, nq * 128), device=DEV)
        _lib.check(lib.chatts_attention_decode_fused(raw[b:b + 1].contiguous().data_ptr(), nq, nkv, None, None, 1e-6,
                                                     cos.data_ptr(), sin.data_ptr(), pos[b], None, C.byref(ca),
                                                     out_a.data_ptr(), splits, ws.data_ptr(), wsb, st()))
        torch.cuda.synchronize()
        assert torch.equal(out_b[b], out_a[0])
        assert torch.equal(kc_b[b, layer], kc_a) and torch.equal(vc_b[b, layer], vc_a)
    # other layers / untouched
    assert torch.equal(kc_b[:, 0], kc0[:, 0]) and torch.equal(kc_b[:, 2], kc0[:, 2])


@pytest.mark.parametrize("group", [4, 5])
@pytest.mark.parametrize("qk_norm", [False, True])
def test_attention_decode_with_the_group_size_compiled_in_is_bitwise_the_run_time_form(lib, monkeypatch, group, qk_norm):
    """attn_decode_kernel<G, EXACT> (round 5: `g < gn` decided by the compiler, one straight-line tile body) == the run-time form
    (ATTN_EXACT=0): outputs and the new cache rows bit for bit, for slots of one tile and of several (the prefetching ping-pong),
    positions at tile edges, a parked sequence."""
    nkv, max_ctx, B = 2, 640, 6
    nq = nkv * group
    g = torch.Generator().manual_seed(group * 3 + qk_norm)
    raw = torch.randn((B, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    kc0 = torch.randn((B, nkv, max_ctx, 128), generator=g).to(DEV)
    vc0 = torch.randn((B, nkv, max_ctx, 128), generator=g).to(DEV)
    qn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    cos, sin = _rope_tables(max_ctx)
    pos_dev = torch.tensor([0, 15, 16, 333, 639, -1], dtype=torch.int32, device=DEV)
    for splits in (1, 3, 16, 40):
        wsb = int(lib.chatts_attn_workspace(B, nq, splits))
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        runs = []
        for exact in ("0", "1"):
            monkeypatch.setenv("CHATTS_ATTN_EXACT", exact)
            kc, vc = kc0.clone(), vc0.clone()
            cache = _lib.KvCache(k=kc.data_ptr(), v=vc.data_ptr(), max_ctx=max_ctx)
            out = torch.full((B, nq * 128), float("nan"), device=DEV)
            _lib.check(lib.chatts_attention_decode_batched(raw.data_ptr(), B, nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6, cos.data_ptr(),
                                                           sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(cache), nkv * max_ctx * 128,
                                                           out.data_ptr(), splits, ws.data_ptr(), wsb, st()))
            torch.cuda.synchronize()
            runs.append((out, kc, vc))
        (o0, k0, v0), (o1, k1, v1) = runs
        assert not torch.isnan(o1[:5]).any() and torch.equal(o0[:5], o1[:5])
        assert torch.equal(k0, k1) and torch.equal(v0, v1)
        assert torch.equal(k1[5], kc0[5]) and not torch.equal(k1[3], kc0[3])


@pytest.mark.parametrize("sk", [2, 5, 8])
@pytest.mark.parametrize("m,n,k,epi", [(16, 5120, 5120, _lib.EPI_RESID), (16, 8192, 2048, _lib.EPI_NONE), (7, 2080, 2048, _lib.EPI_RESID),
                                       (3, 3072, 1024, _lib.EPI_RESID), (16, 4112, 1024, _lib.EPI_NONE)])
def test_few_row_post_norm_epilogue_three_column_groups_per_trip_is_bitwise_one_per_trip(lib, monkeypatch, sk, m, n, k, epi):
    """splitk_epilogue_norm_q_kernel<8, 3> (round 5: slabs / residual of three 1024-column groups requested together) == <8, 1>
    (EPI_NORM_Q_GROUPS=1): c and both planes bit for bit, rows that end inside a batch of groups, at its edge, and past 7 groups."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + sk, scale=2.0)
    g = torch.Generator().manual_seed(7)
    nw = (1 + 0.1 * torch.randn(n, generator=g)).to(DEV)
    hi, lo = _split_planes(lib, a)
    monkeypatch.setenv("CHATTS_GEMM_SK", str(sk))
    wsb = max(int(lib.chatts_linear_workspace(m, n, k)), 8 * m * n * 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    runs = []
    for groups in ("1", "3"):
        monkeypatch.setenv("CHATTS_EPI_NORM_Q_GROUPS", groups)
        out = torch.full((m, n), float("nan"), device=DEV)
        r = resid.clone() if epi == _lib.EPI_RESID else None
        phi = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        plo = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=bias.data_ptr(), resid=_lib.ptr(r), c=out.data_ptr(), norm_w=None,
                             norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=n, epilogue=epi, workspace=ws.data_ptr(),
                             workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
        la.post_norm_w, la.post_norm_eps, la.post_hi, la.post_lo, la.ld_post = nw.data_ptr(), 1e-6, phi.data_ptr(), plo.data_ptr(), n + 8
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        runs.append((out, phi, plo))
    (o0, h0, l0), (o1, h1, l1) = runs
    assert not torch.isnan(o1).any() and torch.equal(o0, o1)
    assert torch.equal(h0[:, :n], h1[:, :n]) and torch.equal(l0[:, :n], l1[:, :n]) and torch.isnan(h1[:, n:].float()).all()
    want = _ref_linear(a, w, bias, resid, epi)
    assert rel_err(o1.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("n,k", [(1024, 512), (5120, 5120), (96, 2048), (5120, 13824), (2048, 16 * 17)])
def test_gemv_fp8_weights_parity(lib, epi, norm, n, k):
    """decode GEMV streaming the fp8 (e4m3fn + per-row power-of-two scale) copy == float64 math on the dequantised
    weights; and == the bf16 GEMV on the dequantised bf16 copy up to summation order."""
    from chatts_amd.modeling import quantize_fp8_rows
    a, w, bias, resid, nw = _rand_problem(1, n, k if k % 32 == 0 else k + (32 - k % 32), seed=n + k + epi)
    k = a.shape[1]
    w[7] *= 37.0                                    # rows with very different scales
    w[11] = 0
    q, scale, deq = quantize_fp8_rows(w)
    assert torch.equal(deq.float(), q.view(torch.float8_e4m3fn).float() * scale[:, None])
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    outs = []
    for use8 in (True, False):
        out = torch.full((1, ncols), float("nan"), device=DEV)
        la = _lib.LinearArgs(a=a.data_ptr(), w=deq.data_ptr(), bias=bias.data_ptr(),
                             resid=resid.data_ptr() if epi == _lib.EPI_RESID else None, c=out.data_ptr(),
                             norm_w=nw.data_ptr() if norm else None, norm_eps=1e-6, m=1, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                             epilogue=epi, workspace=None, workspace_bytes=0,
                             w8=q.data_ptr() if use8 else None, w8_scale=scale.data_ptr() if use8 else None, ldw8=k)
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        outs.append(out)
    want = _ref_linear(a, deq, bias, resid, epi, nw if norm else None)
    assert rel_err(outs[0].cpu().numpy(), want) < 2e-5
    assert rel_err(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("n,k", [(1008, 5120), (27648, 5120), (624, 13824)])
def test_gemv_8bit_geometries_are_bitwise_the_same_sum(lib, monkeypatch, epi, n, k):
    """GEMV8_ROWS x GEMV8_UNR (rows x 1024-element chunks in flight per lane; per-shape defaults: lm_head 4 x 2, gate_up 4 x 1, else 2 x 2,
    DESIGN.md 14.5): a row is summed by one wave in the same lane and chunk order whatever the geometry - every form gives the same bits."""
    from chatts_amd.modeling import quantize_fp8_rows
    if epi == _lib.EPI_SWIGLU:
        n = n // 32 * 32
    a, w, bias, resid, nw = _rand_problem(1, n, k, seed=n + k + epi + 3)
    q, scale, deq = quantize_fp8_rows(w)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    outs = {}
    for rows, unr in ((0, 0), (2, 2), (4, 2), (4, 1), (2, 4)):
        if rows:
            monkeypatch.setenv("CHATTS_GEMV8_ROWS", str(rows))
            monkeypatch.setenv("CHATTS_GEMV8_UNR", str(unr))
        out = torch.full((1, ncols), float("nan"), device=DEV)
        la = _lib.LinearArgs(a=a.data_ptr(), w=deq.data_ptr(), bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), norm_w=nw.data_ptr(), norm_eps=1e-6, m=1, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                             workspace=None, workspace_bytes=0, w8=q.data_ptr(), w8_scale=scale.data_ptr(), ldw8=k)
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        outs[(rows, unr)] = out
    ref = outs[(2, 2)]
    assert not torch.isnan(ref).any() and rel_err(ref.cpu().numpy(), _ref_linear(a, deq, bias, resid, epi, nw)) < 2e-5
    for key, out in outs.items():
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32)), key


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(2, 256, 64), (16, 5120, 5120), (33, 1024, 512), (200, 384, 1024)])
def test_gemm_fp8_weights_parity(lib, epi, m, n, k):
    """MFMA GEMM streaming the fp8 copy of W (widened to bf16 in the LDS staging, row scale in the epilogue)"""
    from chatts_amd.modeling import quantize_fp8_rows
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    w[5] *= 50.0
    q, scale, deq = quantize_fp8_rows(w)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((m, ncols), float("nan"), device=DEV)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=deq.data_ptr(), bias=bias.data_ptr(),
                         resid=resid.data_ptr() if epi == _lib.EPI_RESID else None, c=out.data_ptr(), norm_w=None,
                         norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi, workspace=ws.data_ptr(),
                         workspace_bytes=wsb, w8=q.data_ptr(), w8_scale=scale.data_ptr(), ldw8=k)
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    want = _ref_linear(a, deq, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(2, 128, 128), (5, 160, 384), (16, 5120, 5120), (16, 1024, 13824), (9, 7168, 5120)])
@pytest.mark.parametrize("stages,waves", [(3, 4), (4, 4), (3, 8), (4, 8)])
def test_gemm_stream_fp8_weights_parity(lib, epi, m, n, k, stages, waves, monkeypatch):
    """Batched decode on the fp8 copy of W (BASELINE config 5): the streaming kernel's 128-deep stages, 8-byte fragments
    widened in registers, row scale in the epilogue - against float64 on the dequantised matrix.  (4 or 8 waves per workgroup.)"""
    from chatts_amd.modeling import quantize_fp8_rows
    monkeypatch.setenv("CHATTS_GEMM_STREAM_STAGES", str(stages))
    monkeypatch.setenv("CHATTS_GEMM_STREAM_WAVES", str(waves))
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    w[5] *= 50.0
    q, scale, deq = quantize_fp8_rows(w)
    hi, lo = _split_planes(lib, a)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((m, ncols), float("nan"), device=DEV)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=None, w=deq.data_ptr(), bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                         c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                         workspace=ws.data_ptr(), workspace_bytes=wsb, w8=q.data_ptr(), w8_scale=scale.data_ptr(), ldw8=k,
                         a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    want = _ref_linear(a, deq, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k,norm", [(1, 5120, 5120, True), (1, 1024, 13824, False), (1, 96, 1024, True), (16, 5120, 5120, False),
                                        (9, 7168, 5120, False), (2, 128, 128, False), (40, 256, 512, False)])
def test_linear_int8_weight_copy_parity(lib, epi, m, n, k, norm):
    """ChattsLinearArgs.w8_format = int8 (per-row power-of-two scale, |q| <= 127: a lossless encoding of the bf16 matrix): the decode
    GEMV (M = 1, with / without the fused RMSNorm) and the weight-streaming GEMM (2 <= M <= 16, on planes) stream the int8 copy -
    against float64 math on the dequantised matrix; other shapes (M = 40 here) ignore the copy and stream `w` (same result)."""
    from chatts_amd.modeling import quantize_int8_rows
    a, w, bias, resid, nw = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    w[5] *= 50.0
    q, scale, deq = quantize_int8_rows(w)
    assert torch.equal((q.view(torch.int8).float() * scale[:, None]).to(torch.bfloat16).float(), deq.float())      # exactly bf16
    assert int(q.view(torch.int8).abs().max()) <= 127 and float((deq.float() - w.float()).abs().max()) <= float(scale.max())
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((m, ncols), float("nan"), device=DEV)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=deq.data_ptr(), bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                         c=out.data_ptr(), norm_w=nw.data_ptr() if (norm and m == 1) else None, norm_eps=1e-6, m=m, n=n, k=k, lda=k, ldw=k,
                         ldc=ncols, epilogue=epi, workspace=ws.data_ptr(), workspace_bytes=wsb, w8=q.data_ptr(), w8_scale=scale.data_ptr(),
                         ldw8=k, w8_format=_lib.W8_INT8)
    if 2 <= m <= 16:
        hi, lo = _split_planes(lib, a)
        la.a, la.a_hi, la.a_lo, la.ld_planes = None, hi.data_ptr(), lo.data_ptr(), k
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    want = _ref_linear(a, deq, bias, resid, epi, nw if (norm and m == 1) else None)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k,sk", [(16, 27648, 5120, 1), (16, 5120, 13824, 6), (11, 7168, 5120, 4), (3, 160, 384, 1)])
def test_gemm_stream_fp8_eight_waves_equal_four_bitwise(lib, epi, m, n, k, sk, monkeypatch):
    """gemm_stream_kernel<., true, 1, 8> (16 columns per wave, the SwiGLU pair exchanged through LDS) == the 4-wave form, bit for
    bit: every column sees the same operands in the same K order."""
    from chatts_amd.modeling import quantize_fp8_rows
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    q, scale, deq = quantize_fp8_rows(w)
    hi, lo = _split_planes(lib, a)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    wsb = max(int(lib.chatts_linear_workspace(m, n, k)), 8 * m * n * 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    monkeypatch.setenv("CHATTS_GEMM_SK", str(sk))
    outs = []
    for waves in (4, 8):
        monkeypatch.setenv("CHATTS_GEMM_STREAM_WAVES", str(waves))
        out = torch.full((m, ncols), float("nan"), device=DEV)
        la = _lib.LinearArgs(a=None, w=deq.data_ptr(), bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                             c=out.data_ptr(), norm_w=None, norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=ncols, epilogue=epi,
                             workspace=ws.data_ptr(), workspace_bytes=wsb, w8=q.data_ptr(), w8_scale=scale.data_ptr(), ldw8=k,
                             a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        assert not torch.isnan(out).any()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("m,n,k", [(16, 5120, 5120), (300, 5120, 1536), (16, 5120, 64), (7, 256, 512)])
@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID])
def test_linear_post_norm_planes_equal_separate_rmsnorm(lib, m, n, k, epi):
    """ChattsLinearArgs.post_norm_*: the projection also writes RMSNorm(c) as planes (fused into the split-K epilogue when
    there is one) - bit-identical to chatts_linear followed by chatts_rmsnorm_planes, c itself unchanged."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=2.0)
    g = torch.Generator().manual_seed(3)
    nw = (1 + 0.1 * torch.randn(n, generator=g)).to(DEV)
    hi, lo = _split_planes(lib, a)
    wsb = int(lib.chatts_linear_workspace(m, n, k))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)

    def run(fused):
        out = torch.full((m, n), float("nan"), device=DEV)
        r = resid.clone() if epi == _lib.EPI_RESID else None
        phi = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        plo = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        la = _lib.LinearArgs(a=a.data_ptr(), w=w.data_ptr(), bias=bias.data_ptr(), resid=_lib.ptr(r), c=out.data_ptr(), norm_w=None,
                             norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=n, epilogue=epi, workspace=ws.data_ptr(),
                             workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
        if fused:
            la.post_norm_w, la.post_norm_eps, la.post_hi, la.post_lo, la.ld_post = nw.data_ptr(), 1e-6, phi.data_ptr(), plo.data_ptr(), n + 8
        _lib.check(lib.chatts_linear(la, st()))
        if not fused:
            _lib.check(lib.chatts_rmsnorm_planes(out.data_ptr(), nw.data_ptr(), phi.data_ptr(), plo.data_ptr(), n + 8, m, n, 1e-6, st()))
        torch.cuda.synchronize()
        return out, phi, plo

    o1, h1, l1 = run(False)
    o2, h2, l2 = run(True)
    assert torch.equal(o1, o2) and torch.equal(h1[:, :n], h2[:, :n]) and torch.equal(l1[:, :n], l2[:, :n])
    assert not torch.isnan(h2[:, :n].float()).any() and torch.isnan(h2[:, n:].float()).all()


@pytest.mark.parametrize("sk", [2, 3, 4])
@pytest.mark.parametrize("m,n,k,epi", [(798, 5120, 5120, _lib.EPI_RESID), (300, 5120, 1536, _lib.EPI_NONE), (130, 2080, 2048, _lib.EPI_RESID),
                                       (200, 6144, 1024, _lib.EPI_RESID)])
def test_post_norm_epilogue_with_all_round_trips_at_once_is_bitwise_the_walking_one(lib, monkeypatch, sk, m, n, k, epi):
    """splitk_epilogue_norm_reg_kernel (every slab / residual / weight load of a thread issued up front, the row kept in registers) ==
    splitk_epilogue_norm_kernel (EPI_NORM_REG=0: walks the row, re-reads its stores): c and both planes bit for bit, rows that are not a
    multiple of 1024 columns, 2 .. 4 slabs, with and without the residual."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + sk, scale=2.0)
    g = torch.Generator().manual_seed(5)
    nw = (1 + 0.1 * torch.randn(n, generator=g)).to(DEV)
    hi, lo = _split_planes(lib, a)
    monkeypatch.setenv("CHATTS_GEMM_SK", str(sk))
    wsb = max(int(lib.chatts_linear_workspace(m, n, k)), 4 * m * n * 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    runs = []
    for reg in ("0", "1"):
        monkeypatch.setenv("CHATTS_EPI_NORM_REG", reg)
        out = torch.full((m, n), float("nan"), device=DEV)
        r = resid.clone() if epi == _lib.EPI_RESID else None
        phi = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        plo = torch.full((m, n + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
        la = _lib.LinearArgs(a=None, w=w.data_ptr(), bias=bias.data_ptr(), resid=_lib.ptr(r), c=out.data_ptr(), norm_w=None,
                             norm_eps=0.0, m=m, n=n, k=k, lda=k, ldw=k, ldc=n, epilogue=epi, workspace=ws.data_ptr(),
                             workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=k)
        la.post_norm_w, la.post_norm_eps, la.post_hi, la.post_lo, la.ld_post = nw.data_ptr(), 1e-6, phi.data_ptr(), plo.data_ptr(), n + 8
        _lib.check(lib.chatts_linear(la, st()))
        torch.cuda.synchronize()
        runs.append((out, phi, plo))
    (o0, h0, l0), (o1, h1, l1) = runs
    assert not torch.isnan(o1).any() and torch.equal(o0, o1)
    assert torch.equal(h0[:, :n], h1[:, :n]) and torch.equal(l0[:, :n], l1[:, :n]) and torch.isnan(h1[:, n:].float()).all()
    want = _ref_linear(a, w, bias, resid, epi)
    assert rel_err(o1.cpu().numpy(), want) < 2e-5


def test_ts_normalise_on_device_matches_the_reference_vectors(lib, golden):
    """chatts_ts_normalise (GPU-side normalisation statistics) against the vectors the REFERENCE's
    sp_encoding produced (tests/golden/sp_encoding.npz): identical prompt prefixes (the %.4f text), statistics within one
    float64 ulp-scale, (value, mask) rows equal to the reference's float64 values rounded to float32 save for rounding ties."""
    from chatts_amd import config as cfgmod
    from chatts_amd.processing import ChatTSProcessor
    g = golden("sp_encoding")
    n = int(g["n"])
    series = [g[f"in_{i}"] for i in range(n)]
    proc = ChatTSProcessor.from_pretrained(cfgmod.preset("tiny-qwen2"), prefix_format="sp")
    enc, prefixes, lens = proc.encode_batch_on_device(series)
    lmax = max(lens)
    assert enc.shape == (n, 2 * lmax, 1) and enc.is_cuda
    e = enc.cpu().numpy().reshape(n, lmax, 2)
    for i in range(n):
        want = g[f"enc_{i}"].reshape(-1, 2)                     # float64 [L, 2] from the reference
        L = want.shape[0]
        assert prefixes[i] == str(g[f"prompt_{i}"])             # "[Value Offset: ...|Value Scaling: ...]<ts><ts/>"
        assert np.all(e[i, :L, 1] == 1.0) and np.all(e[i, L:] == 0.0)
        w32 = want[:, 0].astype(np.float32)
        ulp = np.abs(np.spacing(w32))
        assert np.all(np.abs(e[i, :L, 0] - w32) <= ulp)
        assert np.mean(e[i, :L, 0] == w32) > 0.99
    # the HF-style prefix of the stored notebook output (demo/demo_lora.ipynb:147)
    x = np.arange(256)
    ts1 = np.sin(x / 10) * 5.0
    ts1[100:] -= 10.0
    hf = ChatTSProcessor.from_pretrained(cfgmod.preset("tiny-qwen2"))
    _, pf, _ = hf.encode_batch_on_device([ts1, []])
    assert pf[0].startswith("[offset=6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|left=0.0000|right=-8.2047]")
    assert pf[1] == "<ts><ts/>"
    # whole call surface: same ids and the same tensor (to float32 rounding ties) as the host path
    a = hf(text=["A <ts><ts/> B <ts><ts/>"], timeseries=[ts1, x * 0.05], return_tensors="pt")
    b = hf(text=["A <ts><ts/> B <ts><ts/>"], timeseries=[ts1, x * 0.05], return_tensors="pt", device_stats=True)
    assert torch.equal(a["input_ids"], b["input_ids"]) and b["timeseries"].is_cuda
    assert (a["timeseries"] - b["timeseries"].cpu()).abs().max() <= 1e-6


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("n,k,gs", [(1024, 512, 128), (5120, 5120, 128), (96, 1024, 64), (5120, 13824, 128), (2048, 256, 32), (160, 3072, 16)])
def test_gemv_int4_parity(lib, epi, norm, n, k, gs):
    """4-bit decode GEMV: streams the codes and rebuilds bf16_rne(scale * (code - zero)); reference = float64 on that bf16 matrix."""
    from chatts_amd.modeling import pack_int4, quantize_int4_rows
    a, w, bias, resid, nw = _rand_problem(1, n, k, seed=n + k + epi + gs)
    q, sc, z, deq = quantize_int4_rows(w, gs)
    w4, sz = pack_int4(q, sc, z)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    out = torch.full((1, ncols), float("nan"), dtype=torch.float32, device=DEV)
    la = _lib.LinearArgs(a=a.data_ptr(), w=deq.data_ptr(), bias=bias.data_ptr(), resid=resid.data_ptr() if epi == _lib.EPI_RESID else None,
                         c=out.data_ptr(), norm_w=nw.data_ptr() if norm else None, norm_eps=1e-6, m=1, n=n, k=k, lda=k, ldw=k, ldc=ncols,
                         epilogue=epi, workspace=None, workspace_bytes=0, w4=w4.data_ptr(), w4_sz=sz.data_ptr(), ldw4=k // 2, w4_group=gs)
    _lib.check(lib.chatts_linear(la, st()))
    torch.cuda.synchronize()
    want = _ref_linear(a, deq, bias, resid, epi, nw if norm else None)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5
    # and the bf16 GEMV on the dequantised matrix agrees to summation order
    out2 = _linear(lib, a, deq, bias, resid if epi == _lib.EPI_RESID else None, epi, nw if norm else None)
    assert rel_err(out.cpu().numpy(), out2.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID])
@pytest.mark.parametrize("m,n,k", [(17, 5120, 5120), (32, 5120, 320), (33, 1024, 512), (64, 5120, 5120), (65, 7168, 5120), (100, 384, 1024),
                                   (128, 5120, 5120), (128, 5120, 13824), (128, 4096, 4096), (23, 96, 64)])
def test_gemm_stream_multiblock_parity(lib, monkeypatch, epi, m, n, k):
    """gemm_stream_kernel with 2 / 4 / 8 row blocks (17 <= M <= 128, few N-panels: the TS-encoder MLP shapes) vs float64 and
    vs the LDS-DMA kernel it replaces there (CHATTS_GEMM_STREAM_MB=0)."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    r = resid if epi == _lib.EPI_RESID else None
    monkeypatch.setenv("CHATTS_GEMM_STREAM_MB", "128")          # (the shipped default stops at 64 rows: measured crossover)
    out = _linear_planes(lib, a, w, bias, r, epi, with_a=False, ld=k + 64)
    monkeypatch.setenv("CHATTS_GEMM_STREAM_MB", "0")
    base = _linear_planes(lib, a, w, bias, r, epi, with_a=False, ld=k + 64)
    want = _ref_linear(a, w, bias, resid, epi)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), want) < 2e-5
    assert rel_err(out.cpu().numpy(), base.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID])
@pytest.mark.parametrize("m,n,k", [(17, 5120, 5120), (33, 1024, 512), (64, 5120, 5120), (100, 384, 1024), (128, 5120, 5120), (23, 96, 64)])
def test_gemm_stream_multiblock_eight_waves_equal_four_bitwise(lib, monkeypatch, epi, m, n, k):
    """The multi-block streaming kernel with 8 waves as 2 x 4 (each wave half the row blocks x 32 columns: the K-step's DMA pieces are
    spread over twice the waves) == the 4-wave form, bit for bit - every output element sees the same operands in the same K order."""
    a, w, bias, resid, _ = _rand_problem(m, n, k, seed=m + n + k + epi, scale=3.0)
    r = resid if epi == _lib.EPI_RESID else None
    monkeypatch.setenv("CHATTS_GEMM_STREAM_MB", "128")
    outs = []
    for waves in ("4", "8"):
        monkeypatch.setenv("CHATTS_GEMM_STREAM_MB_WAVES", waves)
        outs.append(_linear_planes(lib, a, w, bias, r, epi, with_a=False, ld=k + 64))
    assert not torch.isnan(outs[1]).any()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("nq,nkv", [(4, 2), (5, 1), (8, 8)])
@pytest.mark.parametrize("T,pos0,splits", [(16, 0, 1), (64, 0, 1), (65, 31, 1), (130, 100, 1), (200, 0, 1), (300, 17, 2), (70, 0, 2), (33, 400, 4)])
def test_attention_prefill_bf16x3_parity(lib, monkeypatch, nq, nkv, T, pos0, splits):
    """attn_prefill_bf16x3_kernel (CHATTS_ATTN_BF16X3=1): Q.K^T and P.V on the bf16 matrix pipe, both operands split into hi / lo
    planes, three passes per product - against the float64 reference and against the float32-MFMA kernel."""
    max_ctx = 512
    g = torch.Generator().manual_seed(nq * 100 + T + pos0)
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), generator=g)
    kc = torch.randn((nkv, max_ctx, 128), generator=g)
    vc = torch.randn((nkv, max_ctx, 128), generator=g)
    kc[:, 17] *= 4.0     # a spiky key so the online-softmax rescale path is exercised
    want = _ref_attention(qkv.view(T, nq + 2 * nkv, 128)[:, :nq], kc, vc, pos0)
    qd, kd, vd = qkv.to(DEV), kc.to(DEV), vc.to(DEV)
    cache = _lib.KvCache(k=kd.data_ptr(), v=vd.data_ptr(), max_ctx=max_ctx)
    wsb = int(lib.chatts_attn_workspace(T, nq, splits))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CHATTS_ATTN_BF16X3", mode)
        out = torch.full((T, nq * 128), float("nan"), device=DEV)
        _lib.check(lib.chatts_attention(qd.data_ptr(), T, nq, nkv, pos0, None, C.byref(cache), out.data_ptr(), splits, ws.data_ptr(), wsb, st()))
        torch.cuda.synchronize()
        outs[mode] = out.cpu().numpy().reshape(T, nq, 128)
    assert not np.isnan(outs["1"]).any()
    assert rel_err(outs["1"], want.numpy()) < 3e-5
    assert rel_err(outs["1"], outs["0"]) < 3e-5


# ---- block-paged KV cache (ChattsKvCache.block_table): same arithmetic, blocks scattered through a pool -------------------------
def _to_pool(kc, vc, block, seed, spare=3):
    """contiguous [nkv, max_ctx, 128] caches -> (pool_k, pool_v [n_blocks, nkv, block, 128], table [max_ctx / block] int32) with the
    logical blocks at shuffled pool positions; the spare blocks hold NaN (nothing may ever read them)"""
    nkv, max_ctx, _ = kc.shape
    nb = max_ctx // block
    ids = torch.randperm(nb + spare, generator=torch.Generator().manual_seed(seed))[:nb]
    pk = torch.full((nb + spare, nkv, block, 128), float("nan"), device=kc.device)
    pv = torch.full((nb + spare, nkv, block, 128), float("nan"), device=kc.device)
    for i, b in enumerate(ids.tolist()):
        pk[b] = kc[:, i * block:(i + 1) * block]
        pv[b] = vc[:, i * block:(i + 1) * block]
    return pk, pv, ids.to(torch.int32).to(kc.device)


def _from_pool(pool, table, block):
    return torch.cat([pool[int(b)] for b in table.tolist()], dim=1)        # -> [nkv, max_ctx, 128]


@pytest.mark.parametrize("block", [64, 256])
@pytest.mark.parametrize("bf16x3", ["0", "1"])
@pytest.mark.parametrize("T,pos0,splits", [(1, 0, 1), (1, 333, 16), (7, 60, 1), (70, 0, 1), (65, 31, 1), (130, 100, 2), (200, 300, 1)])
def test_attention_paged_equals_contiguous(lib, monkeypatch, block, bf16x3, T, pos0, splits):
    """chatts_attention (VALU rows kernel, float32-MFMA and bf16x3 prefill kernels) through a shuffled block table == the
    contiguous cache, bit for bit"""
    monkeypatch.setenv("CHATTS_ATTN_BF16X3", bf16x3)
    nq, nkv, max_ctx = 10, 2, 512
    g = torch.Generator().manual_seed(T * 31 + pos0)
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    kc = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    vc = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    kc[:, pos0 + T:] = float("nan")                    # rows past the context must not be touched in either form
    vc[:, pos0 + T:] = float("nan")
    wsb = int(lib.chatts_attn_workspace(T, nq, splits))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    out_a = torch.full((T, nq * 128), float("nan"), device=DEV)
    ca = _lib.KvCache(k=kc.data_ptr(), v=vc.data_ptr(), max_ctx=max_ctx)
    _lib.check(lib.chatts_attention(qkv.data_ptr(), T, nq, nkv, pos0, None, C.byref(ca), out_a.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    pk, pv, table = _to_pool(kc, vc, block, seed=T + block)
    out_b = torch.full((T, nq * 128), float("nan"), device=DEV)
    cb = _lib.KvCache(k=pk.data_ptr(), v=pv.data_ptr(), max_ctx=max_ctx, block_table=table.data_ptr(), block_size=block,
                      table_stride=table.numel())
    _lib.check(lib.chatts_attention(qkv.data_ptr(), T, nq, nkv, pos0, None, C.byref(cb), out_b.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    assert not torch.isnan(out_a).any() and torch.equal(out_a, out_b)


@pytest.mark.parametrize("paged,nq,nkv,T,pos0", [
    (False, 10, 2, 64, 0), (False, 10, 2, 65, 31), (False, 10, 2, 130, 100), (False, 10, 2, 798, 0), (False, 5, 1, 200, 300),
    (False, 5, 1, 1000, 24), (True, 10, 2, 65, 31), (True, 10, 2, 200, 300), (True, 5, 1, 798, 0), (True, 8, 8, 130, 100)])
def test_attention_prefill_on_kv_planes(lib, paged, nq, nkv, T, pos0):
    """attn_prefill_planes_kernel (K / V rows split into bf16 hi / lo planes ONCE by kv_planes_kernel, transposed tiles S^T = K.Q^T,
    O^T = V^T.P^T; taken when the caller's workspace is free and large enough) against the float64 reference and against
    attn_prefill_bf16x3_kernel, which splits every tile while it stages it (workspace too small): the same three-pass products summed
    in another order.  Rows past the context are never read in either form (NaN there), contiguous and block-paged caches."""
    max_ctx = 1024
    g = torch.Generator().manual_seed(T * 7 + pos0 + nq)
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), generator=g)
    kc = torch.randn((nkv, max_ctx, 128), generator=g)
    vc = torch.randn((nkv, max_ctx, 128), generator=g)
    kc[:, 17] *= 4.0     # a spiky key so the online-softmax rescale path is exercised
    want = _ref_attention(qkv.view(T, nq + 2 * nkv, 128)[:, :nq], kc, vc, pos0).numpy()
    qd, kd, vd = qkv.to(DEV), kc.to(DEV), vc.to(DEV)
    kd[:, pos0 + T:] = float("nan")
    vd[:, pos0 + T:] = float("nan")
    if paged:
        pk, pv, table = _to_pool(kd, vd, 64, seed=T)
        cache = _lib.KvCache(k=pk.data_ptr(), v=pv.data_ptr(), max_ctx=max_ctx, block_table=table.data_ptr(), block_size=64,
                             table_stride=table.numel())
    else:
        cache = _lib.KvCache(k=kd.data_ptr(), v=vd.data_ptr(), max_ctx=max_ctx)
    plane_bytes = ((pos0 + T + 31) // 32) * nkv * 4 * 32 * 128 * 2
    outs = []
    for wsb in (plane_bytes - 16, plane_bytes):              # too small (split while staging), then exactly enough (planes)
        ws = torch.full((wsb,), 0xFF, dtype=torch.uint8, device=DEV)
        out = torch.full((T, nq * 128), float("nan"), device=DEV)
        _lib.check(lib.chatts_attention(qd.data_ptr(), T, nq, nkv, pos0, None, C.byref(cache), out.data_ptr(), 1, ws.data_ptr(), wsb, st()))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy().reshape(T, nq, 128))
        assert bool((ws != 0xFF).any()) == (wsb == plane_bytes)      # the planes really were (not) written
    assert not np.isnan(outs[1]).any()
    assert rel_err(outs[1], want) < 3e-5 and rel_err(outs[0], want) < 3e-5
    assert rel_err(outs[1], outs[0]) < 1e-5


def test_paged_cache_argument_errors(lib):
    kc = torch.zeros((2, 192, 128), device=DEV)
    table = torch.zeros(3, dtype=torch.int32, device=DEV)
    q = torch.zeros((1, 6 * 128), device=DEV)
    out = torch.zeros((1, 2 * 128), device=DEV)
    ws = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    for bs, ctx in ((48, 192), (32, 192), (64, 200)):       # not a power of two / below 64 / max_ctx not whole blocks
        c = _lib.KvCache(k=kc.data_ptr(), v=kc.data_ptr(), max_ctx=ctx, block_table=table.data_ptr(), block_size=bs, table_stride=3)
        assert lib.chatts_attention(q.data_ptr(), 1, 2, 2, 0, None, C.byref(c), out.data_ptr(), 1, ws.data_ptr(), 4096, st()) == _lib.E_BADARG
        cos, sin = _rope_tables(8)
        assert lib.chatts_rope_kv_write(q.data_ptr(), 1, 2, 2, None, None, 1e-6, cos.data_ptr(), sin.data_ptr(), 0, None, C.byref(c),
                                        st()) == _lib.E_BADARG


@pytest.mark.parametrize("qk_norm", [False, True])
def test_rope_kv_write_paged(lib, qk_norm):
    T, nq, nkv, pos0, max_ctx, block = 150, 4, 2, 37, 256, 64         # the rows straddle three blocks
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn((T, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    qn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    kn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV) if qk_norm else None
    cos, sin = _rope_tables(max_ctx)
    kc = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    vc = torch.randn((nkv, max_ctx, 128), generator=g).to(DEV)
    pk, pv, table = _to_pool(kc, vc, block, seed=11)
    qa, qb = qkv.clone(), qkv.clone()
    ca = _lib.KvCache(k=kc.data_ptr(), v=vc.data_ptr(), max_ctx=max_ctx)
    cb = _lib.KvCache(k=pk.data_ptr(), v=pv.data_ptr(), max_ctx=max_ctx, block_table=table.data_ptr(), block_size=block,
                      table_stride=table.numel())
    for q, c in ((qa, ca), (qb, cb)):
        _lib.check(lib.chatts_rope_kv_write(q.data_ptr(), T, nq, nkv, _lib.ptr(qn), _lib.ptr(kn), 1e-6, cos.data_ptr(), sin.data_ptr(),
                                            pos0, None, C.byref(c), st()))
    torch.cuda.synchronize()
    assert torch.equal(qa, qb)
    assert torch.equal(_from_pool(pk, table, block), kc) and torch.equal(_from_pool(pv, table, block), vc)
    spare = [b for b in range(pk.shape[0]) if b not in table.tolist()]
    assert torch.isnan(pk[spare]).all()                 # blocks outside the table were not written


@pytest.mark.parametrize("block", [64, 128])
def test_attention_decode_batched_paged_equals_contiguous(lib, block):
    """batched fused decode attention: every sequence reads and writes through its own row of the block table"""
    nq, nkv, max_ctx, B, splits = 10, 2, 256, 5, 16
    g = torch.Generator().manual_seed(77)
    raw = torch.randn((B, (nq + 2 * nkv) * 128), generator=g).to(DEV)
    kc0 = torch.randn((B, nkv, max_ctx, 128), generator=g).to(DEV)
    vc0 = torch.randn((B, nkv, max_ctx, 128), generator=g).to(DEV)
    pos = [0, 63, 64, 200, -1]                          # block edges, and a parked slot
    qn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV)
    kn = (1 + 0.1 * torch.randn(128, generator=g)).to(DEV)
    cos, sin = _rope_tables(max_ctx)
    wsb = int(lib.chatts_attn_workspace(B, nq, splits))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    pos_dev = torch.tensor(pos, dtype=torch.int32, device=DEV)
    ka, va = kc0.clone(), vc0.clone()
    ca = _lib.KvCache(k=ka.data_ptr(), v=va.data_ptr(), max_ctx=max_ctx)
    out_a = torch.zeros((B, nq * 128), device=DEV)
    _lib.check(lib.chatts_attention_decode_batched(raw.data_ptr(), B, nq, nkv, qn.data_ptr(), kn.data_ptr(), 1e-6, cos.data_ptr(),
                                                   sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(ca), nkv * max_ctx * 128,
                                                   out_a.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    # one pool for all sequences: sequence b's logical blocks are interleaved with the others'
    nb = max_ctx // block
    order = torch.randperm(B * nb, generator=g)
    table = order.view(B, nb).to(torch.int32).to(DEV)
    pk = torch.empty((B * nb, nkv, block, 128), device=DEV)
    pv = torch.empty((B * nb, nkv, block, 128), device=DEV)
    for b in range(B):
        for i in range(nb):
            pk[int(table[b, i])] = kc0[b, :, i * block:(i + 1) * block]
            pv[int(table[b, i])] = vc0[b, :, i * block:(i + 1) * block]
    cb = _lib.KvCache(k=pk.data_ptr(), v=pv.data_ptr(), max_ctx=max_ctx, block_table=table.data_ptr(), block_size=block, table_stride=nb)
    out_b = torch.zeros((B, nq * 128), device=DEV)
    _lib.check(lib.chatts_attention_decode_batched(raw.data_ptr(), B, nq, nkv, qn.data_ptr(), kn.data_ptr(), 1e-6, cos.data_ptr(),
                                                   sin.data_ptr(), 0, pos_dev.data_ptr(), C.byref(cb), 0,
                                                   out_b.data_ptr(), splits, ws.data_ptr(), wsb, st()))
    torch.cuda.synchronize()
    live = [b for b in range(B) if pos[b] >= 0]
    assert torch.equal(out_a[live], out_b[live])
    for b in range(B):
        assert torch.equal(_from_pool(pk, table[b], block), ka[b]) and torch.equal(_from_pool(pv, table[b], block), va[b])


@pytest.mark.parametrize("ks", [0, 1, 2, 3, 5, 8, 16])
def test_gemv_ksplit_at_tensor_parallel_shard_shapes(lib, ks, monkeypatch):
    """gemv_ksplit_kernel (several waves share a row group, partial sums meet in LDS): the form a rank of TP = 4 / 8 runs for qkv
    (896 / 1792 x 5120) - picked automatically when the row groups alone leave the chip nearly empty (ks = 0 here: the launcher's
    own choice), forced to 1 .. 16 waves per group otherwise.  Same float64 bar as every GEMV; two calls give the same bits."""
    if ks:
        monkeypatch.setenv("CHATTS_GEMV_KS", str(ks))
    for epi, n, k, norm in [(_lib.EPI_NONE, 896, 5120, True), (_lib.EPI_SWIGLU, 6912, 5120, True), (_lib.EPI_RESID, 5120, 1728, False),
                            (_lib.EPI_NONE, 1792, 5120, True), (_lib.EPI_NONE, 96, 13824, False), (_lib.EPI_NONE, 19008, 5120, True)]:
        a, w, bias, resid, nw = _rand_problem(1, n, k, seed=n + k + ks)
        nob = epi == _lib.EPI_RESID
        out = _linear(lib, a, w, None if nob else bias, resid if nob else None, epi, nw if norm else None)
        want = _ref_linear(a, w, None if nob else bias, resid, epi, nw if norm else None)
        assert rel_err(out.cpu().numpy(), want) < 2e-5, (epi, n, k)
        again = _linear(lib, a, w, None if nob else bias, resid if nob else None, epi, nw if norm else None)
        assert torch.equal(out, again)


def _fp8_ref_quant(x, norm_w=None, eps=1e-6):
    """chatts_quantize_rows_fp8's arithmetic in torch: (optional RMSNorm,) scale = amax / 448, e4m3fn codes of x * (1 / scale)"""
    xf = x.float()
    if norm_w is not None:
        rstd = torch.rsqrt((xf.double() ** 2).mean(dim=1, keepdim=True) + eps).float()
        xf = norm_w[None] * (xf * rstd)
    amax = xf.abs().amax(dim=1)
    s = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    q = (xf * (1.0 / s)[:, None]).to(torch.float8_e4m3fn)
    return q, s


@pytest.mark.parametrize("epi", [_lib.EPI_NONE, _lib.EPI_GELU, _lib.EPI_RESID, _lib.EPI_SWIGLU])
@pytest.mark.parametrize("m,n,k", [(200, 256, 384), (130, 5120, 5120), (17, 640, 128), (1207, 896, 640), (64, 1024, 13824)])
def test_linear_fp8_speed_mode_is_the_exact_product_of_the_quantised_operands(lib, epi, m, n, k):
    """gemm_fp8.hip (v_mfma_scale_f32_16x16x128_f8f6f4, SPEED mode): the kernel's only approximation is the quantisation itself - its
    output equals, to float32 summation order, the float64 product of the e4m3 codes x scales it was given, for every epilogue, ragged
    M / N tiles, K from one to 108 MFMA steps.  (How far fp8 activations are from the float32 path is measured end to end:
    profiles/r4_fp8_speed_mode.json - this test pins the kernel.)"""
    import ctypes as C
    g = torch.Generator().manual_seed(m + n + k + epi)
    x = (torch.randn((m, k), generator=g) * 0.7).to(DEV)
    x[:, 0] = torch.linspace(-3, 3, m).to(DEV)                          # asymmetric structure (guide rule 16)
    w = (torch.randn((n, k), generator=g) * 0.05).to(DEV)
    w[:, 0] = torch.linspace(-1, 1, n).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    ncols = n // 2 if epi == _lib.EPI_SWIGLU else n
    resid = torch.randn((m, ncols), generator=g).to(DEV)
    q8 = torch.empty((m, k), dtype=torch.uint8, device=DEV)
    sa = torch.empty(m, dtype=torch.float32, device=DEV)
    _lib.check(lib.chatts_quantize_rows_fp8(x.data_ptr(), m, k, k, None, 0.0, q8.data_ptr(), k, sa.data_ptr(), _lib.stream_ptr()))
    q_ref, s_ref = _fp8_ref_quant(x)
    assert torch.allclose(sa, s_ref, rtol=3e-7, atol=0)                            # (the device's amax / 448 may differ in the last bit)
    q_dev = (x * (1.0 / sa)[:, None]).to(torch.float8_e4m3fn)                      # the codes for the device's own scales
    assert (q8 != q_dev.view(torch.uint8)).float().mean().item() < 2e-3           # (ties of x * (1 / s) may round the other way)
    wq, ws = _fp8_ref_quant(w)
    w8 = wq.view(torch.uint8).contiguous()
    out = torch.full((m, ncols), float("nan"), device=DEV)
    fa = _lib.LinearFp8Args(a8=q8.data_ptr(), a_scale=sa.data_ptr(), w8=w8.data_ptr(), w_scale=ws.data_ptr(), bias=bias.data_ptr(),
                            resid=resid.data_ptr() if epi == _lib.EPI_RESID else None, c=out.data_ptr(), m=m, n=n, k=k, lda8=k, ldw8=k,
                            ldc=ncols, epilogue=epi)
    _lib.check(lib.chatts_linear_fp8(C.byref(fa), _lib.stream_ptr()))
    a_deq = q8.view(torch.float8_e4m3fn).double().cpu() * sa.double().cpu()[:, None]
    w_deq = w8.view(torch.float8_e4m3fn).double().cpu() * ws.double().cpu()[:, None]
    y = a_deq @ w_deq.T + bias.double().cpu()
    if epi == _lib.EPI_GELU:
        y = torch.nn.functional.gelu(y)
    elif epi == _lib.EPI_RESID:
        y = y + resid.double().cpu()
    elif epi == _lib.EPI_SWIGLU:
        v = y.view(m, n // 32, 2, 16)
        y = (torch.nn.functional.silu(v[:, :, 0]) * v[:, :, 1]).reshape(m, n // 2)
    assert not torch.isnan(out).any()
    assert rel_err(out.cpu().numpy(), y.numpy()) < 1e-4      # (measured 1e-5 .. 3e-5: the block-scaled MFMA's internal summation + float32 epilogue)


def test_quantize_rows_fp8_fuses_rmsnorm(lib):
    m, k = 37, 5120
    g = torch.Generator().manual_seed(3)
    x = torch.randn((m, k), generator=g).to(DEV)
    x[5] = 0.0                                                              # an all-zero row: scale 1, codes 0
    nw = (1 + 0.1 * torch.randn(k, generator=g)).to(DEV)
    q8 = torch.empty((m, k), dtype=torch.uint8, device=DEV)
    sa = torch.empty(m, dtype=torch.float32, device=DEV)
    _lib.check(lib.chatts_quantize_rows_fp8(x.data_ptr(), m, k, k, nw.data_ptr(), 1e-6, q8.data_ptr(), k, sa.data_ptr(), _lib.stream_ptr()))
    deq = q8.view(torch.float8_e4m3fn).double().cpu() * sa.double().cpu()[:, None]
    xd = x.double().cpu()
    want = nw.double().cpu()[None] * (xd * torch.rsqrt((xd ** 2).mean(dim=1, keepdim=True) + 1e-6))
    assert sa[5].item() == 1.0 and torch.count_nonzero(q8[5] & 0x7f) == 0
    assert rel_err(deq.numpy(), want.numpy()) < 0.04                       # e4m3: 3 mantissa bits, round to nearest
    assert (deq.abs().amax(dim=1)[:5] - want.abs().amax(dim=1)[:5]).abs().max() < 1e-4      # the row maximum maps to 448 exactly
