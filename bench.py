#!/usr/bin/env python
"""bench.py - generated tokens/s + p50 TTFT, ChatTS-14B, 8 series x 256 steps, tensor parallel over N GPUs.

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE greedy decode step of the whole model (all 48 layers + lm_head + argmax, batch 1) after the
8x256 prompt has been encoded and prefilled; `value` = steps / max-over-ranks wall time.  Inputs (weights,
prompt, series tensor) are resident in HBM when the timed region starts.  Extra keys: ttft_ms_p50 (processor +
H2D + TS encoder + merge + prefill + first token), `roofline` (dominant kernel = the gate_up weight-streaming
GEMV, timed live with HIP events), `cpu_baseline` (the CPU float32 oracle on this box's host cores, bounded sample).
Weights are synthetic (counter-hash, bf16-exact) because no checkpoint can reach the box; shapes are the real
ChatTS-14B ones.  Multi-GPU = tensor parallel (strong scaling), RCCL all-reduce via torch.distributed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def workload_lengths(rng, n_series, length, mode="uniform"):
    """Series lengths of a workload.  "uniform": n_series x length.  "mixed" (BASELINE.json config 4):
    rng.integers(64, 1025, n_series) drawn from the workload's generator BEFORE the values, with two entries forced to ragged
    tails (L % 16 != 0: the last-value pad / padding_idx branch of chatts_vllm.py:121-129 must be on the path)."""
    if mode != "mixed":
        return [int(length)] * n_series
    lengths = [int(v) for v in rng.integers(64, 1025, n_series)]
    lengths[3 % n_series], lengths[17 % n_series] = 1000, 65
    return lengths


def build_inputs(cfg, n_series=8, length=256, mode="uniform"):
    import numpy as np
    from chatts_amd.processing import ChatTSProcessor
    proc = ChatTSProcessor.from_pretrained(cfg)
    rng = np.random.default_rng(1234)                       # SURVEY.md section 8d synthetic inputs
    lengths = workload_lengths(rng, n_series, length, mode)
    series = [50 + 2 * np.cumsum(rng.standard_normal(L)) for L in lengths]
    body = f"I have {n_series} time series. " + " ".join(
        f"TS{i} is of length {L}: <ts><ts/>;" for i, L in enumerate(lengths)) + \
        " Please analyze the local changes in these time series first and then conclude if these time series " \
        "show local changes near the same time?"
    prompt = ("<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\n" + body +
              "<|im_end|><|im_start|>assistant\n")
    return proc, prompt, series, lengths


def build_batched_requests(cfg, batch, n_series=8, length=1024):
    """BASELINE.json config 5 inputs: `batch` DIFFERENT prompts of n_series x length (one generator, series drawn request by request)."""
    import numpy as np
    from chatts_amd.processing import ChatTSProcessor
    proc = ChatTSProcessor.from_pretrained(cfg)
    rng = np.random.default_rng(1234)
    lengths = [length] * n_series
    body = f"I have {n_series} time series. " + " ".join(f"TS{i} is of length {L}: <ts><ts/>;" for i, L in enumerate(lengths)) + \
        " Please analyze the local changes in these time series first and then conclude if these time series show local changes near the same time?"
    prompt = "<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\n" + body + "<|im_end|><|im_start|>assistant\n"
    reqs = [[50 + 2 * np.cumsum(rng.standard_normal(L)) for L in lengths] for _ in range(batch)]
    return proc, prompt, reqs, lengths


def admit_batched(model, proc, prompt, reqs, budget, pack=True):
    """The engine's admission loop over all cache slots: short prompts are prefilled together (chatts_decoder_prefill_packed).
    -> (prompt tokens, per-request TTFT ms, packed passes)."""
    import torch
    pending = []
    for series in reqs:
        inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
        pending.append((inputs["input_ids"][0].tolist(), inputs["timeseries"], list(proc.last_lengths), budget))
    free = list(range(len(reqs)))
    packs, ttfts, T = 0, [], None
    while pending:
        t0 = time.perf_counter()
        pk = model.plan_pack(pending[:len(free)], free) if pack else []
        group = [pending[j] for j in pk] if pk else pending[:1]
        items = [(free.pop(0),) + g for g in group]
        if len(items) > 1:
            T = model._admit_packed(items)[0]
            packs += 1
        else:
            T = model._admit(*items[0])
        for g in group:
            pending.remove(g)
        torch.cuda.synchronize()
        ttfts += [(time.perf_counter() - t0) * 1e3] * len(items)
    return T, ttfts, packs


def pmc_traffic_entry(entry, files_key="code_files"):
    """A counter-measured traffic figure of profiles/pmc_traffic.json is quoted only for the code it was measured on: the entry carries
    a digest of its kernel sources (tools/pmc_traffic.py); if the checked-out sources differ the caller gets (None, why)."""
    import hashlib
    if not entry or not entry.get("code_digest"):
        return None, "no counter pass recorded for this kernel on the current code (entry has no code_digest)"
    h = hashlib.sha256()
    try:
        for f in entry.get(files_key, []):
            h.update(f.encode())
            with open(os.path.join(ROOT, f), "rb") as fh:
                h.update(fh.read())
    except OSError as e:
        return None, f"cannot hash the kernel sources: {e}"
    if h.hexdigest()[:16] != entry["code_digest"]:
        return None, f"stale: {entry.get('source', 'the recorded pass')} was collected on other kernel sources (digest {entry['code_digest']})"
    return entry, None


def median(xs):
    s = sorted(xs)
    return s[len(s) // 2]


def roofline_gate_up(model, reps=2, m=1):
    """Dominant kernel: gemv_kernel<.., SWIGLU, NORM> on gate_up (2*I_local x H bf16 per launch).  Launch it once
    per layer over every layer's own weights (13.6 GB at TP=1: no cache reuse) between two HIP events recorded on
    the launching stream; achieved = algorithmic bytes per launch / average launch duration."""
    import torch
    from chatts_amd import _lib
    lib, cfg, plan = model.lib, model.config, model.plan
    H = cfg.hidden_size
    n = 2 * plan.inter
    out = torch.empty((m, plan.inter), dtype=torch.float32, device=model.device)
    x = torch.randn((m, H), dtype=torch.float32, device=model.device)
    stream = torch.cuda.current_stream()
    if m > 1:        # batched decode: the weight-streaming MFMA kernel on bf16 hi / lo planes (what the B-wide step launches)
        hi, lo = torch.empty((m, H), dtype=torch.bfloat16, device=model.device), torch.empty((m, H), dtype=torch.bfloat16, device=model.device)
        chi = torch.empty((m, plan.inter), dtype=torch.bfloat16, device=model.device)
        clo = torch.empty((m, plan.inter), dtype=torch.bfloat16, device=model.device)
        _lib.check(lib.chatts_split_bf16x2(x.data_ptr(), m, H, H, hi.data_ptr(), lo.data_ptr(), H, stream.cuda_stream))
        wsb = max(int(lib.chatts_linear_workspace(m, n, H)), 16)
        ws = torch.empty(wsb, dtype=torch.uint8, device=model.device)

    def sweep():
        for lw in model.layers:
            if m == 1:
                la = _lib.LinearArgs(a=x.data_ptr(), w=lw["gate_up"].data_ptr(), bias=None, resid=None, c=out.data_ptr(),
                                     norm_w=lw["post_norm"].data_ptr(), norm_eps=cfg.rms_norm_eps, m=1, n=n, k=H, lda=H,
                                     ldw=H, ldc=plan.inter, epilogue=_lib.EPI_SWIGLU, workspace=None, workspace_bytes=0,
                                     w8=_lib.ptr(lw.get("gate_up8")), w8_scale=_lib.ptr(lw.get("gate_up8_scale")), ldw8=H,
                                     w4=_lib.ptr(lw.get("gate_up4")), w4_sz=_lib.ptr(lw.get("gate_up4_sz")), ldw4=H // 2,
                                     w4_group=model.int4_group,
                                     w8_format=_lib.W8_INT8 if model.weight_format == "int8" else _lib.W8_FP8)
            else:
                la = _lib.LinearArgs(a=None, w=lw["gate_up"].data_ptr(), bias=None, resid=None, c=None, norm_w=None, norm_eps=0.0,
                                     m=m, n=n, k=H, lda=H, ldw=H, ldc=plan.inter, epilogue=_lib.EPI_SWIGLU, workspace=ws.data_ptr(),
                                     workspace_bytes=wsb, w8=_lib.ptr(lw.get("gate_up8")), w8_scale=_lib.ptr(lw.get("gate_up8_scale")),
                                     ldw8=H, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=H, c_hi=chi.data_ptr(),
                                     c_lo=clo.data_ptr(), ld_cplanes=plan.inter,
                                     w8_format=_lib.W8_INT8 if model.weight_format == "int8" else _lib.W8_FP8)
            _lib.check(lib.chatts_linear(la, stream.cuda_stream))

    sweep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        sweep()
    e1.record(stream)
    torch.cuda.synchronize()
    launches = reps * len(model.layers)
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches
    # algorithmic bytes per launch: bf16 weights + f32 x in + norm weights + f32 out (SURVEY.md 8d per-unit figure)
    fp8 = "gate_up8" in model.layers[0]
    int4 = m == 1 and "gate_up4" in model.layers[0]
    bytes_per_launch = n * H * (1 if fp8 else 2) + (n * 4 if fp8 else 0) + m * (H * 4 + H * 4 + plan.inter * 4)
    if int4:
        bytes_per_launch = n * H // 2 + n * (H // model.int4_group) * 8 + H * 4 + H * 4 + plan.inter * 4
    if m > 1:
        return dict(kernel=f"gemm_stream_kernel<4,{'fp8' if fp8 else 'bf16'}> (gate_up_proj + SwiGLU, M = {m} sequences, bf16 planes in/out)",
                    launches=launches, avg_us=avg_s * 1e6, bytes_per_launch=bytes_per_launch, gbs=bytes_per_launch / avg_s / 1e9)
    return dict(kernel=("gemv4_ldsx_kernel" if int4 else "gemv8_ldsx_kernel" if fp8 else "gemv_ldsx_kernel") +
                "<2,2,SWIGLU,NORM> (gate_up_proj + fused RMSNorm + SwiGLU)", launches=launches,
                avg_us=avg_s * 1e6, bytes_per_launch=bytes_per_launch, gbs=bytes_per_launch / avg_s / 1e9)


def prefill_mfma_gate_up(model, T, reps=1):
    """The prefill's dominant GEMM (gate_up, M = prompt tokens, gemm_ring_kernel on bf16 planes) between two HIP events on
    the launching stream, over every layer's weights: MFMA-issued flops = 2 passes (hi, lo) x 2 M N K."""
    import torch
    from chatts_amd import _lib
    lib, cfg, plan = model.lib, model.config, model.plan
    H, n = cfg.hidden_size, 2 * plan.inter
    if T < 96 or H % 64:
        return None
    x = torch.randn((T, H), dtype=torch.float32, device=model.device)
    hi = torch.empty((T, H), dtype=torch.bfloat16, device=model.device)
    lo = torch.empty((T, H), dtype=torch.bfloat16, device=model.device)
    out = torch.empty((T, plan.inter), dtype=torch.float32, device=model.device)
    stream = torch.cuda.current_stream()
    _lib.check(lib.chatts_split_bf16x2(x.data_ptr(), T, H, H, hi.data_ptr(), lo.data_ptr(), H, stream.cuda_stream))
    wsb = max(int(lib.chatts_linear_workspace(T, n, H)), 16)
    ws = torch.empty(wsb, dtype=torch.uint8, device=model.device)

    def sweep():
        for i, lw in enumerate(model.layers):
            la = _lib.LinearArgs(a=None, w=lw["gate_up"].data_ptr(), bias=None, resid=None, c=out.data_ptr(), norm_w=None,
                                 norm_eps=0.0, m=T, n=n, k=H, lda=H, ldw=H, ldc=plan.inter, epilogue=_lib.EPI_SWIGLU,
                                 workspace=ws.data_ptr(), workspace_bytes=wsb, a_hi=hi.data_ptr(), a_lo=lo.data_ptr(), ld_planes=H)
            la.w_tiled = _lib.ptr(model._tiled[i].get("gate_up")) if i < len(model._tiled) else None      # as the decoder calls it
            _lib.check(lib.chatts_linear(la, stream.cuda_stream))

    sweep()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        sweep()
    e1.record(stream)
    torch.cuda.synchronize()
    launches = reps * len(model.layers)
    avg_s = e0.elapsed_time(e1) * 1e-3 / launches
    issued = 2 * 2.0 * T * n * H
    return {"kernel": "gemm_ring_kernel (gate_up_proj + SwiGLU, M = prompt tokens)", "bound": "mfma", "avg_us": avg_s * 1e6,
            "achieved": issued / avg_s / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": issued / avg_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, "useful_tflops": issued / 2 / avg_s / 1e12,
            "launches_timed": launches,
            "note": "MFMA-issued flops (bf16x2: two passes per product) of the BEST projection timed alone; the whole prefill is prefill_e2e. "
                    "With these operand statistics the matrix pipe alone sustains 2.09 PF at a power-bound 2.07 GHz "
                    "(profiles/r5_mfma_power_probe.txt), this kernel's clock under load is 1.6-1.9 GHz (profiles/r5_ring_probe_first.txt)"}


def ts_encoder_roofline(model, ser, lengths, reps=20):
    """TS encoder (chatts_ts_encode: patchify + 5-layer MLP) between two HIP events on the launching stream, inputs resident.
    Bound: HBM on the MLP weights for P <= ~1k patches (212.6 MB at H = 5120), MFMA above.
    Algorithmic bytes per call = bf16 weights + f32 biases + touched position rows + series + f32 output."""
    import torch
    enc = model.ts_encoder
    out = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=lengths)
    P = sum(o.shape[0] for o in out)
    H = enc.hidden_size
    stream = torch.cuda.current_stream()
    enc.replay_last()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        enc.replay_last()
    e1.record(stream)
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / reps
    w_bytes = sum(enc.layer_in_features(l) * H * 2 + H * 4 for l in range(enc.num_layers))
    pos_rows = min(max(lengths) + 1, enc.max_sequence_length + 1) * enc.embedding_dim * 4 if enc.mode == 1 else 0
    byts = w_bytes + pos_rows + 8 * sum(lengths) + 4 * P * H
    flops = 2.0 * P * sum(enc.layer_in_features(l) * H for l in range(enc.num_layers))
    hbm_floor_us, mfma_floor_us = byts / HBM_PEAK_GBS / 1e3, 2 * flops / MFMA_BF16_PEAK_TFLOPS / 1e6     # bf16x2: two MFMA passes
    bound = "hbm" if hbm_floor_us >= mfma_floor_us else "mfma"
    res = {"kernel": "chatts_ts_encode (ts_patchify + 5 x gemm_{stream,ring}_kernel on bf16 planes, GELU fused)", "patches": P,
           "bound": bound, "avg_us": avg_s * 1e6, "bytes_per_call": byts, "flops_per_call": flops, "calls_timed": reps}
    if bound == "hbm":
        res.update(achieved=byts / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=byts / avg_s / 1e9 / HBM_PEAK_GBS)
    else:
        res.update(achieved=2 * flops / avg_s / 1e12, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s",
                   frac=2 * flops / avg_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, note="MFMA-issued flops (bf16x2)")
    return res


def cpu_ts_encode(model, ser, reps=5):
    """SURVEY.md section 8(d) "report TS-encoder ms": TimeSeriesEmbedding.forward on the host cores of THIS box, beside
    ts_encoder_roofline - the oracle's patch-feature restatement (numpy, chatts_vllm.py:94-183) followed by the reference's module
    stack itself (nn.Linear / nn.GELU, :83-91) as float32 torch calls with the thread count the decoder leg picked, weights copied
    back from the GPU.  Median of `reps` calls after one warm-up, same series as the GPU line."""
    import torch
    from oracle import from_device, ts_embedding
    tsw = {k[len("ts_encoder."):]: v for k, v in from_device.ts_encoder_state_dict(model).items()}
    x = ser.float().cpu().numpy()
    cfg = model.config.ts
    n_layers = int(cfg["num_layers"])
    lin = [(torch.from_numpy(tsw[f"mlp.{2 * l}.weight"]), torch.from_numpy(tsw[f"mlp.{2 * l}.bias"])) for l in range(n_layers)]

    def once():
        t0 = time.perf_counter()
        feat, _ = ts_embedding.patch_features(x, cfg, tsw.get("position_embedding.weight"))
        h = torch.from_numpy(feat)
        with torch.no_grad():
            for l, (w, b) in enumerate(lin):
                h = torch.nn.functional.linear(h, w, b)
                if l < n_layers - 1:
                    h = torch.nn.functional.gelu(h)
        return time.perf_counter() - t0, int(h.shape[0])

    once()
    runs = [once() for _ in range(reps)]
    return median([r[0] for r in runs]) * 1e3, runs[0][1]


def cpu_baseline(model, prompt_tokens, depth=8, ser=None):
    """The reference's own decoder dependency - stock `transformers` Qwen2ForCausalLM / Qwen3ForCausalLM, float32, eager
    attention (oracle/hf_reference.py) - timed on the host cores of THIS box.  Bounded sample: the model's real widths at
    `depth` layers (weights copied back from the GPU, so both arms hold identical values), prefill of the real prompt length +
    4 decode tokens, then the same with the first depth/2 layers only; per-layer and fixed (embedding, lm_head) costs are
    fitted from the two depths and extrapolated to the full depth."""
    import torch
    from oracle import from_device, hf_reference
    cfg = model.config
    ncores_box = os.cpu_count() or 1
    t_start = time.time()
    sd = from_device.head_tensors(model)
    depth = min(depth, cfg.num_hidden_layers)
    for l in range(depth):
        sd.update(from_device.layer_tensors(model, l))
    m = hf_reference.build(cfg, sd, num_layers=depth)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((prompt_tokens, cfg.hidden_size), generator=g) * 0.02
    # thread count: all host cores is the default, but torch's intra-op pool collapses on many-core boxes for the
    # memory-bound decode GEMVs; pick the fastest of a few counts on a decode probe and REPORT the count used.
    best = None
    for nt in sorted({ncores_box, min(ncores_box, 64), min(ncores_box, 32), min(ncores_box, 16)}, reverse=True):
        torch.set_num_threads(nt)
        _, _, _, td = hf_reference.prefill_and_decode(m, emb[:8], 3, embed_table_rows=4096)
        dt = median(td[1:])
        if best is None or dt < best[0]:
            best = (dt, nt)
    ncores = best[1]
    torch.set_num_threads(ncores)
    res = {}
    layers_all = m.model.layers
    for L in (depth, depth // 2):
        m.model.layers = layers_all[:L]
        _, _, t_pre, td = hf_reference.prefill_and_decode(m, emb, 4, embed_table_rows=4096)
        res[L] = (t_pre, median(td))
    m.model.layers = layers_all
    half = depth // 2
    per_layer_dec = (res[depth][1] - res[half][1]) / (depth - half)
    fixed_dec = res[depth][1] - depth * per_layer_dec
    per_layer_pre = (res[depth][0] - res[half][0]) / (depth - half)
    fixed_pre = res[depth][0] - depth * per_layer_pre
    Lfull = cfg.num_hidden_layers
    dec_s = fixed_dec + Lfull * per_layer_dec
    pre_s = fixed_pre + Lfull * per_layer_pre
    import transformers
    ts_ms, ts_patches = (None, None)
    if ser is not None:
        try:
            ts_ms, ts_patches = cpu_ts_encode(model, ser)
        except Exception as e:      # reported, never fatal
            ts_ms, ts_patches = f"failed: {type(e).__name__}: {e}", None
    return dict(value=1.0 / dec_s, ts_encode_ms=ts_ms, ts_encode_patches=ts_patches, unit="tokens/s", cores=ncores, kind="reference", host_cores=ncores_box,
                sample=(f"stock transformers {transformers.__version__} {type(m).__name__} (the reference's decoder dependency), float32, eager "
                        f"attention, on {ncores} of {ncores_box} host threads (fastest of a decode probe); {cfg.name} widths, depths {depth} and {half} "
                        f"of {Lfull} layers measured (prefill {prompt_tokens} tok + 4 decode tok each), linearly extrapolated to {Lfull} "
                        f"layers; decode {per_layer_dec * 1e3:.1f} ms/layer + {fixed_dec * 1e3:.0f} ms embedding + lm_head"),
                ttft_s_extrapolated=pre_s, measured={"depth": depth, "prefill_s": res[depth][0], "decode_s_per_token": res[depth][1]},
                wall_s=time.time() - t_start)


def workload_key(args):
    """file-name key of a workload: model tag + series x length (or 'mixed') + weight format + batch"""
    tag = {"chatts-14b": "14b", "chatts-8b": "8b"}.get(args.model)
    if tag is None:
        return None
    shape = f"{args.series}xmixed" if getattr(args, "lengths", "uniform") == "mixed" else f"{args.series}x{args.length}"
    return f"{tag}_{shape}_{args.weights}_b{max(1, args.batch)}"


def parity_reference(args):
    """The committed FULL-DEPTH oracle run of this workload (tools/parity_full_depth.py: CPU float32 oracle, all layers, same
    inputs, seed 0), or None.  Files are keyed by the workload; the newest round's run wins (round 5 re-ran every workload on the
    rebuilt prefill GEMM, whose split-K choices - and with them the summation order of o_proj / down_proj - changed)."""
    key = workload_key(args)
    if key is None:
        return None, None
    cands = [os.path.join(ROOT, "profiles", f"r{rnd}_parity_{key}_full.json") for rnd in (6, 5, 4, 3)]
    if getattr(args, "precision", "bf16x2") == "f16q":      # the opt-in mode's own full-depth run first (its recorded logits errors are its own)
        cands.insert(0, os.path.join(ROOT, "profiles", f"r6_parity_{key}_f16q_full.json"))
    if args.weights == "bf16" and args.batch <= 1 and getattr(args, "lengths", "uniform") == "uniform":
        cands.append(os.path.join(ROOT, "profiles", f"r2_parity_{key.split('_')[0]}_full.json"))
    for path in cands:
        if os.path.exists(path):
            with open(path) as f:
                ref = json.load(f)
            if (ref.get("series"), ref.get("length")) == (args.series, args.length) or ref.get("workload_key") == key:
                return path, ref
    return None, None


def parity_check(args, toks):
    """Compare the tokens this run generated with the committed full-depth oracle run of the SAME workload.  toks: the token list
    (batch 1) or one list per cache slot (batched workloads: the oracle run records the slots it covered)."""
    if getattr(args, "precision", "bf16x2") not in ("bf16x2", "f16q"):
        return False, {"reason": "speed mode: logits are outside the 1e-3 tolerance by construction (profiles/r2_speed_mode_14b.json, "
                                 "profiles/r4_fp8_speed_mode.json)"}
    if args.layers is not None:
        return False, {"reason": "truncated depth (debug run)"}
    path, ref = parity_reference(args)
    if ref is None:
        return False, {"reason": "no committed full-depth oracle run for this workload"}
    want = ref["tokens_oracle"]
    if isinstance(want, dict):              # batched: {slot: tokens}
        pairs = [(toks[int(sl)], w) for sl, w in sorted(want.items(), key=lambda kv: int(kv[0]))]
        slots = sorted(int(sl) for sl in want)
    else:
        pairs, slots = [(toks, want)], None
    n = min(min(len(w), len(t)) for t, w in pairs)
    ok = n > 0 and all(t[:n] == w[:n] for t, w in pairs)
    out = {"source": os.path.relpath(path, ROOT), "tokens_compared": n, "tokens_match": ok,
           "first_token_logits_rel_err_recorded": ref.get("first_token_logits_rel_err"),
           "max_step_logits_rel_err_recorded": ref.get("max_step_logits_rel_err", max(ref.get("step_logits_rel_err", [0.0]))),
           "max_abs_err_over_max_logit_recorded": ref.get("max_abs_err_over_max_logit"), "tolerance": ref.get("tolerance", 1e-3)}
    if slots is not None:
        out["slots_compared"] = slots
    return ok, out


def workload_name(args, world):
    names = {"chatts-14b": "ChatTS-14B", "chatts-8b": "ChatTS-8B"}
    if args.model not in names or args.layers is not None:
        return f"DEBUG {args.model} layers={args.layers}"
    shape = (f"{args.series} series x mixed lengths 64-1024 (rng 1234, ragged tails forced)" if args.lengths == "mixed"
             else f"{args.series} series x {args.length} steps")
    return f"{names[args.model]} {args.weights} weights, {shape}, greedy decode, TP={world}"


def bench_batched(args, model, cfg, comm, world, device):
    """BASELINE.json config 5 shape on this many GPUs: `--batch` different prompts (each `--series` x `--length`) are admitted
    into the cache slots (TTFT = per-request processor + TS encode + merge + prefill + first token, p50 over the requests),
    then decode TOGETHER: a step = one B-wide decode step (one hipGraph: M = B weight-streaming GEMMs, per-sequence
    attention, per-sequence token selection); value = B * steps / max-over-ranks time."""
    import torch
    import torch.distributed as dist
    B = args.batch
    proc, prompt, reqs, lengths = build_batched_requests(cfg, B, args.series, args.length)
    budget = 1 + args.warmup + args.steps
    Bf = model.buf
    Bf["pos_all"].zero_(); Bf["step_all"].zero_(); Bf["token_all"].zero_()
    comm.barrier()
    torch.cuda.synchronize()
    t_admit0 = time.perf_counter()
    T, ttfts, packs = admit_batched(model, proc, prompt, reqs, budget, pack=not args.no_pack)
    admit_ms = (time.perf_counter() - t_admit0) * 1e3
    for _ in range(args.warmup):
        model.batched_step()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.batched_step()
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    toks = Bf["out_tokens_all"][:, :budget].tolist()
    roof = roofline_gate_up(model, m=B)
    step_bytes = model.weight_bytes_local()
    names = {"chatts-14b": "ChatTS-14B", "chatts-8b": "ChatTS-8B"}
    label = (f"{names.get(args.model, args.model)} {args.weights} weights, {args.series} series x {args.length} steps, batch {B} continuous "
             f"prompts, TP={world}") if args.layers is None and args.model in names else f"DEBUG {args.model} layers={args.layers}"
    res = {
        "metric": f"generated tokens/sec (aggregate over {B} sequences decoding together) + p50 TTFT, {names.get(args.model, args.model)}, "
                  f"{args.series}x{args.length}-step TS prompts, TP=N",
        "value": B * args.steps / dt, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16" if args.weights == "bf16" else f"{'int8' if args.weights == 'int8' else 'fp8-e4m3'} weights (pow2 row scales, exactly representable in bf16), bf16x2 MFMA / f32 accumulate",
        "data": "synthetic",
        "config": {"workload": label, "model": args.model, "batch": B, "prompt_tokens": T, "ts_patches": sum((L + 15) // 16 for L in lengths),
                   "parallelism": f"tp{world}", "decode_graph": model.graph_capturable(), "kv_block": model.kv_block_size or None,
                   "tp_exchange": None if world == 1 else ("p2p over IPC-mapped buffers (csrc/tp.hip): decode exchanges inside the o_proj / down_proj GEMV launches, prefill sums = two-shot "
                                                          "bulk kernel inside chatts_decoder_prefill" if model._tp is not None else "rccl, host-driven"),
                   "weights_note": None if args.weights == "bf16" else
                   "fp8 copies are streamed by the decode GEMMs and widened to bf16 while staged (no v_mfma fp8 issue: the 1e-3 "
                   "logit bar needs f32-exact products of the f32 activations); prefill keeps the bf16 copy" if args.precision != "fp8" else
                   "SPEED MODE precision=fp8 - NOT the parity-grade line: prefill chunks and the TS encoder quantise activations per row to "
                   "e4m3 and multiply fp8 x fp8 (v_mfma_scale_f32_16x16x128_f8f6f4); decode steps as in the default",
                   "first_tokens_seq0": toks[0][:8]},
        "ttft_ms_p50": median(ttfts), "batch_admit_ms_total": admit_ms, "packed_prefill_passes": packs,
        "per_sequence_tokens_per_s": args.steps / dt,
        "decode_hbm_gbs_per_gpu": step_bytes / (dt / args.steps) / 1e9,
        "decode_hbm_frac_of_8TBs": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "achieved": roof["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roof["gbs"] / HBM_PEAK_GBS,
                     "traffic": None, "kernel": roof["kernel"], "avg_us": roof["avg_us"], "bytes_per_launch": roof["bytes_per_launch"],
                     "launches_timed": roof["launches"]},
    }
    res["parity_checked"], res["parity"] = parity_check(args, toks)
    res["config"]["rccl_world_size"] = dist.get_world_size() if world > 1 else 1
    # the step also reads every sequence's KV cache (float32 K and V rows of all layers): at 16 sequences x ctx 1.2k that is a
    # third of the step's bytes, so the weight-only rate above understates the HBM stream
    ctx_mid = T + 1 + args.warmup + args.steps / 2.0
    kv_bytes = B * ctx_mid * model.plan.nkv * 128 * 4 * 2 * cfg.num_hidden_layers
    res["decode_kv_bytes_per_step"] = kv_bytes
    res["decode_hbm_gbs_incl_kv"] = (step_bytes + kv_bytes) / (dt / args.steps) / 1e9
    res["decode_hbm_frac_of_8TBs_incl_kv"] = res["decode_hbm_gbs_incl_kv"] / HBM_PEAK_GBS
    try:        # HBM bytes per launch of the dominant kernel from the separate rocprofv3 --pmc FETCH_SIZE pass of THIS workload
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get("batched", {}).get(workload_key(args))
        if pmc and world == 1:
            ok, why = pmc_traffic_entry(pmc)
            res["roofline"]["traffic"] = ok["hbm_bytes_per_launch"] if ok else None
            res["roofline"]["traffic_source"] = ok["source"] if ok else why
    except (OSError, ValueError, KeyError):
        pass
    try:        # the TS encoder at the patch count of the WHOLE batch (config 5: 16 x 8 x 1024 points = 8192 patches: MFMA-bound)
        sers = [proc(text=[prompt], timeseries=r, padding=True, return_tensors="pt")["timeseries"] for r in reqs]
        ser_all = torch.cat(sers, dim=0).to(device)
        res["ts_encoder_roofline"] = ts_encoder_roofline(model, ser_all, list(lengths) * B, reps=5)
    except Exception as e:
        res["ts_encoder_roofline"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (the driver's call shape; the reference's is one process asking vLLM for
    tensor_parallel_size=N, demo/demo_vllm.py:30): re-execute this command under torch.distributed.run, one rank per GPU on this
    node, rendezvous on 127.0.0.1.  Rank 0 of the re-executed job prints the one JSON line."""
    from chatts_amd.tp_spawn import free_port
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without a launcher: re-executing under torch.distributed.run ({n} ranks)")
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def launch_check(args):
    """--launch-check: prove the N-rank launch path without touching a model (CPU-testable with gloo): every rank joins the
    process group, the ranks sum a one, rank 0 prints what it saw."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CHATTS_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("CHATTS_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{torch.cuda.current_device()}"))
        else:
            dist.init_process_group(backend=backend)
        one = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        seen = int(one.item())
        backend_name = dist.get_backend()
        ws = dist.get_world_size()
        dist.destroy_process_group()
    else:
        backend_name, ws = None, 1
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "requested_gpus": args.gpus, "ranks_summed": seen,
                          "rccl_world_size": ws, "backend": backend_name}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="chatts-14b")
    ap.add_argument("--series", type=int, default=8)
    ap.add_argument("--length", type=int, default=256)
    ap.add_argument("--lengths", default="uniform", choices=["uniform", "mixed"], help="mixed = BASELINE.json config 4: --series lengths "
                    "drawn from rng(1234).integers(64, 1025) with ragged tails forced (use with --series 30)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: truncate depth (result marked invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ttft-runs", type=int, default=5)
    ap.add_argument("--max-ctx", type=int, default=2048, help="KV-cache length (longer prompts, e.g. --series 30, need more)")
    ap.add_argument("--batch", type=int, default=1, help="> 1: continuous-batching workload (BASELINE.json config 5): B prompts decode "
                    "together; a step = one B-wide decode step; value = aggregate tokens/s")
    ap.add_argument("--precision", default="bf16x2", choices=["bf16x2", "bf16", "fp8", "f16q"], help="f16q = the opt-in PARITY-GRADE prefill on the f16 + fp8 matrix pipes (slower; DESIGN.md 14); bf16 / fp8 = the optional SPEED modes (bf16: "
                    "single-pass bf16 activations in the prefill GEMMs; fp8: --weights fp8 only, prefill GEMMs and the TS encoder on the fp8 "
                    "matrix pipe); not parity grade, labelled as such; default bf16x2")
    ap.add_argument("--kv-block", type=int, default=0, help="block-paged KV cache with this many positions per block (0 = one "
                    "contiguous cache per slot, the default)")
    ap.add_argument("--no-pack", action="store_true", help="--batch: admit the prompts one by one instead of packed prefill passes")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8", "int8", "int4"],
                    help="fp8 = BASELINE.json config 5 weight format, int4 = the GPTQ-Int4 checkpoint's (NOT the headline: separate workloads)")
    ap.add_argument("--launch-check", action="store_true", help="only prove the N-rank launch path (no model): rank 0 prints the world it saw")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)          # does not return
    if args.launch_check:
        return launch_check(args)

    import torch
    import torch.distributed as dist
    from chatts_amd import _lib
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.tp import Comm, LocalComm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: pass the same N to both "
                         "(or run `python bench.py --gpus N` alone: it launches its own ranks)")
    # test hooks (functional TP check on a single-GPU box): CHATTS_FORCE_DEVICE pins every rank to one device,
    # CHATTS_DIST_BACKEND=gloo replaces RCCL.  Never set by the driver.
    dev_index = int(os.environ.get("CHATTS_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("CHATTS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend=backend)
        comm = Comm()
    else:
        comm = LocalComm()

    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = build_inputs(cfg, args.series, args.length, args.lengths)
    t0 = time.time()
    max_ctx = args.max_ctx
    if args.batch <= 1:         # the cache must hold prompt + every generated token (config 4's prompt alone is ~2.2k tokens)
        n_ids = len(proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")["input_ids"][0])
        need = n_ids - 2 * len(lengths) + sum((L + 15) // 16 for L in lengths) + args.warmup + args.steps + 16
        max_ctx = max(max_ctx, -(-need // 256) * 256)
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, device=device, comm=comm, max_ctx=max_ctx,
                                             max_prefill_tokens=1024, use_graph=not args.no_graph,
                                             weight_format=args.weights, max_batch=max(1, args.batch), precision=args.precision,
                                             kv_block_size=args.kv_block or None)
    torch.cuda.synchronize()
    log(f"[bench] {args.model} TP={world} materialised in {time.time() - t0:.1f}s, "
        f"{model.weight_bytes_local() / 1e9:.2f} GB decoder weights on this rank")
    if args.batch > 1:
        result = bench_batched(args, model, cfg, comm, world, device)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # TTFT + warm-up run twice at most: under TP a peer-to-peer exchange that stalls on this node (bounded spins, status word) is
    # replaced by the host-driven RCCL path on every rank and the stage is repeated - the line then says tp_exchange = rccl
    # first-run-readiness for a multi-GPU node: which device each rank sits on, whether it can reach its peers (hipDeviceCanAccessPeer as
    # torch reports it) and, at the end, the exchange's status word - so that a fallback or a stall is explained by the record itself
    tp_devices = None
    if world > 1:
        n_dev = torch.cuda.device_count()
        mine = {"rank": rank, "device": dev_index, "device_count": n_dev,
                "can_access_peer": [bool(torch.cuda.can_device_access_peer(dev_index, d)) if d != dev_index else True for d in range(n_dev)]}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        tp_devices = gathered
    tp_exchange = None if world == 1 else ("p2p over IPC-mapped buffers (csrc/tp.hip): decode exchanges inside the o_proj / down_proj GEMV launches, prefill sums = two-shot "
                                                          "bulk kernel inside chatts_decoder_prefill" if model._tp is not None else "rccl, host-driven")
    if world > 1 and model._tp is not None and rank == 0 and os.environ.get("CHATTS_BENCH_INJECT_P2P_STALL"):
        # test hook (tools/jobs/tp2_single_device.sh): rank 0 enters a collective alone - it times out and leaves the exchange broken
        model._tp.all_reduce(torch.ones(64, device=device))
    # the release form of the prefill-sized sums is decided per communicator when the exchange is created (P2PExchange.first_contact:
    # ranks on different devices use the system-scope fence unless >= 64 test sums validated the light form on these links); the line
    # records that decision.  Second line of defence here: if the warm-up tokens still differ from the committed oracle run under the
    # light form, every rank switches to the fence and the stage is repeated
    tp_release = None
    if world > 1 and model._tp is not None:
        tp_release = model._tp.release_note or model._tp.bulk_release()
    _, ref_run = parity_reference(args)
    ref_toks = ref_run.get("tokens_oracle") if ref_run else None
    for attempt in range(3):
        # ---- TTFT: processor -> H2D -> TS encoder -> merge -> prefill -> first token (p50) ---------------------
        ttfts, enc_ms = [], []
        T = None
        for i in range(args.ttft_runs + 1):
            comm.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
            ids = inputs["input_ids"][0].tolist()
            ser = inputs["timeseries"].to(device)
            mm = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
            full = model.expand_input_ids(ids, [(L + 15) // 16 for L in lengths])
            T = len(full)
            emb = model.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
            model.reset()
            last = model.prefill(emb, 0, for_next_token=True)        # what generate() does: next token + KV cache, no hidden states
            model.buf["pos"].fill_(T)
            model._first_token(last)
            first = model.buf["out_tokens"][:1].tolist()
            dt = (time.perf_counter() - t0) * 1e3
            if i > 0:                                   # run 0 is the warm-up
                ttfts.append(dt)
        ttft = median(ttfts)
        for i in range(args.ttft_runs):                 # the TS encoder alone, host call to results ready (its own loop: a sync inside
            torch.cuda.synchronize()                    # the TTFT region would stall the launch queue the real generate() keeps full)
            te0 = time.perf_counter()
            model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
            torch.cuda.synchronize()
            enc_ms.append((time.perf_counter() - te0) * 1e3)

        # ---- decode: W warm-up steps (captures the hipGraph), then exactly K timed steps ------------------------
        for _ in range(args.warmup):
            model.decode_step()
        if world == 1 or model._tp is None:
            break
        torch.cuda.synchronize()
        bad = torch.tensor([float(model._tp.status() != 0)], device=device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad.item() == 0:
            got = model.buf["out_tokens"][:1 + args.warmup].tolist()
            k = min(len(got), len(ref_toks)) if isinstance(ref_toks, list) else 0
            inject = attempt == 0 and bool(os.environ.get("CHATTS_BENCH_INJECT_RELEASE_MISMATCH"))     # test hook (tools/jobs/r5_tp_self_launch.sh)
            wrong = torch.tensor([float((k > 0 and got[:k] != ref_toks[:k]) or inject)], device=device)
            dist.all_reduce(wrong, op=dist.ReduceOp.MAX)
            if wrong.item() == 0 or model._tp.bulk_release() != "light":
                break
            log("[bench] tokens differ from the committed oracle run with the light release of the bulk sums: repeating with the fence")
            model._tp.set_bulk_release("fence")
            _lib.set_option("TP_BULK_FENCE", 1)
            tp_release = "fence (the light release gave different tokens on this node)"
            continue
        log("[bench] the peer-to-peer exchange timed out on some rank: falling back to RCCL for the decode-sized sums")
        ex = model._tp
        model.attach_exchange(None)
        model.use_p2p = False
        ex.close()
        tp_exchange = "rccl (p2p exchange stalled)"
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.decode_step()
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    toks = model.buf["out_tokens"][:1 + args.warmup + args.steps].tolist()
    tok_s = args.steps / dt
    ms_step = dt / args.steps * 1e3

    tp_status = None
    if world > 1:
        st_word = model._tp.status() if getattr(model, "_tp", None) is not None else -1          # -1: no p2p exchange attached (host-driven sums)
        gathered = [None] * world
        dist.all_gather_object(gathered, int(st_word))
        tp_status = gathered
    roof = roofline_gate_up(model)
    traffic, traffic_note, pmc = None, None, None     # HBM bytes per launch from the separate rocprofv3 --pmc pass (TP=1 shape only)
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if world == 1 and args.model == "chatts-14b" and args.weights == "bf16":
            ok, traffic_note = pmc_traffic_entry(pmc)
            traffic = ok["hbm_bytes_per_launch"] if ok else None
    except Exception as e:
        traffic_note = f"profiles/pmc_traffic.json unreadable: {e}"
    step_bytes = model.weight_bytes_local()
    result = {
        "metric": f"generated tokens/sec (greedy, batch 1) + p50 TTFT, {'ChatTS-8B' if args.model == 'chatts-8b' else 'ChatTS-14B'}, "
                  f"{args.series}x{'mixed(64-1024)' if args.lengths == 'mixed' else args.length}-step TS prompt, TP=N",
        "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp8": "fp8-e4m3 weights (pow2 row scales), f32 math",
                  "int8": "int8 weights (pow2 row scales: a lossless encoding of the bf16 matrix), f32 math",
                  "int4": "int4 codes + fp16 group scales (round-to-nearest, groups of 128) = a bf16 matrix, f32 math"}[args.weights], "data": "synthetic",
        "config": {"workload": workload_name(args, world),
                   "model": args.model, "prompt_tokens": T, "ts_patches": sum((L + 15) // 16 for L in lengths),
                   "parallelism": f"tp{world}", "batch": 1, "decode_graph": model.graph_capturable(), "kv_block": model.kv_block_size or None,
                   "tp_exchange": tp_exchange, "rccl_world_size": dist.get_world_size() if world > 1 else 1,
                   "tp_devices": tp_devices, "tp_status": tp_status, "tp_release": tp_release,
                   "precision": ("bf16 weights; f32 activations, KV cache and accumulation (bf16x2 MFMA split in "
                                 "prefill, exact f32 FMA in decode)") if model.precision == "bf16x2" else
                                ("PARITY GRADE, opt-in: prefill projections on the f16q split (f16 MFMA + block-scaled e4m3 MFMA, chatts_linear_f16q), "
                                 "everything else as the default; slower than the default (DESIGN.md 14)") if model.precision == "f16q" else
                                ("SPEED MODE precision=bf16 - NOT the parity-grade line: prefill GEMMs multiply bf16-rounded activations "
                                 "(one MFMA pass), logits ~3e-2 of the default mode (profiles/r2_speed_mode_14b.json); decode as in the default"),
                   "first_tokens": toks[:8],
                   "weight_bytes_decode_stream": step_bytes, "weight_bytes_tiled_prefill_copies": model.tiled_weight_bytes_local(),
                   "weight_bytes_f16q_copies": model.f16q_weight_bytes_local()},
        "ttft_ms_p50": ttft, "ts_encode_ms_p50": median(enc_ms),
        "decode_hbm_gbs_per_gpu": step_bytes / (dt / args.steps) / 1e9,
        "decode_hbm_frac_of_8TBs": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        "decode_frac_of_bf16_mfma_roofline": (2.0 * step_bytes / 2 / (dt / args.steps)) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
        "roofline": {"bound": "hbm", "achieved": roof["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": roof["gbs"] / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": (pmc or {}).get("source") if traffic else traffic_note, "kernel": roof["kernel"],
                     "avg_us": roof["avg_us"], "bytes_per_launch": roof["bytes_per_launch"],
                     "launches_timed": roof["launches"]},
    }
    result["parity_checked"], result["parity"] = parity_check(args, toks)
    try:        # north_star: "rocprof HBM GB/s on the TS-encoder" - the encoder's own roofline line, at the workload's shape
        result["ts_encoder_roofline"] = ts_encoder_roofline(model, ser, lengths)
        if world == 1 and args.model == "chatts-14b" and (args.series, args.length, args.lengths) == (8, 256, "uniform"):
            ok, why = pmc_traffic_entry((pmc or {}).get("ts_encoder"))
            result["ts_encoder_roofline"]["traffic"] = ok["hbm_fetch_bytes_per_call"] if ok else None
            result["ts_encoder_roofline"]["traffic_source"] = ok["source"] if ok else why
    except Exception as e:
        result["ts_encoder_roofline"] = {"error": f"{type(e).__name__}: {e}"}
    try:        # secondary evidence: MFMA utilisation of the prefill's dominant GEMM (north_star asks for it beside the HBM rate)
        result["prefill_roofline"] = prefill_mfma_gate_up(model, T)
    except Exception as e:
        result["prefill_roofline"] = {"error": f"{type(e).__name__}: {e}"}
    # the prefill END TO END: useful flops of the prompt (2 x decoder parameters streamed per token on this rank; attention and the
    # lm_head of the last row left out) over the measured TTFT (processor + TS encoder + merge + prefill + first token) - the figure
    # that says how far the whole prefill is from the matrix pipe, not its best kernel
    params = sum(lw[k].numel() for lw in model.layers for k in ("qkv", "o", "gate_up", "down") if k in lw)
    useful = 2.0 * T * params
    result["prefill_e2e"] = {"useful_tflops": useful / (ttft * 1e-3) / 1e12, "frac_of_bf16_peak": useful / (ttft * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                             "issued_frac_of_bf16_peak": (2.0 if model.precision == "bf16x2" else 1.0) * useful / (ttft * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                             "ttft_ms": ttft, "prompt_tokens": T,
                             "note": "issued = useful x 2 in the parity-grade bf16x2 mode (hi and lo pass of every product)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(model, T, ser=ser)
        except Exception as e:          # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
