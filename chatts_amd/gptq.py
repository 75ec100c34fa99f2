"""GPTQ-Int4 checkpoints (the published ``ChatTS-14B-GPTQ-Int4``; NetManAIOps/ChatTS README.md:52,262-263; the reference hands
``quant_config`` to vLLM's GPTQ linear layers, chatts/vllm/chatts_vllm.py:475,481 - kernels NOT IN REFERENCE).

AutoGPTQ tensor layout of one ``nn.Linear(K -> N)`` (bits = 4, 8 values per int32, little end first):
    qweight int32 [K / 8, N]        q[k, n]  = (qweight[k // 8, n] >> 4 * (k % 8)) & 15
    qzeros  int32 [K / g, N / 8]    z0[j, n] = (qzeros[j, n // 8] >> 4 * (n % 8)) & 15
    scales  fp16  [K / g, N]
    g_idx   int32 [K]               group of input channel k (k // g unless the checkpoint was quantised with desc_act)
    W[n, k] = scales[g_idx[k], n] * (q[k, n] - (z0[g_idx[k], n] + 1))       ("gptq" format; "gptq_v2" stores z without the -1)

What the engine holds: the dequantised matrix rounded to bf16 - the weight format of every kernel - so prefill (bf16 MFMA),
decode and the oracle all see the same numbers, exactly as with the fp8 copies.  `pack_rows` additionally
keeps the 4-bit codes in the row-major layout the weight-streaming decode GEMV wants (K contiguous per output row).
"""
import numpy as np
import torch

SUFFIXES = ("qweight", "qzeros", "scales", "g_idx")


def unpack_int4(packed, axis):
    """int32 tensor holding 8 4-bit values per word along `axis` -> uint8 tensor, that axis 8x longer."""
    p = packed.to(torch.int32)
    shifts = torch.arange(0, 32, 4, dtype=torch.int32, device=p.device)
    if axis == 0:       # [R, C] -> [R * 8, C]
        v = (p[:, None, :] >> shifts[None, :, None]) & 15
        return v.reshape(p.shape[0] * 8, p.shape[1]).to(torch.uint8)
    v = (p[:, :, None] >> shifts[None, None, :]) & 15
    return v.reshape(p.shape[0], p.shape[1] * 8).to(torch.uint8)


def codes(qweight, qzeros, scales, g_idx=None, group_size=128, v2=False):
    """-> (codes uint8 [N, K], scale f32 [N, G], zero f32 [N, G]) in the nn.Linear orientation, or None when the checkpoint's
    channel -> group map is not the sequential k // group_size (act-order): then only the dequantised bf16 matrix is used."""
    K = qweight.shape[0] * 8
    if g_idx is not None and not torch.equal(g_idx.to(torch.long).cpu(), torch.arange(K) // group_size):
        return None
    q = unpack_int4(qweight, 0).t().contiguous()                               # [N, K]
    N = q.shape[0]
    z = unpack_int4(qzeros, 1)[:, :N].to(torch.float32) + (0.0 if v2 else 1.0)  # [G, N]
    return q, scales.float().t().contiguous(), z.t().contiguous()


def dequantize(qweight, qzeros, scales, g_idx=None, group_size=128, v2=False):
    """-> float32 [N, K] (the nn.Linear.weight layout), every element EXACT in float32 (fp16 scale x small integer)."""
    q = unpack_int4(qweight, 0).to(torch.int32)                # [K, N]
    K, N = q.shape
    z = unpack_int4(qzeros, 1).to(torch.int32)[:, :N]           # [G, N]
    if not v2:
        z = z + 1
    if g_idx is None:
        g_idx = torch.arange(K, device=q.device) // group_size
    g = g_idx.to(torch.long)
    w = scales.float()[g] * (q - z[g]).float()                  # [K, N]
    return w.t().contiguous()


def is_gptq_config(cfg_extra):
    qc = (cfg_extra or {}).get("quantization_config") or {}
    return qc.get("quant_method") == "gptq"


def dequantized_pairs(pairs, quant_cfg, has_g_idx=None):
    """(name, tensor) stream of a GPTQ checkpoint -> stream where every quantised Linear appears as `<module>.weight` (bf16);
    other tensors pass through.  Also yields nothing for the consumed qweight / qzeros / scales / g_idx entries.
    has_g_idx: whether the checkpoint stores `<module>.g_idx` tensors, when the caller knows (the safetensors key index does):
    False = a module is emitted as soon as its three tensors are there; None = unknown - modules wait for a g_idx and those that
    never get one are flushed at the end of the stream (which buffers the whole int4 checkpoint for g_idx-less files)."""
    if int(quant_cfg.get("bits", 4)) != 4:
        raise ValueError(f"GPTQ bits={quant_cfg.get('bits')} is not supported (4-bit checkpoints only)")
    gs = int(quant_cfg.get("group_size", 128))
    v2 = quant_cfg.get("checkpoint_format", "gptq") == "gptq_v2"
    pending = {}

    def emit(base, d):
        g = gs if gs > 0 else d["qweight"].shape[0] * 8
        w = dequantize(d["qweight"], d["qzeros"], d["scales"], d.get("g_idx"), g, v2)
        yield base + ".weight", w.to(torch.bfloat16)
        # the int4 decode GEMV takes ONE group size for all projections of the model (a power of two >= 16): a per-column
        # checkpoint (group_size = -1: g = K, different per module) keeps streaming the dequantised bf16 weights instead
        cz = codes(d["qweight"], d["qzeros"], d["scales"], d.get("g_idx"), g, v2) if gs >= 16 and gs & (gs - 1) == 0 else None
        if cz is not None:                  # (None: act-order channel -> group map)
            yield base + ".gptq_codes", cz  # consumed by ChatTSForCausalLM.load_weights (int4 decode GEMV)

    for name, t in pairs:
        base, _, suffix = name.rpartition(".")
        if suffix not in SUFFIXES:
            yield name, t
            continue
        d = pending.setdefault(base, {})
        d[suffix] = t
        # AutoGPTQ checkpoints store g_idx for every module (trivial when desc_act is false) and it may arrive after the other
        # three tensors: a module is complete when all four are there; one that never gets a g_idx is emitted by the flush below
        need = ("qweight", "qzeros", "scales") if has_g_idx is False else ("qweight", "qzeros", "scales", "g_idx")
        if all(k in d for k in need):
            if has_g_idx is False and quant_cfg.get("desc_act"):
                raise ValueError(f"GPTQ module {base}: desc_act checkpoints need their g_idx tensors")
            yield from emit(base, pending.pop(base))
    for base, d in list(pending.items()):
        if all(k in d for k in ("qweight", "qzeros", "scales")) and not quant_cfg.get("desc_act"):
            yield from emit(base, d)        # no g_idx in the checkpoint: groups are consecutive (desc_act = false)
        else:
            raise ValueError(f"incomplete GPTQ module {base}: has {sorted(d)}")


def quantize_rows(w, group_size=128):
    """TEST / tooling helper (round-to-nearest, asymmetric 4-bit - NOT the GPTQ solver): float [N, K] -> AutoGPTQ tensors."""
    N, K = w.shape
    assert K % group_size == 0 and K % 8 == 0 and N % 8 == 0
    wt = w.float().t().contiguous().view(K // group_size, group_size, N)          # [G, g, N]
    lo, hi = wt.amin(1), wt.amax(1)
    scale = ((hi - lo) / 15.0).clamp_min(1e-8).to(torch.float16)
    zero = torch.clamp(torch.round(-lo / scale.float()), 0, 15).to(torch.int32)    # [G, N]
    q = torch.clamp(torch.round(wt / scale.float()[:, None, :]) + zero[:, None, :], 0, 15).to(torch.int32).view(K, N)
    shifts = torch.arange(0, 32, 4, dtype=torch.int64)
    qweight = (q.view(K // 8, 8, N).to(torch.int64) << shifts[None, :, None]).sum(1)
    z0 = (zero - 1) & 15                                                             # stored minus one ("gptq" format)
    qzeros = (z0.view(-1, N // 8, 8).to(torch.int64) << shifts[None, None, :]).sum(2)
    to_i32 = lambda v: torch.from_numpy((v.numpy() & 0xFFFFFFFF).astype(np.uint32).view(np.int32).copy())
    g_idx = (torch.arange(K) // group_size).to(torch.int32)
    return {"qweight": to_i32(qweight), "qzeros": to_i32(qzeros), "scales": scale, "g_idx": g_idx}
