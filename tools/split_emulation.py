#!/usr/bin/env python
"""Round-6 gate (i): FULL-DEPTH numerics of candidate operand splits for the prefill projections, emulated in torch float64 on the GPU.

    python tools/split_emulation.py --out gpurun_out/r6_split_emulation.json

The headline prompt (ChatTS-14B, 8 x 256 -> 798 tokens, synthetic weights, seed 0) is embedded by the HIP path (TS encoder + merge), then
the 48 decoder layers run here in torch - one residual stream PER ARM, all float64 except that every GEMM operand is first rounded to
float32 (what the kernels hold) and then split the way the arm says:

  ref          a32 . W                        (float64 accumulate: the float32-reference class)
  bf16x2       (bf16 hi + bf16 lo) . W        today's parity mode
  f16          f16(a) . W                     one f16 pass
  f16+mx8      f16(a) . W  +  mx-e4m3(a - f16(a)) . mx-e4m3(W)       the 1.5-pass split (block scale e8m0 per 32 K-values, both operands)
  f16+row8     the same with ONE power-of-two scale per row instead of block scales
  bf16+mx8     bf16 hi . W + mx-e4m3(lo) . mx-e4m3(W)
  f16+mx8/w16  f16(a) . W  +  mx-e4m3(lo) . W  (lo quantised, weights exact: isolates the W8 error)

Attention, norms, RoPE, SwiGLU, residuals are float64 in every arm (those kernels do not change).  Reported: norm-wise relative error of
the first-token logits against `ref`, max|d| / max|logit|, the greedy token, and the per-GEMM error of layer 0.
NOT a product path and not the oracle: a design probe.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="chatts-14b")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--out", default="gpurun_out/r6_split_emulation.json")
    args = ap.parse_args()

    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM

    over = {} if args.layers is None else {"num_hidden_layers": args.layers}
    cfg = cfgmod.preset(args.model, **over)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256, "uniform")
    enc = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    ids = enc["input_ids"][0].tolist()
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=2048, max_prefill_tokens=1024)
    ser = enc["timeseries"].cuda()
    ps = cfg.ts["patch_size"]
    counts = [(int(v) + ps - 1) // ps for v in proc.last_lengths]
    mm_rows = model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
    full = model.expand_input_ids(list(ids), counts)
    emb = model.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm_rows).clone()
    T, H = emb.shape
    dev = emb.device
    f64 = torch.float64

    # ---- quantisers (exact restatements in float32 arithmetic) ---------------------------------------------------------------
    def round_e4m3(y):
        """RNE to OCP e4m3fn of |y| <= 448 (normal exponents -6 .. 8, 3 mantissa bits, subnormal step 2^-9)"""
        a = y.abs()
        _, ex = torch.frexp(a)                       # a = m 2^ex, m in [0.5, 1)
        e = (ex - 1).clamp(min=-6)
        step = torch.exp2((e - 3).to(torch.float32))
        q = torch.round(a / step) * step
        return torch.sign(y) * q.clamp(max=448.0)

    def block_scale(amax):
        """smallest power of two s with amax / s <= 448 (no saturation); 1 for an all-zero block"""
        m, ex = torch.frexp(amax / 448.0)
        E = ex - (m == 0.5).to(ex.dtype)
        E = torch.where(amax > 0, E, torch.zeros_like(E)).clamp(min=-127, max=127)
        return torch.exp2(E.to(torch.float32))

    def mx8(x, block=32):
        R, K = x.shape
        xb = x.view(R, K // block, block)
        s = block_scale(xb.abs().amax(-1, keepdim=True))
        return (round_e4m3(xb / s) * s).view(R, K)

    def row8(x):
        s = block_scale(x.abs().amax(-1, keepdim=True))
        return round_e4m3(x / s) * s

    def mm(a, w64):
        return a.to(f64) @ w64.t()

    arms = ["ref", "bf16x2", "f16", "f16+mx8", "f16+row8", "bf16+mx8", "f16+mx8/w16"]

    def lin(arm, a64, w64, w8mx, w8row):
        a32 = a64.to(torch.float32)
        if arm == "ref":
            return mm(a32, w64)
        if arm == "bf16x2":
            hi = a32.to(torch.bfloat16).to(torch.float32)
            lo = (a32 - hi).to(torch.bfloat16).to(torch.float32)
            return mm(hi, w64) + mm(lo, w64)
        if arm == "f16":
            return mm(a32.to(torch.float16).to(torch.float32), w64)
        if arm.startswith("f16+"):
            hi = a32.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)
            lo = a32 - hi
            if arm == "f16+mx8":
                return mm(hi, w64) + mm(mx8(lo), w8mx)
            if arm == "f16+row8":
                return mm(hi, w64) + mm(row8(lo), w8row)
            return mm(hi, w64) + mm(mx8(lo), w64)
        if arm == "bf16+mx8":
            hi = a32.to(torch.bfloat16).to(torch.float32)
            return mm(hi, w64) + mm(mx8(a32 - hi), w8mx)
        raise ValueError(arm)

    # ---- decoder pieces in float64 ---------------------------------------------------------------------------------------------
    d, nq, nkv, I = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    eps = cfg.rms_norm_eps
    pos = torch.arange(T, device=dev, dtype=f64)
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, device=dev, dtype=f64) / d))
    ang = pos[:, None] * inv[None, :]
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
    causal = torch.ones(T, T, device=dev, dtype=torch.bool).tril()

    def rms(x, w):
        return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w

    def rope(x):                                                     # [T, heads, d]
        x1, x2 = x[..., :d // 2], x[..., d // 2:]
        return x * cos[:, None, :] + torch.cat([-x2, x1], -1) * sin[:, None, :]

    def attention(q, k, v):
        q, k, v = q.view(T, nq, d), k.view(T, nkv, d), v.view(T, nkv, d)
        if cfg.qk_norm:
            raise NotImplementedError
        q, k = rope(q), rope(k)
        g = nq // nkv
        k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
        s = torch.einsum("thd,shd->hts", q, k) / (d ** 0.5)
        s = s.masked_fill(~causal[None], float("-inf"))
        p = torch.softmax(s, -1)
        return torch.einsum("hts,shd->thd", p, v).reshape(T, nq * d)

    x = {a: emb.to(f64).clone() for a in arms}
    layer0 = {}
    L = cfg.num_hidden_layers
    w16_exact = True
    for l in range(L):
        lw = model.layers[l]
        W = {k: lw[k].to(torch.float32) for k in ("qkv", "o", "gate_up", "down")}
        W64 = {k: v.to(f64) for k, v in W.items()}
        W8mx = {k: mx8(v).to(f64) for k, v in W.items()}
        W8row = {k: row8(v).to(f64) for k, v in W.items()}
        if l == 0:
            w16_exact = all(bool((v.to(torch.float16).to(torch.float32) == v).all()) for v in W.values())
        bias = lw["qkv_bias"].to(f64) if "qkv_bias" in lw else None
        n1, n2 = lw["input_norm"].to(f64), lw["post_norm"].to(f64)
        for a in arms:
            xa = x[a]
            h = rms(xa, n1)
            qkv = lin(a, h, W64["qkv"], W8mx["qkv"], W8row["qkv"])
            if l == 0:
                ref = lin("ref", h, W64["qkv"], None, None)
                layer0[a] = {"qkv": float((qkv - ref).norm() / ref.norm())}
            if bias is not None:
                qkv = qkv + bias
            att = attention(qkv[:, :nq * d], qkv[:, nq * d:(nq + nkv) * d], qkv[:, (nq + nkv) * d:])
            xa = xa + lin(a, att, W64["o"], W8mx["o"], W8row["o"])
            h = rms(xa, n2)
            gu = lin(a, h, W64["gate_up"], W8mx["gate_up"], W8row["gate_up"]).view(T, I // 16, 2, 16)
            act = (torch.nn.functional.silu(gu[:, :, 0]) * gu[:, :, 1]).reshape(T, I)
            dn = lin(a, act, W64["down"], W8mx["down"], W8row["down"])
            if l == 0:
                ref = lin("ref", act, W64["down"], None, None)
                layer0[a]["down"] = float((dn - ref).norm() / ref.norm())
                layer0[a]["act_absmax"] = float(act.abs().max())
            x[a] = xa + dn
        del W, W64, W8mx, W8row
        if (l + 1) % 8 == 0 or l + 1 == L:
            r = x["ref"]
            print(f"layer {l + 1:2d}: " + "  ".join(f"{a} {float((x[a] - r).norm() / r.norm()):.2e}" for a in arms[1:]), flush=True)

    fn = model._tensors["final_norm"].to(f64)
    lm = model._tensors["lm_head"].to(f64)
    logits = {a: rms(x[a][-1:], fn) @ lm.t() for a in arms}
    ref = logits["ref"][0]
    top2 = torch.topk(ref, 2).values
    res = {"what": "full-depth emulation of operand splits for the prefill projections (tools/split_emulation.py); logits of the first generated token",
           "model": args.model, "layers": L, "prompt_tokens": T, "w_bf16_exact_in_f16_layer0": w16_exact,
           "ref_top2_margin": float(top2[0] - top2[1]), "ref_token": int(ref.argmax()), "arms": {}}
    for a in arms[1:]:
        dlt = logits[a][0] - ref
        res["arms"][a] = {"logits_rel_err": float(dlt.norm() / ref.norm()), "max_abs_over_max_logit": float(dlt.abs().max() / ref.abs().max()),
                          "token": int(logits[a][0].argmax()), "resid_rel_err": float((x[a] - x["ref"]).norm() / x["ref"].norm()),
                          "layer0_per_gemm_rel_err": layer0[a]}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
