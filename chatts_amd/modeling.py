"""ChatTSForCausalLM - the MI355X-native model behind the reference's two call surfaces.

Mirrors (NetManAIOps/ChatTS):
  HF surface      AutoModelForCausalLM.from_pretrained(...); model.generate(**inputs, max_new_tokens=...)
                  README.md:88-103, demo/demo_hf.ipynb cells 3-5, chatts/utils/inference_tsmllm_deepspeed.py:89-105
                  (returned ids begin with the UN-expanded input_ids; model.config.ts['patch_size'] readable)
  vLLM plugin     Qwen2TSForCausalLM / Qwen3TSForCausalLM hooks, chatts/vllm/chatts_vllm.py:
                  get_multimodal_embeddings :538-562, get_input_embeddings :564-574, forward :576-599,
                  compute_logits :601-610, load_weights :612-625, packed_modules_mapping :454-464
All arithmetic runs in libchatts_amd.so (hand-written HIP, gfx950); torch only owns device memory,
streams and the RCCL process group.  There is no CPU/PyTorch fallback.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, synth
from .config import ChatTSConfig, preset
from .tp import Comm, LocalComm, ShardPlan
from .ts_encoder import TimeSeriesEmbedding

packed_modules_mapping = {"qkv_proj": ["q_proj", "k_proj", "v_proj"], "gate_up_proj": ["gate_proj", "up_proj"]}


def interleave_gate_up(gate, up):
    """[I,H],[I,H] -> [2I,H] with gate/up alternating in blocks of 16 rows (CHATTS_EPI_SWIGLU layout)."""
    I, H = gate.shape
    out = torch.empty((2 * I, H), dtype=gate.dtype, device=gate.device)
    v = out.view(I // 16, 2, 16, H)
    v[:, 0] = gate.view(I // 16, 16, H)
    v[:, 1] = up.view(I // 16, 16, H)
    return out


def quantize_int8_rows(w):
    """[N,K] bf16 -> (q uint8 view of int8 [N,K], scale f32 [N] (powers of two), deq bf16 [N,K]).

    w ~= scale[n] * q[n,k], q in [-127, 127], scale a power of two: the dequantised value has <= 7 significant bits and is EXACTLY
    representable in bf16 - the same lossless-re-encoding argument as quantize_fp8_rows, on a uniform grid (the weight-only 8-bit
    format in the role of the HF demo's `load_in_8bit`, demo/demo_hf.ipynb:78-80)."""
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 127.0)))
    scale = torch.where(amax / scale > 127.0, scale * 2.0, scale)          # guard log2 rounding: never overflow int8
    q = torch.clamp(torch.round(wf / scale[:, None]), -127, 127).to(torch.int8)
    deq = (q.float() * scale[:, None]).to(torch.bfloat16)
    return q.view(torch.uint8).contiguous(), scale.contiguous(), deq.contiguous()


def quantize_fp8_rows(w):
    """[N,K] bf16 -> (q uint8 view of float8_e4m3fn [N,K], scale f32 [N] (powers of two), deq bf16 [N,K]).

    w ~= scale[n] * float(q[n,k]) with scale a power of two, so the dequantised value has <= 4 significant bits and is
    EXACTLY representable in bf16: `deq` replaces the bf16 weight and both copies describe the same matrix."""
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    scale = torch.where(amax / scale > 448.0, scale * 2.0, scale)          # guard log2 rounding: never overflow e4m3
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[:, None]).to(torch.bfloat16)
    return q.view(torch.uint8).contiguous(), scale.contiguous(), deq.contiguous()


def quantize_int4_rows(w, group_size=128):
    """[N,K] bf16 -> (codes uint8 [N,K] in 0..15, scale f32 [N,G] (fp16-representable, like GPTQ's), zero f32 [N,G], deq bf16 [N,K]).
    Round-to-nearest asymmetric 4-bit per group of `group_size` weights of a row (NOT the GPTQ solver: that runs offline and its
    checkpoints are loaded as they are, chatts_amd/gptq.py).  deq = bf16_rne(scale * (code - zero)) REPLACES the bf16 weight, so
    every kernel and the oracle see one matrix; the int4 decode GEMV rebuilds exactly these values from the codes."""
    N, K = w.shape
    assert K % group_size == 0 and group_size % 16 == 0
    wf = w.float().view(N, K // group_size, group_size)
    lo, hi = wf.amin(2), wf.amax(2)
    scale = ((hi - lo) / 15.0).clamp_min(1e-8).to(torch.float16).float()
    zero = torch.clamp(torch.round(-lo / scale), 0, 15)
    q = torch.clamp(torch.round(wf / scale[:, :, None]) + zero[:, :, None], 0, 15)
    deq = (q * scale[:, :, None] - (scale * zero)[:, :, None]).to(torch.bfloat16).view(N, K)
    return q.to(torch.uint8).view(N, K).contiguous(), scale.contiguous(), zero.contiguous(), deq.contiguous()


def pack_int4(codes, scale, zero):
    """codes uint8 [N,K] -> (row-major nibbles uint8 [N,K/2]: byte j = code 2j | code 2j+1 << 4;  (scale, scale*zero) f32 [N,G,2])."""
    packed = (codes[:, 0::2] | (codes[:, 1::2] << 4)).contiguous()
    return packed, torch.stack([scale, scale * zero], dim=-1).contiguous()


def rope_tables(cfg, max_pos, device):
    """float32 cos/sin [max_pos, d/2], computed like Qwen2RotaryEmbedding (inv_freq = theta^(-2i/d), f32)."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    fr = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return fr.cos().contiguous().to(device), fr.sin().contiguous().to(device)


class ChatTSForCausalLM:
    packed_modules_mapping = packed_modules_mapping

    def __init__(self, config, device="cuda", comm=None, max_ctx=2048, max_prefill_tokens=2048, use_graph=True,
                 max_batch=1, weight_format="bf16", use_p2p=True, enable_prefix_caching=True, kv_block_size=None,
                 kv_pool_blocks=None, precision=None, prefill_tiled_weights=None):
        if not torch.cuda.is_available():
            raise RuntimeError("chatts_amd needs a ROCm GPU: there is no CPU fallback for the model")
        self.lib = _lib.load()
        self.config = config
        # precision: None / "bf16x2" = the parity-grade default (float32 activations carried as bf16 hi + lo planes, two MFMA passes
        # in the prefill GEMMs: logits ~5e-5 of the float32 oracle).  "bf16" = SPEED mode: the prefill GEMMs
        # multiply the bf16-rounded activations only - half the matrix work, logits ~1e-2, what a bf16 HF / vLLM run computes.
        # PROCESS-WIDE (the library's GEMM_PRECISION option, chatts_set_option): every model of this process follows the last setting.
        # "fp8" = SPEED mode on the CDNA4 fp8 matrix pipe (needs weight_format="fp8", BASELINE.json config 5): prefill chunks and the
        # TS encoder quantise their activations per row to e4m3 and multiply fp8 x fp8 (v_mfma_scale_f32_16x16x128_f8f6f4: 4x less
        # matrix time than bf16x2), logits ~1e-2 from the default; decode steps are unchanged.  Per model, not process-wide.
        # "f16q" = PARITY GRADE like the default, on other matrix pipes: prefill chunks of >= 96 rows multiply an f16 high part
        # on the f16 MFMA and an e4m3 residual x e4m3 weights on the CDNA4 block-scaled fp8 MFMA (1.5 instead of 2 pass-equivalents per
        # product; logits ~1.3e-4 of the float32 oracle against ~5e-5).  Measured SLOWER than the default (the kernel is paced by its
        # operand feed, which the split does not shrink) and costs 3 more bytes per weight: opt-in, never the default.  Per model.
        if precision not in (None, "bf16x2", "bf16", "fp8", "f16q"):
            raise ValueError("precision must be None / 'bf16x2' or 'f16q' (parity grade), 'bf16' or 'fp8' (speed modes)")
        if precision == "fp8" and weight_format != "fp8":
            raise ValueError("precision='fp8' multiplies the fp8 weight copies: it needs weight_format='fp8'")
        if precision is not None:
            _lib.set_option("GEMM_PRECISION", 1 if precision == "bf16" else None)
        self.precision = precision or "bf16x2"
        self.device = torch.device(device)
        self.comm = comm or LocalComm()
        self.plan = ShardPlan(config, self.comm.rank, self.comm.world)
        self.max_ctx = int(max_ctx)
        # block-paged KV cache (vLLM's block_size / num_gpu_blocks): kv_block_size = positions per block (power of two >= 64),
        # None = one contiguous cache per slot.  kv_pool_blocks < max_batch * max_ctx / block oversubscribes the slots: requests
        # reserve blocks for prompt + max_new_tokens at admission and wait while the pool cannot cover them (chatts_amd/kv_blocks.py)
        self.kv_block_size = int(kv_block_size or 0)
        if self.kv_block_size:
            if self.kv_block_size < 64 or self.kv_block_size & (self.kv_block_size - 1):
                raise ValueError(f"kv_block_size={kv_block_size} must be a power of two >= 64")
            self.max_ctx = -(-self.max_ctx // self.kv_block_size) * self.kv_block_size      # whole blocks
        self._kv_pool_blocks = None if kv_pool_blocks is None else int(kv_pool_blocks)
        self._kv = None                          # kv_blocks.BlockPool once the buffers exist
        self._kv_replaced = {}                   # slot -> logical blocks swapped for private ones since its last prefix decision
        self._kv_dynamic = False
        self._cur_slot = 0
        if weight_format not in ("bf16", "fp8", "int8", "int4"):
            raise ValueError("weight_format must be 'bf16', 'fp8', 'int8' or 'int4'")
        # "fp8": decode GEMVs stream an e4m3 copy (BASELINE.json config 5); "int4": they stream 4-bit codes + group scales (what a
        # GPTQ-Int4 checkpoint brings along; a bf16 checkpoint is quantised round-to-nearest at load)
        self.weight_format = weight_format
        self.int4_group = 128
        self._gptq_codes = {}                    # HF module name -> (codes, scale, zero) of a GPTQ checkpoint, consumed by _load_with
        self.max_batch = int(max_batch)          # KV-cache slots for batched decode (continuous batching)
        self.t_max = int(max(min(max_prefill_tokens, self.max_ctx), self.max_batch))
        # a second copy of the four projection matrices of every layer in the prefill kernel's tiled layout (chatts_tile_bf16: its LDS-DMA
        # pieces then read consecutive memory; bit-identical results, prefill projections -4 %, profiles/r6_tiled_check.txt).  None = when
        # a prefill chunk can reach that kernel (>= 96 rows) and CHATTS_TILED_WEIGHTS != 0; costs the bf16 weight bytes once more.
        if prefill_tiled_weights is None:
            prefill_tiled_weights = self.t_max >= 96 and os.environ.get("CHATTS_TILED_WEIGHTS", "1") != "0" and self.precision != "f16q"
        self.prefill_tiled_weights = bool(prefill_tiled_weights)
        self._tiled = []                         # per layer: {projection: tiled copy}
        self.use_graph = use_graph
        self.use_p2p = use_p2p                   # TP: decode-sized exchanges through csrc/tp.hip instead of RCCL (chatts_amd/tp.py)
        self._tp = None                          # P2PExchange of this rank once attached
        # prefix reuse (vLLM's enable_prefix_caching; demo/demo_vllm.py:55 submits 100 identical prompts, multi-turn chats
        # re-send their history): per cache slot, the identity of every token whose K/V rows are resident
        self.enable_prefix_caching = bool(enable_prefix_caching)
        self._slot_idents = [[] for _ in range(max(1, int(max_batch)))]
        self.prefix_stats = {"requests": 0, "hits": 0, "tokens_reused": 0, "tokens_prefilled": 0}
        self.ts_encoder = TimeSeriesEmbedding(config.ts, device=self.device)
        self._tensors = {}            # keeps every device tensor alive (the C side borrows pointers)
        self._decoder = None
        self._graph = None
        self.layers = []
        self.loaded = set()

    # ---------------------------------------------------------------------------------------------
    # weights
    # ---------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config, seed=0, **kw):
        """Random-init model whose weights are the counter-hash of chatts_amd/synth.py (bf16-exact)."""
        if isinstance(config, str):
            config = preset(config)
        m = cls(config, **kw)
        dev, plan, cfg = m.device, m.plan, config
        specs = {s.name: s for s in synth.all_specs(cfg)}

        def block(name, row0=0, rows=None, col0=0, cols=None, f32=False):
            s = specs[name]
            rows = s.rows - row0 if rows is None else rows
            cols = s.cols - col0 if cols is None else cols
            t = torch.empty((rows, cols), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
            synth.fill_device(t, s, seed, row0=row0, col0=col0, rows=rows, cols=cols)
            return t[0] if s.is_1d else t

        m._load_with(block, tied=cfg.tie_word_embeddings)
        m.ts_encoder.load_synthetic(synth.ts_encoder_specs(cfg), seed)
        m.seed = seed
        m._finalize()
        return m

    @classmethod
    def register_for_auto_class(cls, auto_class="AutoModelForCausalLM"):
        """transformers calls this on a class resolved through config.json's auto_map (trust_remote_code); nothing to register."""

    @classmethod
    def from_pretrained(cls, path, *model_args, trust_remote_code=True, device_map=None, torch_dtype=None, dtype=None,
                        lora_adapter=None, config=None, **kw):
        """HF checkpoint directory (config.json + *.safetensors with the names of SURVEY.md section 5).
        Also the target of ``AutoModelForCausalLM.from_pretrained(path, trust_remote_code=True, device_map=..., torch_dtype=...)``
        (README.md:88) through the auto_map ChatTSConfig.save_pretrained writes: transformers' own keywords are accepted and
        ignored (`config` is re-read from config.json; activations are float32 whatever torch_dtype says).
        lora_adapter: a peft LoRA adapter directory merged into the weights while loading (chatts_amd/lora.py)."""
        import glob
        import os
        from safetensors import safe_open
        cfg = ChatTSConfig.from_pretrained(path)
        for hf_only in ("cache_dir", "force_download", "local_files_only", "token", "revision", "use_safetensors", "subfolder",
                        "low_cpu_mem_usage", "attn_implementation", "code_revision", "_from_auto", "adapter_kwargs", "use_auth_token",
                        "proxies", "resume_download", "_commit_hash", "_from_pipeline", "quantization_config", "weights_only"):
            kw.pop(hf_only, None)
        if isinstance(device_map, int):
            device_map = f"cuda:{device_map}"
        if isinstance(device_map, dict):
            device_map = "cuda"
        if device_map not in (None, "auto") and "device" not in kw:
            kw["device"] = device_map if isinstance(device_map, str) else "cuda"
        m = cls(cfg, **kw)
        handles, index = [], {}
        for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
            h = safe_open(f, framework="pt", device="cpu")
            handles.append(h)
            for k in h.keys():
                index[k] = h
        pairs = ((k, h.get_tensor(k)) for k, h in index.items())
        from . import gptq
        if gptq.is_gptq_config(cfg.extra):          # ChatTS-14B-GPTQ-Int4: unpack qweight / qzeros / scales / g_idx to bf16 weights
            pairs = gptq.dequantized_pairs(pairs, cfg.extra["quantization_config"], has_g_idx=any(k.endswith(".g_idx") for k in index))
        if lora_adapter is not None:
            from . import lora
            pairs = lora.merged(pairs, lora_adapter)
        m.load_weights(pairs)
        m._checkpoint_path, m._checkpoint_kw = path, dict(kw)
        return m

    def load_weights(self, weights):
        """chatts_vllm.py:612-625: consume (name, tensor) pairs with HF names; q/k/v and gate/up are packed,
        a missing lm_head means tied embeddings (:619-623).  Returns the set of loaded names."""
        sd = {}
        for name, t in weights:
            if name.endswith(".gptq_codes"):          # (codes, scale, zero) of a GPTQ module (chatts_amd/gptq.py)
                self._gptq_codes[name[:-len(".gptq_codes")]] = t
                continue
            if name.startswith("ts_encoder."):
                self.ts_encoder.load_tensor(name[len("ts_encoder."):], t)
                self.loaded.add(name)
            else:
                sd[name] = t
        dev = self.device

        def block(name, row0=0, rows=None, col0=0, cols=None, f32=False):
            t = sd[name]
            if t.dim() == 1:
                t = t[col0:(None if cols is None else col0 + cols)]
            else:
                t = t[row0:(None if rows is None else row0 + rows), col0:(None if cols is None else col0 + cols)]
            self.loaded.add(name)
            return t.to(dev).to(torch.float32 if f32 else torch.bfloat16).contiguous()

        self._load_with(block, tied="lm_head.weight" not in sd)
        self._finalize()
        return self.loaded

    def _load_with(self, block, tied):
        cfg, plan = self.config, self.plan
        d = cfg.head_dim
        T = self._tensors
        T["embed"] = block("model.embed_tokens.weight")
        if tied:
            T["lm_head"] = T["embed"][plan.v0:plan.v0 + plan.vocab].contiguous() if plan.world > 1 else T["embed"]
        else:
            T["lm_head"] = block("lm_head.weight", row0=plan.v0, rows=plan.vocab)
        T["final_norm"] = block("model.norm.weight", f32=True)
        for l in range(cfg.num_hidden_layers):
            p = f"model.layers.{l}."
            q = block(p + "self_attn.q_proj.weight", row0=plan.q0 * d, rows=plan.nq * d)
            k = block(p + "self_attn.k_proj.weight", row0=plan.kv0 * d, rows=plan.nkv * d)
            v = block(p + "self_attn.v_proj.weight", row0=plan.kv0 * d, rows=plan.nkv * d)
            L = {"qkv": torch.cat([q, k, v], dim=0).contiguous()}
            del q, k, v
            if cfg.attention_bias:
                L["qkv_bias"] = torch.cat([
                    block(p + "self_attn.q_proj.bias", col0=plan.q0 * d, cols=plan.nq * d, f32=True),
                    block(p + "self_attn.k_proj.bias", col0=plan.kv0 * d, cols=plan.nkv * d, f32=True),
                    block(p + "self_attn.v_proj.bias", col0=plan.kv0 * d, cols=plan.nkv * d, f32=True)]).contiguous()
            if cfg.qk_norm:
                L["q_norm"] = block(p + "self_attn.q_norm.weight", f32=True)
                L["k_norm"] = block(p + "self_attn.k_norm.weight", f32=True)
            L["o"] = block(p + "self_attn.o_proj.weight", col0=plan.q0 * d, cols=plan.nq * d)
            g = block(p + "mlp.gate_proj.weight", row0=plan.i0, rows=plan.inter)
            u = block(p + "mlp.up_proj.weight", row0=plan.i0, rows=plan.inter)
            L["gate_up"] = interleave_gate_up(g, u)
            del g, u
            L["down"] = block(p + "mlp.down_proj.weight", col0=plan.i0, cols=plan.inter)
            L["input_norm"] = block(p + "input_layernorm.weight", f32=True)
            L["post_norm"] = block(p + "post_attention_layernorm.weight", f32=True)
            self._fuse_gptq_codes(L, p)
            self.layers.append(L)

    def _fuse_gptq_codes(self, L, p):
        """GPTQ checkpoint: pack the per-module 4-bit codes the way the weights were packed (q|k|v rows, gate/up interleaved) so
        that the decode GEMV can stream them.  Only for tensor_parallel_size 1 (a K-slice of down_proj is not group aligned)."""
        G = self._gptq_codes
        names = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
                 "mlp.down_proj"]
        if self.plan.world != 1 or not all(p + n in G for n in names):
            return
        dev = self.device
        c = {n: tuple(t.to(dev) for t in G.pop(p + n)) for n in names}
        cat = lambda ns: tuple(torch.cat([c[n][i] for n in ns], dim=0).contiguous() for i in range(3))
        fused = {"qkv": cat(names[:3]), "o": c[names[3]], "down": c[names[6]],
                 "gate_up": tuple(interleave_gate_up(c[names[4]][i], c[names[5]][i]) for i in range(3))}
        for k, (q, sc, z) in fused.items():
            self.int4_group = q.shape[1] // sc.shape[1]
            L[k + "4"], L[k + "4_sz"] = pack_int4(q, sc, z)

    def _finalize(self):
        """Allocate activations / KV cache and create the C-side decoder handle (borrowing all pointers)."""
        cfg, plan, dev, lib = self.config, self.plan, self.device, self.lib
        T = self._tensors
        H, d = cfg.hidden_size, cfg.head_dim
        max_pos = max(self.max_ctx, 64)
        T["cos"], T["sin"] = rope_tables(cfg, max_pos, dev)
        if self.weight_format in ("fp8", "int8") and "lm_head8" not in T:
            # quantise every decoder projection once; the bf16 tensors are REPLACED by the (bf16-exact) dequantised
            # values so prefill (bf16 MFMA GEMM), batched decode and the 8-bit decode GEMV all see the same weights
            quant = quantize_fp8_rows if self.weight_format == "fp8" else quantize_int8_rows
            for lw in self.layers:
                for name in ("qkv", "o", "gate_up", "down"):
                    lw[name + "8"], lw[name + "8_scale"], lw[name] = quant(lw[name])
            tied = T["lm_head"].data_ptr() == T["embed"].data_ptr()
            T["lm_head8"], T["lm_head8_scale"], deq = quant(T["lm_head"])
            T["lm_head"] = deq
            if tied:
                T["embed"] = deq
        if self.weight_format == "int4" and self.layers and "qkv4" not in self.layers[0]:
            # round-to-nearest 4-bit codes of every decoder projection; the bf16 tensors are REPLACED by the dequantised values
            for lw in self.layers:
                for name in ("qkv", "o", "gate_up", "down"):
                    q, sc, z, lw[name] = quantize_int4_rows(lw[name], self.int4_group)
                    lw[name + "4"], lw[name + "4_sz"] = pack_int4(q, sc, z)
        # decode attention: one wave per 16-key tile; slots beyond the live context exit immediately
        self.n_splits = max(1, min(64, (self.max_ctx + 15) // 16))
        dc = _lib.DecoderConfig(hidden=H, n_layers=cfg.num_hidden_layers, n_q=plan.nq, n_kv=plan.nkv, head_dim=d,
                                inter=plan.inter, vocab_local=plan.vocab, vocab_offset=plan.v0,
                                rms_eps=cfg.rms_norm_eps, max_ctx=self.max_ctx, max_pos=max_pos, tp_world=plan.world,
                                embed_rows=cfg.vocab_size, embed_offset=0,       # the embedding table is replicated
                                w8_format=_lib.W8_INT8 if self.weight_format == "int8" else _lib.W8_FP8)
        ws_bytes = int(lib.chatts_decoder_workspace(C.byref(dc), self.t_max, self.n_splits))
        ws_bytes = max(ws_bytes, int(lib.chatts_linear_workspace(self.t_max, plan.vocab, H)))
        for m in range(2, self.max_batch + 1):     # batched lm_head
            ws_bytes = max(ws_bytes, int(lib.chatts_linear_workspace(m, plan.vocab, H)))
        ws_bytes = max(ws_bytes, int(lib.chatts_attn_workspace(self.max_batch, plan.nq, self.n_splits)) + 256)
        # the prefill attention's K / V planes (kv_planes_kernel: 32 KB per (kv head, 32-key tile)) for a sequence of max_ctx positions:
        # with the workspace sized for them the planes kernel is taken at EVERY chunk position, so the same prompt never changes
        # kernel (and summation order) with where a chunk starts
        ws_bytes = max(ws_bytes, ((self.max_ctx + 31) // 32) * plan.nkv * 4 * 32 * d * 2 + 256)
        MB = self.max_batch
        f32 = dict(dtype=torch.float32, device=dev)
        qkv_n = (plan.nq + 2 * plan.nkv) * d
        L = cfg.num_hidden_layers
        if self.kv_block_size:
            from .kv_blocks import BlockPool
            bps = self.max_ctx // self.kv_block_size
            nblk = MB * bps if self._kv_pool_blocks is None else self._kv_pool_blocks
            if nblk < 1 or nblk > MB * bps:
                raise ValueError(f"kv_pool_blocks={nblk} must lie in 1..{MB * bps} (max_batch x max_ctx / kv_block_size)")
            kv_shape = (L, nblk, plan.nkv, self.kv_block_size, d)             # layer-major block pools
            self._kv = BlockPool(nblk, self.kv_block_size, MB, bps)
            self._kv_dynamic = nblk < MB * bps
        else:
            kv_shape = (MB, L, plan.nkv, self.max_ctx, d)                     # slot-major: one contiguous cache per sequence
        B = {
            "kv_k": torch.zeros(kv_shape, **f32),
            "kv_v": torch.zeros(kv_shape, **f32),
            "kv_table": torch.zeros((MB, self.max_ctx // self.kv_block_size), dtype=torch.int32, device=dev) if self.kv_block_size else None,
            "x": torch.zeros((self.t_max, H), **f32), "xn": torch.zeros((self.t_max, H), **f32),
            "qkv": torch.zeros((self.t_max, qkv_n), **f32), "attn": torch.zeros((self.t_max, plan.nq * d), **f32),
            "act": torch.zeros((self.t_max, plan.inter), **f32), "delta": torch.zeros((self.t_max, H), **f32),
            "logits": torch.zeros(plan.vocab, **f32),
            # two pairs of bf16 hi / lo planes (prefill: LDS-DMA GEMM operands; pair 0 = projection input, pair 1 = SwiGLU out)
            "planes": torch.zeros((4, (self.t_max + 31) // 32 * 32, max(H, plan.nq * d, plan.inter)), dtype=torch.bfloat16, device=dev),   # (rows: whole 16-row blocks for the tiled f16q planes)
            "ws": torch.zeros(ws_bytes, dtype=torch.uint8, device=dev),
            # decode-loop state lives on the device so a captured step can be replayed
            "pos_all": torch.zeros(MB, dtype=torch.int32, device=dev), "step_all": torch.zeros(MB, dtype=torch.int32, device=dev),
            "token_all": torch.zeros(MB, dtype=torch.int64, device=dev), "token_logit_all": torch.zeros(MB, **f32),
            "out_tokens_all": torch.zeros((MB, self.max_ctx + 8), dtype=torch.int64, device=dev),
            "logits_all": torch.zeros((MB, plan.vocab), **f32) if MB > 1 else None,
            "scan": torch.zeros(self.t_max + 8, dtype=torch.int32, device=dev),
            "status": torch.zeros(1, dtype=torch.int32, device=dev),
            # tensor parallel: scratch of the token agreement (chatts_decoder_select_tokens)
            "tp_pair_logit": torch.zeros(MB, **f32) if plan.world > 1 else None,
            "tp_pair_token": torch.zeros(MB, dtype=torch.int64, device=dev) if plan.world > 1 else None,
            "logits_full": torch.zeros((MB, plan.vocab * plan.world), **f32) if plan.world > 1 else None,
        }
        # single-sequence views (slot 0): the batch-1 fast path and its hipGraph use these
        B["pos"], B["step"], B["token"], B["token_logit"] = B["pos_all"][:1], B["step_all"][:1], B["token_all"][:1], B["token_logit_all"][:1]
        B["out_tokens"] = B["out_tokens_all"][0]
        self.buf = B
        if self._kv is not None and not self._kv_dynamic:       # a full pool: every slot owns max_ctx positions for good
            for sl in range(MB):
                self._kv.reserve(sl, self.max_ctx)
                self._kv.retire(sl)
                self._push_kv_row(sl)
        self._tiled = []
        tile_ok = self.prefill_tiled_weights
        for lw in self.layers:
            tiled = {}
            if tile_ok:
                # the copies are an optimisation: never let them take the memory the KV cache / activations of a large configuration need
                need = sum(lw[n].numel() * 2 for n in ("qkv", "o", "gate_up", "down"))
                free = torch.cuda.mem_get_info(dev)[0]
                if free < need + (8 << 30):
                    import warnings
                    warnings.warn(f"prefill_tiled_weights: {free / 2**30:.1f} GiB free - layers {len(self._tiled)}.. keep the row-major weights only")
                    tile_ok = False
            if tile_ok:
                for name in ("qkv", "o", "gate_up", "down"):
                    w = lw[name]
                    rows, k = w.shape
                    if k % 64 == 0 and w.is_contiguous():
                        out = torch.empty(int(lib.chatts_tile_bf16_elems(rows, k)), dtype=torch.bfloat16, device=dev)
                        _lib.check(lib.chatts_tile_bf16(w.data_ptr(), rows, k, k, out.data_ptr(), _lib.stream_ptr()))
                        tiled[name] = out
            self._tiled.append(tiled)
        self._f16q = []
        for lw in self.layers:
            qd = {}
            if self.precision == "f16q":
                for name in ("qkv", "o", "gate_up", "down"):
                    w = lw[name]
                    rows, k = w.shape
                    if k % 128 or not w.is_contiguous():
                        raise ValueError(f"precision='f16q' needs K % 128 == 0 for every projection ({name}: K={k})")
                    w16 = torch.empty((rows, k), dtype=torch.float16, device=dev)
                    w8 = torch.empty((rows, k), dtype=torch.uint8, device=dev)
                    w8e = torch.empty((rows,), dtype=torch.uint8, device=dev)
                    _lib.check(lib.chatts_weights_f16q(w.data_ptr(), rows, k, k, w16.data_ptr(), w8.data_ptr(), w8e.data_ptr(), k, _lib.stream_ptr()))
                    # ... stored in the kernel's own block order: an LDS-DMA piece is 1 KB of consecutive memory
                    w16t = torch.empty(int(lib.chatts_tile_bf16_elems(rows, k)), dtype=torch.float16, device=dev)
                    w8t = torch.empty(int(lib.chatts_tile_e4m3_bytes(rows, k)), dtype=torch.uint8, device=dev)
                    _lib.check(lib.chatts_tile_bf16(w16.data_ptr(), rows, k, k, w16t.data_ptr(), _lib.stream_ptr()))
                    _lib.check(lib.chatts_tile_e4m3(w8.data_ptr(), rows, k, k, w8t.data_ptr(), _lib.stream_ptr()))
                    qd[name] = (w16t, w8t, w8e)
            self._f16q.append(qd)
        arr = (_lib.LayerWeights * L)()
        for i, lw in enumerate(self.layers):
            tl = self._tiled[i]
            fq = {f"{n}{suf}": _lib.ptr(t3[j]) for n, t3 in self._f16q[i].items() for j, suf in enumerate(("16", "_q8", "_q8e"))}
            arr[i] = _lib.LayerWeights(**fq, qkv_t=_lib.ptr(tl.get("qkv")), o_t=_lib.ptr(tl.get("o")), gate_up_t=_lib.ptr(tl.get("gate_up")),
                                       down_t=_lib.ptr(tl.get("down")),
                                       input_norm=_lib.ptr(lw["input_norm"]), qkv=_lib.ptr(lw["qkv"]),
                                       qkv_bias=_lib.ptr(lw.get("qkv_bias")), q_norm=_lib.ptr(lw.get("q_norm")),
                                       k_norm=_lib.ptr(lw.get("k_norm")), o=_lib.ptr(lw["o"]),
                                       post_norm=_lib.ptr(lw["post_norm"]), gate_up=_lib.ptr(lw["gate_up"]),
                                       down=_lib.ptr(lw["down"]),
                                       qkv8=_lib.ptr(lw.get("qkv8")), qkv8_scale=_lib.ptr(lw.get("qkv8_scale")),
                                       o8=_lib.ptr(lw.get("o8")), o8_scale=_lib.ptr(lw.get("o8_scale")),
                                       gate_up8=_lib.ptr(lw.get("gate_up8")), gate_up8_scale=_lib.ptr(lw.get("gate_up8_scale")),
                                       down8=_lib.ptr(lw.get("down8")), down8_scale=_lib.ptr(lw.get("down8_scale")),
                                       qkv4=_lib.ptr(lw.get("qkv4")), qkv4_sz=_lib.ptr(lw.get("qkv4_sz")),
                                       o4=_lib.ptr(lw.get("o4")), o4_sz=_lib.ptr(lw.get("o4_sz")),
                                       gate_up4=_lib.ptr(lw.get("gate_up4")), gate_up4_sz=_lib.ptr(lw.get("gate_up4_sz")),
                                       down4=_lib.ptr(lw.get("down4")), down4_sz=_lib.ptr(lw.get("down4_sz")),
                                       w4_group=self.int4_group)
        self._layer_arr = arr
        dw = _lib.DecoderWeights(layers=arr, final_norm=_lib.ptr(T["final_norm"]), lm_head=_lib.ptr(T["lm_head"]),
                                 lm_head8=_lib.ptr(T.get("lm_head8")), lm_head8_scale=_lib.ptr(T.get("lm_head8_scale")),
                                 embed=_lib.ptr(T["embed"]), cos_tab=_lib.ptr(T["cos"]), sin_tab=_lib.ptr(T["sin"]))
        db = _lib.DecoderBuffers(kv_k=_lib.ptr(B["kv_k"]), kv_v=_lib.ptr(B["kv_v"]), x=_lib.ptr(B["x"]),
                                 xn=_lib.ptr(B["xn"]), qkv=_lib.ptr(B["qkv"]), attn=_lib.ptr(B["attn"]),
                                 act=_lib.ptr(B["act"]), delta=_lib.ptr(B["delta"]), logits=_lib.ptr(B["logits"]),
                                 workspace=_lib.ptr(B["ws"]), workspace_bytes=ws_bytes, t_max=self.t_max,
                                 max_batch=self.max_batch, planes_hi=_lib.ptr(B["planes"][0]),
                                 planes_lo=_lib.ptr(B["planes"][1]), planes2_hi=_lib.ptr(B["planes"][2]),
                                 planes2_lo=_lib.ptr(B["planes"][3]), tp_pair_logit=_lib.ptr(B["tp_pair_logit"]),
                                 tp_pair_token=_lib.ptr(B["tp_pair_token"]), logits_full=_lib.ptr(B["logits_full"]),
                                 kv_block_table=_lib.ptr(B["kv_table"]), kv_block_size=self.kv_block_size,
                                 kv_table_stride=(self.max_ctx // self.kv_block_size) if self.kv_block_size else 0,
                                 kv_pool_blocks=self._kv.n_blocks if self._kv is not None else 0)
        h = lib.chatts_decoder_create(C.byref(dc), C.byref(dw), C.byref(db))
        if not h:
            raise _lib.ChattsError(-1, lib.chatts_last_error().decode())
        self._decoder = C.c_void_p(h)
        self._graph = None
        self._graph_batched = None
        if self.precision == "fp8":              # speed mode: prefill chunks and the TS encoder on the fp8 matrix pipe
            _lib.check(lib.chatts_decoder_set_prefill_fp8(self._decoder, 1))
            self.ts_encoder.set_precision("fp8")
        if self.precision == "f16q":             # parity-grade prefill on the f16 + fp8 matrix pipes (the TS encoder keeps bf16x2)
            _lib.check(lib.chatts_decoder_set_prefill_f16q(self._decoder, 1))
        if plan.world > 1 and self.use_p2p and getattr(self.comm, "dist", None) is not None and self._tp is None:
            from .tp import P2PExchange
            ex, err = None, None
            try:
                ex = P2PExchange.create(self.comm, self.exchange_elems(), self.exchange_bulk_elems())
            except Exception as e:               # e.g. no IPC mapping between these two devices
                err = e
            # every rank must take the same path: agree before anyone attaches
            oks = [None] * plan.world
            self.comm.dist.all_gather_object(oks, err is None, group=self.comm.group)
            if all(oks):
                self.attach_exchange(ex)
            else:
                if ex is not None:
                    ex.close()
                if self.max_batch > 1:
                    raise RuntimeError(f"the peer-to-peer exchange could not be set up on ranks {[r for r, ok in enumerate(oks) if not ok]} "
                                       f"({err}); batched decode under tensor parallelism needs it")
                import warnings
                warnings.warn(f"peer-to-peer exchange unavailable on ranks {[r for r, ok in enumerate(oks) if not ok]} ({err}): "
                              "decode-sized sums go through RCCL from the host (slower, no hipGraph)")
                self.use_p2p = False

    def exchange_bulk_elems(self):
        """float32 elements of the largest prefill-sized sum: one chunk of partial [t_max, H] rows (chatts_allreduce_bulk)"""
        return self.t_max * self.config.hidden_size

    def exchange_elems(self):
        """float32 elements per rank the largest in-step collective moves: [max_batch, H] partial sums, or the logits gather."""
        return max(self.max_batch * self.config.hidden_size, self.max_batch * self.plan.vocab, 2 * self.max_batch)

    def attach_exchange(self, exchange):
        """Bind this rank's P2PExchange (chatts_amd/tp.py): the whole TP decode step then runs behind one C call / one hipGraph."""
        _lib.check(self.lib.chatts_decoder_set_tp(self._decoder, exchange.handle if exchange is not None else None))
        self._tp = exchange
        self._graph = None
        self._graph_batched = None

    def __del__(self):
        try:
            if self._decoder:
                self.lib.chatts_decoder_destroy(self._decoder)
            if self._tp is not None:
                self._tp.close()
        except Exception:
            pass

    def weight_bytes_local(self):
        """Bytes one batch-1 decode step streams on this rank (fp8 copies replace their bf16 twins when present)."""
        n = 0
        for lw in self.layers:
            for k, t in lw.items():
                if k + "8" in lw or k + "4" in lw:
                    continue                       # the fp8 / 4-bit copy is the one the decode GEMV reads
                n += t.numel() * t.element_size()
        T = self._tensors
        head = (T["lm_head8"].numel() + T["lm_head8_scale"].numel() * 4) if "lm_head8" in T else T["lm_head"].numel() * 2
        return n + head + T["final_norm"].numel() * 4

    def tiled_weight_bytes_local(self):
        """Bytes of the prefill kernel's tiled weight copies on this rank (resident beside the tensors weight_bytes_local() counts)."""
        return sum(t.numel() * t.element_size() for tl in self._tiled for t in tl.values())

    def f16q_weight_bytes_local(self):
        """Bytes of the f16q weight copies (precision='f16q' only)"""
        return sum(t.numel() * t.element_size() for qd in self._f16q for t3 in qd.values() for t in t3)

    # ---------------------------------------------------------------------------------------------
    # vLLM-plugin-shaped hooks (same names/order as chatts_vllm.py:538-610)
    # ---------------------------------------------------------------------------------------------
    def get_multimodal_embeddings(self, timeseries=None, valid_lengths=None, **kw):
        """-> list of [patch_cnt_i, H] tensors (empty [0,H] for zero-length series), or None."""
        if timeseries is None:
            return None
        ts = self._parse_and_validate_ts_input(timeseries)
        feats, patch_cnt = self.ts_encoder(ts, valid_lengths=valid_lengths)
        if valid_lengths is not None:
            ps = self.ts_encoder.patch_size
            counts = [(int(v) + ps - 1) // ps for v in valid_lengths]
        else:
            counts = patch_cnt.tolist()
        out, s = [], 0
        for c in counts:
            out.append(feats[s:s + c])
            s += c
        return out

    def _parse_and_validate_ts_input(self, timeseries):
        """chatts_vllm.py:493-536: accept the padded tensor, or a list of (ts_tokens, encoded [1,2L,1]) items."""
        if isinstance(timeseries, torch.Tensor):
            return timeseries.to(self.device, dtype=torch.float32)
        if isinstance(timeseries, (list, tuple)):
            encs = []
            for item in timeseries:
                if isinstance(item, (list, tuple)) and len(item) == 2 and not np.isscalar(item[1]):
                    encs.append(np.asarray(item[1], dtype=np.float32).reshape(1, -1, 1))
                else:
                    encs.append(np.asarray(item, dtype=np.float32).reshape(1, -1, 1))
            lmax = max(e.shape[1] for e in encs)
            out = np.zeros((len(encs), lmax, 1), dtype=np.float32)
            for i, e in enumerate(encs):
                out[i, :e.shape[1]] = e[0]
            return torch.from_numpy(out).to(self.device)
        raise ValueError(f"Incorrect type of ts input features. Got type: {type(timeseries)}")

    def get_input_embeddings(self, input_ids, multimodal_embeddings=None, out=None):
        """embed_tokens(ids) with rows at ts_token_start_index overwritten in order by the TS rows (:564-574).
        input_ids: 1-D (already expanded) ids, host list / CPU tensor / device tensor."""
        ids_host = None
        if not isinstance(input_ids, torch.Tensor):
            input_ids = torch.tensor(list(input_ids), dtype=torch.int64)
        if not input_ids.is_cuda:
            ids_host = input_ids.to(torch.int64).contiguous()
            ids_dev = ids_host.to(self.device, non_blocking=True)
        else:
            ids_dev = input_ids.to(torch.int64).contiguous()
        t = ids_dev.numel()
        H = self.config.hidden_size
        rows, n_rows = None, 0
        if multimodal_embeddings is not None:
            rows = (multimodal_embeddings if isinstance(multimodal_embeddings, torch.Tensor)
                    else (torch.cat(list(multimodal_embeddings), dim=0) if len(multimodal_embeddings) else
                          torch.empty((0, H), dtype=torch.float32, device=self.device)))
            rows = rows.contiguous()
            n_rows = rows.shape[0]
        if out is None:
            out = torch.empty((t, H), dtype=torch.float32, device=self.device)
        scan = self.buf["scan"] if t + 1 <= self.buf["scan"].numel() else torch.empty(t + 1, dtype=torch.int32, device=self.device)
        if ids_host is None:
            self.buf["status"].zero_()
        _lib.check(self.lib.chatts_embed_merge(
            _lib.ptr(ids_dev), _lib.ptr(ids_host) if ids_host is not None else None, t, _lib.ptr(self._tensors["embed"]),
            self.config.vocab_size, H, _lib.ptr(rows) if n_rows else None, n_rows, self.config.ts_token_start_index,
            _lib.ptr(out), _lib.ptr(scan), _lib.ptr(self.buf["status"]), _lib.stream_ptr()))
        if ids_host is None and int(self.buf["status"].item()) & 1:
            # device-resident ids: the count check ran on the GPU (one D2H sync, as vLLM's merge_multimodal_embeddings does)
            raise ValueError(f"Attempted to assign {n_rows} multimodal tokens to a different number of <ts> placeholders")
        return out

    def forward(self, input_ids=None, positions=None, intermediate_tensors=None, inputs_embeds=None, **kw):
        """Prefill `inputs_embeds` [T,H] (or embed input_ids [+ timeseries kw]) at positions pos0..pos0+T-1;
        returns the residual-stream hidden states [T,H] BEFORE the final norm (compute_logits applies it)."""
        if inputs_embeds is None:
            mm = self.get_multimodal_embeddings(**kw)
            inputs_embeds = self.get_input_embeddings(input_ids, mm)
        pos0 = 0 if positions is None else int(positions[0])
        T = inputs_embeds.shape[0]
        if T > self.t_max:
            raise ValueError(f"{T} tokens exceed max_prefill_tokens={self.t_max}; use prefill() for chunking")
        self.buf["x"][:T].copy_(inputs_embeds)
        self._run_layers(T, pos0)
        return self.buf["x"][:T]

    def compute_logits(self, hidden_states=None, sampling_metadata=None, row=None):
        """final RMSNorm + lm_head on ONE row of the residual stream (default: the last) -> [V_local] float32."""
        if hidden_states is not None and hidden_states.data_ptr() != self.buf["x"].data_ptr():
            self.buf["x"][:hidden_states.shape[0]].copy_(hidden_states)
        if row is None:
            row = (hidden_states.shape[0] - 1) if hidden_states is not None else 0
        _lib.check(self.lib.chatts_decoder_logits(self._decoder, int(row), _lib.stream_ptr()))
        return self.buf["logits"]

    # ---------------------------------------------------------------------------------------------
    # engine
    # ---------------------------------------------------------------------------------------------
    def _tp_bulk(self, T):
        """tensor parallel: the attached exchange can sum [T, H] partials itself (chatts_allreduce_bulk) -> one C call per chunk"""
        import os
        if os.environ.get("CHATTS_TP_BULK", "1") == "0":          # A/B and fault isolation: the host-driven sums (RCCL / gloo) instead
            return False
        return self._tp is not None and self._tp.bulk_elems >= T * self.config.hidden_size

    def _run_layers(self, T, pos0, pos_dev=None, n_splits=1, last_only=False):
        lib, st = self.lib, _lib.stream_ptr()
        if pos_dev is None and (self.plan.world == 1 or self._tp_bulk(T)):
            if last_only:         # only the next token + the KV cache are wanted: the final layer runs for the last row only
                _lib.check(lib.chatts_decoder_prefill_last(self._decoder, T, pos0, st))
            else:
                _lib.check(lib.chatts_decoder_prefill(self._decoder, T, pos0, st))
            return
        H = self.config.hidden_size
        delta = self.buf["delta"][:T]
        tp = self.plan.world > 1
        p2p = tp and self._tp is not None and T <= 16 and T * H <= self._tp.max_elems
        pending = 0                      # TP: the all-reduced delta of the previous part is added by the next C call
        for l in range(self.config.num_hidden_layers):
            for part in (0, 1):
                _lib.check(lib.chatts_decoder_layer_part_add(self._decoder, pending, l, part, T, pos0, _lib.ptr(pos_dev),
                                                             n_splits, st))
                if tp and p2p:
                    self._tp.all_reduce(delta, out=self.buf["x"][:T], resid=self.buf["x"][:T])      # x += sum of the partials
                elif tp:
                    self.comm.all_reduce(delta)         # RCCL for prefill-sized messages
                    pending = 1
        if pending:
            _lib.check(lib.chatts_residual_add(_lib.ptr(self.buf["x"]), _lib.ptr(delta), T * H, st))

    # ---- block-paged KV cache: host bookkeeping (chatts_amd/kv_blocks.py) ------------------------------------------------
    def _push_kv_row(self, slot):
        row = self._kv.rows[slot]
        if row:                                  # stream-ordered: kernels enqueued after this see the new row
            self.buf["kv_table"][slot, :len(row)] = torch.tensor(row, dtype=torch.int32)

    def _kv_evicted(self, victim):
        self._slot_idents[victim] = []           # nothing is resident there any more (its stale table row is never read: parked)

    def reserve_kv(self, slot, n_tokens, idents=None, protect=(), write_from=0):
        """Make cache slot `slot` able to hold n_tokens positions (no-op with a contiguous cache) and mark it active.  idents = the
        request's token identities: the slot whose resident prefix will be reused is protected from eviction (like `protect`), and
        every block the request is going to WRITE - everything past the reused prefix; `write_from` positions for direct callers -
        is made private first if another slot shares it.  Raises kv_blocks.KvPoolExhausted (nothing changed) when even evicting
        every finished sequence does not free enough blocks."""
        if self._kv is None:
            return
        if n_tokens > self.max_ctx:
            raise ValueError(f"{n_tokens} positions exceed max_ctx={self.max_ctx}")
        keep = set(protect)
        first = int(write_from) // self._kv.block_size
        if idents is not None and self.enable_prefix_caching:
            n, src = self._match_prefix(slot, idents)
            n = min(n, len(idents) - 1)
            if src >= 0 and n >= 16:
                keep.add(src)
                first = n // self._kv.block_size          # own slot: rows below stay; other slot: blocks below get adopted
        order = sorted(range(self.max_batch), key=lambda sl: len(self._slot_idents[sl]))       # least resident first
        changed = self._kv.reserve(slot, n_tokens, protect=keep, evict_order=order, on_evict=self._kv_evicted, private_from=first)
        self._kv_replaced[slot] = self._kv_replaced.get(slot, set()) | self._kv.last_replaced
        if changed:
            self._push_kv_row(slot)

    def kv_fits(self, token_counts):
        """could requests needing these many positions (prompt + new tokens each) be admitted into free slots now?"""
        return self._kv is None or self._kv.fits(token_counts)

    def request_tokens(self, ids, series=None, lengths=None):
        """expanded prompt length of a request (host arithmetic when the series lengths are known)"""
        ps = self.config.ts["patch_size"]
        if series is not None and series.shape[0] > 0 and lengths is None:
            lengths = self.ts_encoder.get_patch_cnt(series.to(self.device, dtype=torch.float32))[0].tolist()
        counts = [(int(v) + ps - 1) // ps for v in (lengths or [])]
        return len(self.expand_input_ids(list(ids), counts))

    def kv_stats(self):
        if self._kv is None:
            return None
        return {"block_size": self._kv.block_size, "blocks": self._kv.n_blocks, "free": len(self._kv.free),
                "active_slots": len(self._kv.active), "evictions": self._kv.evictions, "dynamic": self._kv_dynamic,
                "shared_blocks": self._kv.shared_blocks()}

    def reset(self):
        self.buf["pos"].zero_()
        self.buf["step"].zero_()

    def prefill(self, inputs_embeds, pos0=0, for_next_token=False):
        """Chunked prefill of [T,H] embeddings starting at cache position pos0; leaves the last chunk in x and returns the
        number of rows whose LAST one yields the next token's logits (chatts_decoder_logits(row = returned - 1)).
        for_next_token: the hidden states are not wanted, only the next token and the cache - the final layer of the last chunk
        then runs attention / o_proj / MLP for the last row only (chatts_decoder_prefill_last) and that row is row 0: returns 1."""
        T = inputs_embeds.shape[0]
        if pos0 + T > self.max_ctx:
            raise ValueError(f"sequence of {pos0 + T} tokens exceeds max_ctx={self.max_ctx}")
        if self._kv is not None and (self._kv.capacity_tokens(self._cur_slot) < pos0 + T or self._kv.shared_blocks()):
            self.reserve_kv(self._cur_slot, pos0 + T, write_from=pos0)      # direct callers; generate_* reserve prompt + new tokens up front
        fast_last = for_next_token and (self.plan.world == 1 or self._tp_bulk(min(T, self.t_max)))
        done, last = 0, 0
        while done < T:
            n = min(self.t_max, T - done)
            self.buf["x"][:n].copy_(inputs_embeds[done:done + n])
            final = done + n >= T
            self._run_layers(n, pos0 + done, last_only=fast_last and final)
            done += n
            last = 1 if (fast_last and final) else n
        return last

    def set_sampling(self, temperature=0.0, top_k=0, top_p=1.0, seed=0):
        """Token selection of every following step: greedy (temperature 0 / None, the default) or temperature / top-k /
        top-p sampling on the GPU (chatts_sample_batched).  Same seed + same prompt => same tokens."""
        if not temperature:
            args = None
        else:
            if temperature < 0 or not (0.0 < top_p <= 1.0):
                raise ValueError(f"temperature={temperature} must be >= 0 and top_p={top_p} in (0, 1]")
            args = _lib.SamplingArgs(temperature=float(temperature), top_k=int(top_k or 0), top_p=float(top_p),
                                     seed=int(seed) & 0xFFFFFFFF, n_kept=None, kept_mass=None)
        key = None if args is None else (args.temperature, args.top_k, args.top_p, args.seed)
        if key == getattr(self, "_sampling_key", None):
            return
        _lib.check(self.lib.chatts_decoder_set_sampling(self._decoder, None if args is None else C.byref(args)))
        self._sampling, self._sampling_key = args, key
        self._graph = None               # the captured steps have the old selection kernel baked in
        self._graph_batched = None

    def _sampling_arrays(self):
        if "samp_temp" not in self.buf:
            dev, n = self.device, max(1, self.max_batch)
            self.buf["samp_temp"] = torch.zeros(n, dtype=torch.float32, device=dev)     # 0 = the slot decodes greedily
            self.buf["samp_topk"] = torch.zeros(n, dtype=torch.int32, device=dev)
            self.buf["samp_topp"] = torch.ones(n, dtype=torch.float32, device=dev)
            self.buf["samp_seed"] = torch.zeros(n, dtype=torch.int32, device=dev)       # (bit pattern of the uint32 seed)
        return self.buf

    def _rows_args(self, slot):
        B = self._sampling_arrays()
        return _lib.SamplingArgs(temperature=1.0, top_k=0, top_p=1.0, seed=0, n_kept=None, kept_mass=None,
                                 temperature_rows=B["samp_temp"][slot:].data_ptr(), top_k_rows=B["samp_topk"][slot:].data_ptr(),
                                 top_p_rows=B["samp_topp"][slot:].data_ptr(), seed_rows=B["samp_seed"][slot:].data_ptr())

    def set_sampling_rows(self):
        """Per-slot sampling (ChattsSamplingArgs.*_rows): every cache slot draws with its own (temperature, top_k, top_p, seed), read on
        the device at run time from four small arrays - requests with different settings share one captured step, and
        set_slot_sampling() changes a slot without re-capturing.  Slots start greedy (temperature 0)."""
        if getattr(self, "_sampling_key", None) == "rows":
            return
        args = self._rows_args(0)
        _lib.check(self.lib.chatts_decoder_set_sampling(self._decoder, C.byref(args)))
        self._sampling, self._sampling_key = args, "rows"
        self._graph = None
        self._graph_batched = None

    def set_slot_sampling(self, slot, temperature=0.0, top_k=0, top_p=1.0, seed=0):
        """Sampling settings of the sequence in cache slot `slot`: used by every step in per-slot mode (set_sampling_rows), inert in
        the other modes.  A plain device write - no re-capture."""
        temperature = 0.0 if temperature is None else float(temperature)
        if temperature < 0 or not (0.0 < top_p <= 1.0):
            raise ValueError(f"temperature={temperature} must be >= 0 and top_p={top_p} in (0, 1]")
        B = self._sampling_arrays()
        B["samp_temp"][slot] = temperature
        B["samp_topk"][slot] = max(int(top_k or 0), 0)
        B["samp_topp"][slot] = float(top_p)
        s32 = int(seed) & 0xFFFFFFFF
        B["samp_seed"][slot] = s32 - (1 << 32) if s32 >= (1 << 31) else s32

    def _select_token(self):
        """logits of this rank -> next token in B['token'] (appended to out_tokens, step bumped) on every rank: greedy or the
        configured sampler, agreed across TP ranks by chatts_decoder_select_tokens ((max, idx) pairs / gathered logits)."""
        lib, st, B = self.lib, _lib.stream_ptr(), self.buf
        if self.plan.world == 1 or self._tp is not None:
            _lib.check(lib.chatts_decoder_select_tokens(self._decoder, _lib.ptr(B["logits"]), 1, self.plan.vocab, _lib.ptr(B["token"]),
                                                        _lib.ptr(B["token_logit"]), _lib.ptr(B["out_tokens"]), 0,
                                                        _lib.ptr(B["step"]), None, 0, None, st))
            return
        # TP without the peer-to-peer exchange (use_p2p=False): host-driven RCCL gathers
        sa = getattr(self, "_sampling", None)
        if sa is None:      # greedy: one (logit, index) pair per rank instead of the [V / W] logits
            _lib.check(lib.chatts_argmax(_lib.ptr(B["logits"]), self.plan.vocab, self.plan.v0, _lib.ptr(B["token"]),
                                         _lib.ptr(B["token_logit"]), None, None, None, st))
            B["token"].copy_(self.comm.argmax_pair(B["token_logit"], B["token"]))
            B["out_tokens"].index_copy_(0, B["step"].to(torch.int64), B["token"])
            B["step"] += 1
            return
        full = self.comm.all_gather_cat(B["logits"])     # every rank draws from the full logits with the same seed
        _lib.check(lib.chatts_sample_batched(_lib.ptr(full), 1, full.numel(), full.numel(), 0, C.byref(sa),
                                             _lib.ptr(B["token"]), _lib.ptr(B["token_logit"]), _lib.ptr(B["out_tokens"]), 0,
                                             _lib.ptr(B["step"]), None, 0, st))

    def _first_token(self, last_rows):
        """logits of the last prompt row -> first generated token -> next input embedding in x[0]."""
        _lib.check(self.lib.chatts_decoder_logits(self._decoder, last_rows - 1, _lib.stream_ptr()))
        self._select_token()
        self._load_token_embedding()

    def _load_token_embedding(self):
        B = self.buf
        _lib.check(self.lib.chatts_embed_token(_lib.ptr(B["token"]), _lib.ptr(self._tensors["embed"]), 0,
                                               self.config.vocab_size, self.config.hidden_size, _lib.ptr(B["x"]),
                                               _lib.stream_ptr()))

    def _decode_step_eager(self):
        lib, st, B = self.lib, _lib.stream_ptr(), self.buf
        if self.plan.world == 1 or self._tp is not None:         # one C call enqueues the whole (TP) step
            _lib.check(lib.chatts_decoder_decode_step(self._decoder, _lib.ptr(B["pos"]), _lib.ptr(B["step"]),
                                                      _lib.ptr(B["token"]), _lib.ptr(B["token_logit"]),
                                                      _lib.ptr(B["out_tokens"]), self.n_splits, st))
            return
        self._run_layers(1, 0, pos_dev=B["pos"], n_splits=self.n_splits)
        _lib.check(lib.chatts_decoder_logits(self._decoder, 0, st))
        B["pos"] += 1
        self._select_token()
        self._load_token_embedding()

    def decode_step(self):
        """One token.  A hipGraph of the whole step (captured on first use) is replayed - under tensor parallelism too, when the
        peer-to-peer exchange is attached (its collectives are ordinary kernels with device-resident epochs)."""
        if self.graph_capturable():
            if self._graph is None:
                self._capture()
            self._graph.replay()
        else:
            self._decode_step_eager()

    def graph_capturable(self):
        return bool(self.use_graph and (self.plan.world == 1 or self._tp is not None))

    def _capture(self, warm=True):
        """Capture one decode step.  warm: run one eager step first (populates every lazy host-side cache; under tensor
        parallelism all ranks do, so it really exchanges) and restore the state; capturing itself executes nothing."""
        B = self.buf
        saved = {k: B[k].clone() for k in ("pos", "step", "token", "token_logit", "x", "out_tokens")}
        torch.cuda.synchronize()
        if warm:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._decode_step_eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for k, v in saved.items():
                B[k].copy_(v)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._decode_step_eager()
        torch.cuda.synchronize()
        for k, v in saved.items():          # capture does not execute, but keep the state pristine anyway
            B[k].copy_(v)
        self._graph = g

    # ---------------------------------------------------------------------------------------------
    # batched decode / continuous batching (TP = 1)
    # ---------------------------------------------------------------------------------------------
    def select_sequence(self, slot):
        _lib.check(self.lib.chatts_decoder_select_sequence(self._decoder, int(slot)))
        self._cur_slot = int(slot)

    def _n_splits_batched(self):
        """16-key slots per sequence of the batched decode attention: with many sequences in flight the grid is full anyway and
        fewer, longer slots save partials traffic (round 2, B = 16 at ctx 0.8k: 64 -> 8 slots, 7.10 -> 6.91 ms per step; round 3, with
        the next tile prefetched under the current one and ctx 1.2k: 16 slots, profiles/r3_cfg5_nsplits_sweep.txt)."""
        import os
        forced = int(os.environ.get("CHATTS_BATCH_NSPLITS", "0") or 0)          # tuning only
        if forced > 0:
            return max(1, min(self.n_splits, forced))
        if self.max_batch <= 4:
            return self.n_splits
        return max(1, min(self.n_splits, max(16, 256 // self.max_batch)))      # (B = 16 at ctx 1.2k: 8 -> 16 slots, 7.83 -> 7.01 ms per step)

    def _batched_step_eager(self):
        B = self.buf
        _lib.check(self.lib.chatts_decoder_decode_step_batched(
            self._decoder, self.max_batch, _lib.ptr(B["pos_all"]), _lib.ptr(B["step_all"]), _lib.ptr(B["token_all"]),
            _lib.ptr(B["token_logit_all"]), _lib.ptr(B["out_tokens_all"]), B["out_tokens_all"].shape[1],
            _lib.ptr(B["logits_all"]), self._n_splits_batched(), _lib.stream_ptr()))

    def batched_step(self):
        """One greedy token for EVERY cache slot (idle slots compute harmlessly; their position saturates)."""
        if self.plan.world != 1 and self._tp is None:
            raise NotImplementedError("batched decode under tensor parallelism needs the peer-to-peer exchange (use_p2p=True)")
        if not self.use_graph:
            return self._batched_step_eager()
        if self._graph_batched is None:
            B = self.buf
            keys = ("pos_all", "step_all", "token_all", "token_logit_all", "out_tokens_all")
            saved = {k: B[k].clone() for k in keys}
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._batched_step_eager()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for k in keys:
                B[k].copy_(saved[k])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._batched_step_eager()
            torch.cuda.synchronize()
            for k in keys:
                B[k].copy_(saved[k])
            self._graph_batched = g
        self._graph_batched.replay()

    # ---- prefix reuse ---------------------------------------------------------------------------------
    @staticmethod
    def _token_idents(full, series, lengths, counts, ts0):
        """Identity of every token of an EXPANDED prompt: the id for text tokens; for a <ts> placeholder the digest of its
        series' (value, mask) row + the patch index - two placeholder rows hold the same embedding iff those agree."""
        import hashlib
        digests = []
        if series is not None and len(counts):
            host = series.detach().cpu().numpy() if hasattr(series, "detach") else np.asarray(series)
            for i, L in enumerate(lengths):
                digests.append(hashlib.blake2b(np.ascontiguousarray(host[i].reshape(-1)[:2 * int(L)]).tobytes(), digest_size=8).digest())
        out, k, j = [], -1, 0
        owner = []                                   # series index of every placeholder, in order
        for i, c in enumerate(counts):
            owner += [(i, p) for p in range(int(c))]
        for t in full:
            if t == ts0:
                i, pidx = owner[j]
                j += 1
                out.append((digests[i], pidx))
            else:
                out.append(t)
        return out

    def _request_idents(self, ids, series, lengths):
        """token identities of a request before it is admitted (slot choice): host-side only, no device work"""
        if not self.enable_prefix_caching:
            return None
        ps = self.config.ts["patch_size"]
        if series is not None and series.shape[0] > 0 and lengths is None:
            return None
        counts = [(int(v) + ps - 1) // ps for v in (lengths or [])]
        try:
            full = self.expand_input_ids(list(ids), counts)
        except ValueError:
            return None
        return self._token_idents(full, series, lengths, counts, self.config.ts_token_start_index)

    def _match_prefix(self, slot, idents, exclude=()):
        """(n, src): the longest prefix of `idents` resident in some cache slot (ties prefer `slot` itself: zero copy)."""
        best, src = 0, -1
        for s, have in enumerate(self._slot_idents):
            if s in exclude:
                continue
            n = 0
            for a, b in zip(idents, have):
                if a != b:
                    break
                n += 1
            if n > best or (n == best and s == slot and n > 0):
                best, src = n, s
        return best, src

    def _reuse_prefix(self, slot, idents, T, exclude=()):
        """Longest prefix of `idents` already resident in some slot's cache -> copy those K/V rows into `slot` (nothing to copy
        when it is the slot itself) and return how many leading tokens need no prefill (at most T - 1: the last prompt
        position is always recomputed, it yields the logits)."""
        self.prefix_stats["requests"] += 1
        replaced = self._kv_replaced.pop(slot, set())      # (paged) blocks reserve_kv swapped for private, EMPTY ones
        if not self.enable_prefix_caching:
            return 0
        best, src = self._match_prefix(slot, idents, exclude)
        n = min(best, T - 1)
        if n < 16:                                   # not worth a copy + a ragged prefill start
            return 0
        if self._kv is not None:
            # paged: nothing is copied.  Another slot's prefix is SHARED block by block (reference counted), so the reuse ends at a
            # block boundary; the slot's own prefix ends where reserve_kv had to swap a shared block for a private (empty) one.
            bs = self._kv.block_size
            if src != slot:
                nshare = n // bs
                if nshare == 0 or len(self._kv.rows[slot]) < nshare or len(self._kv.rows[src]) < nshare:
                    return 0                     # less than a block in common (or not reserved by the caller): recompute
                if self._kv.adopt(slot, src, nshare):
                    self._push_kv_row(slot)
                n = nshare * bs
            else:
                lost = [i for i in replaced if i * bs < n]
                if lost:
                    n = min(lost) * bs
                if n < 16:
                    return 0
        elif src != slot:
            B = self.buf
            B["kv_k"][slot, :, :, :n].copy_(B["kv_k"][src, :, :, :n])
            B["kv_v"][slot, :, :, :n].copy_(B["kv_v"][src, :, :, :n])
        self.prefix_stats["hits"] += 1
        self.prefix_stats["tokens_reused"] += n
        return n

    def note_generated(self, slot, tokens):
        """After a request finished: its generated tokens (all but the last, which was never fed back) also have K/V rows in
        the slot - a follow-up turn that re-sends prompt + answer reuses them."""
        if self.enable_prefix_caching and tokens:
            self._slot_idents[slot] = self._slot_idents[slot] + [int(t) for t in tokens[:-1]]
        if self._kv is not None:
            self._kv.retire(slot)                # blocks stay resident for prefix reuse; evictable from now on

    def pick_slot(self, free_slots, idents=None):
        """Which free cache slot a new request should take: the one whose resident prefix matches best (zero copy), else the
        one whose resident tokens are least worth keeping (fewest)."""
        if not self.enable_prefix_caching or idents is None:
            return free_slots[0]
        def match(s):
            n = 0
            for a, b in zip(idents, self._slot_idents[s]):
                if a != b:
                    break
                n += 1
            return n
        best = max(free_slots, key=match)
        if match(best) >= 16:
            return best
        return min(free_slots, key=lambda s: len(self._slot_idents[s]))

    def admit_begin(self, slot, ids, series, lengths, max_new_tokens=1):
        """First half of an admission into cache slot `slot`: TS encoder, prompt expansion, embeddings, KV reservation and prefix
        reuse - everything but the prefill itself.  -> the state admit_step() consumes."""
        cfg = self.config
        ps = cfg.ts["patch_size"]
        mm, counts = None, []
        if series is not None and series.shape[0] > 0:
            series = series.to(self.device, dtype=torch.float32)
            if lengths is None:
                lengths = self.ts_encoder.get_patch_cnt(series)[0].tolist()
            counts = [(int(v) + ps - 1) // ps for v in lengths]
            mm = self.get_multimodal_embeddings(timeseries=series, valid_lengths=lengths)
        full = self.expand_input_ids(list(ids), counts)
        T = len(full)
        if T + max_new_tokens > self.max_ctx:       # same bound as generate_one: a sequence must never outgrow its cache
            raise ValueError(f"prompt ({T}) + max_new_tokens ({max_new_tokens}) exceeds max_ctx={self.max_ctx}")
        emb = self.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
        idents = self._token_idents(full, series, lengths, counts, cfg.ts_token_start_index)
        self.reserve_kv(slot, T + max_new_tokens, idents)
        n0 = self._reuse_prefix(slot, idents, T)
        self._slot_idents[slot] = []                # not a prefix source for others until all of its rows exist
        self.prefix_stats["tokens_prefilled"] += T - n0
        return {"slot": slot, "emb": emb, "T": T, "done": n0, "idents": idents}

    def admit_step(self, st, max_rows=None):
        """Prefill the next <= max_rows prompt rows of an admission begun with admit_begin (all that is left when None).  Between two
        calls the engine may run decode steps of the OTHER slots (vLLM's chunked prefill: a long prompt does not stall the running
        sequences for its whole prefill); this slot stays parked until its last row is in, then its first token is produced.
        -> True when the admission is complete."""
        slot, T, a = st["slot"], st["T"], st["done"]
        b = T if max_rows is None else min(T, a + max(int(max_rows), 1))
        final = b >= T
        self.select_sequence(slot)
        last = self.prefill(st["emb"][a:b], a, for_next_token=final)
        st["done"] = b
        if final:
            self._first_token_into_slot(slot, last - 1, T)
            self._slot_idents[slot] = st["idents"]
            st["emb"] = None
        self.select_sequence(0)
        return final

    def _admit(self, slot, ids, series, lengths, max_new_tokens=1):
        """Prefill one request into cache slot `slot` and produce its first token (out_tokens_all[slot, 0])."""
        st = self.admit_begin(slot, ids, series, lengths, max_new_tokens)
        self.admit_step(st)
        return st["T"]

    def _first_token_into_slot(self, slot, row, T):
        """logits of x[row] -> first token of the sequence in cache slot `slot` (out_tokens_all[slot, 0]); position / step set."""
        B, st = self.buf, _lib.stream_ptr()
        _lib.check(self.lib.chatts_decoder_logits(self._decoder, row, st))
        B["pos_all"][slot] = T
        B["step_all"][slot] = 0
        sa = getattr(self, "_sampling", None)
        sa1 = None
        if getattr(self, "_sampling_key", None) == "rows":
            sa1 = self._rows_args(slot)          # this single-row call sees row 0: hand it the slot's entries
        elif sa is not None:   # the batched steps draw with (seed, slot, step); this single-row call would see slot 0: fold the slot into the seed
            sa1 = _lib.SamplingArgs(temperature=sa.temperature, top_k=sa.top_k, top_p=sa.top_p,
                                    seed=(sa.seed ^ (0x51ED27 * (slot + 1))) & 0xFFFFFFFF, n_kept=None, kept_mass=None)
        _lib.check(self.lib.chatts_decoder_select_tokens(
            self._decoder, _lib.ptr(B["logits"]), 1, self.plan.vocab, B["token_all"][slot:].data_ptr(),
            B["token_logit_all"][slot:].data_ptr(), B["out_tokens_all"][slot].data_ptr(), B["out_tokens_all"].shape[1],
            B["step_all"][slot:].data_ptr(), None, 0, None if sa1 is None else C.byref(sa1), st))

    def plan_pack(self, candidates, free_slots):
        """Which of the waiting requests should be prefilled TOGETHER (chatts_decoder_prefill_packed).  candidates: list of
        (ids, series, lengths, max_new_tokens) in arrival order.  Greedy: take requests while their not-yet-cached rows fit the
        prefill buffers and slots are free; a request that shares most of its prompt with one already in the pack is left for the
        next round (it will then reuse that request's K/V rows instead of computing them again).  -> indices into candidates."""
        if self.plan.world != 1 or len(free_slots) < 2 or len(candidates) < 2:
            return []
        ps = self.config.ts["patch_size"]
        pack, rows, pack_idents = [], 0, []
        for i, (ids, series, lengths, max_new) in enumerate(candidates):
            if len(pack) >= len(free_slots):
                break
            idents = self._request_idents(ids, series, lengths)
            if idents is None:          # lengths unknown / prefix caching off: size by the expanded prompt alone
                counts = [(int(v) + ps - 1) // ps for v in (lengths or [])]
                try:
                    T = len(self.expand_input_ids(list(ids), counts))
                except ValueError:
                    continue
                need, idents = T, None
            else:
                T = len(idents)
                n, _ = self._match_prefix(-1, idents)
                n = min(n, T - 1)
                need = T - (n if n >= 16 else 0)
                dup = False
                for other in pack_idents:
                    if other is None:
                        continue
                    k = 0
                    for a, b in zip(idents, other):
                        if a != b:
                            break
                        k += 1
                    if k >= max(16, T // 2):
                        dup = True
                        break
                if dup:
                    continue
            if T + max_new > self.max_ctx or rows + need > self.t_max:
                continue
            pack.append(i)
            pack_idents.append(idents)
            rows += need
        return pack if len(pack) >= 2 else []

    def _admit_packed(self, items):
        """Prefill several requests in ONE packed pass.  items: list of (slot, ids, series, lengths, max_new_tokens) with distinct
        free slots whose uncached rows fit max_prefill_tokens together (plan_pack).  Returns the prompt lengths."""
        cfg, B = self.config, self.buf
        ps = cfg.ts["patch_size"]
        segs, embs, Ts, row, late = [], [], [], 0, []
        if self._kv_dynamic:        # oversubscribed block pool: reserve for EVERY member before anything changes, or for none
            from .kv_blocks import KvPoolExhausted
            newly, touched = [], []
            try:
                for slot, ids, series, lengths, max_new in items:
                    was_active = slot in self._kv.active
                    touched.append(slot)
                    self.reserve_kv(slot, self.request_tokens(ids, series, lengths) + max_new, self._request_idents(ids, series, lengths))
                    if not was_active:
                        newly.append(slot)
            except KvPoolExhausted:
                # roll back: the members reserved so far may already have had shared prefix blocks swapped for EMPTY private ones, so
                # what their slots advertise as resident is no longer true - forget it (a later request must not adopt those blocks)
                for sl in touched:
                    self._slot_idents[sl] = []
                    self._kv_replaced.pop(sl, None)
                for sl in newly:
                    self._kv.retire(sl)
                raise
        for slot, ids, series, lengths, max_new in items:
            mm, counts = None, []
            if series is not None and series.shape[0] > 0:
                series = series.to(self.device, dtype=torch.float32)
                if lengths is None:
                    lengths = self.ts_encoder.get_patch_cnt(series)[0].tolist()
                counts = [(int(v) + ps - 1) // ps for v in lengths]
                mm = self.get_multimodal_embeddings(timeseries=series, valid_lengths=lengths)
            full = self.expand_input_ids(list(ids), counts)
            T = len(full)
            if T + max_new > self.max_ctx:
                raise ValueError(f"prompt ({T}) + max_new_tokens ({max_new}) exceeds max_ctx={self.max_ctx}")
            emb = self.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
            idents = self._token_idents(full, series, lengths, counts, cfg.ts_token_start_index)
            # reuse from ANY slot, the member's own and other members' included: every K/V row copy is enqueued here, i.e. before
            # the packed pass overwrites anything; a slot stops being a source once its member has been processed (idents cleared)
            self.reserve_kv(slot, T + max_new, idents)       # (dynamic pool: already covered by the pre-pass above)
            n0 = self._reuse_prefix(slot, idents, T)
            self._slot_idents[slot] = []
            if row + T - n0 > self.t_max:                    # an earlier member's copy took away the prefix this one counted on
                late.append((slot, ids, series, lengths, max_new))
                self.prefix_stats["requests"] -= 1
                self.prefix_stats["hits"] -= int(n0 > 0)
                self.prefix_stats["tokens_reused"] -= n0
                Ts.append(None)
                continue
            self.prefix_stats["tokens_prefilled"] += T - n0
            segs.append((row, T - n0, n0, slot, idents))
            embs.append(emb[n0:])
            Ts.append(T)
            row += T - n0
        if segs:
            B["x"][:row].copy_(torch.cat(embs, dim=0))
            arr = (_lib.PrefillSegment * len(segs))(*[_lib.PrefillSegment(row0=r0, t=t, pos0=p0, slot=sl) for r0, t, p0, sl, _ in segs])
            _lib.check(self.lib.chatts_decoder_prefill_packed(self._decoder, arr, len(segs), _lib.stream_ptr()))
            self.prefix_stats["packed_prefills"] = self.prefix_stats.get("packed_prefills", 0) + 1
            for r0, t, p0, sl, idents in segs:
                self._slot_idents[sl] = idents
                self._first_token_into_slot(sl, r0 + t - 1, p0 + t)
        for it in late:                                      # (rare) did not fit after all: the ordinary path
            Ts[Ts.index(None)] = self._admit(*it)
        return Ts

    @torch.no_grad()
    def generate_batch(self, requests, max_new_tokens=64, eos_token_id=None, sync_every=8):
        """Continuous batching over `max_batch` cache slots.  requests: list of (ids, series [n,2Lmax,1] | None,
        lengths | None).  A finished sequence frees its slot for the next waiting request at the next sync point.
        Returns the list of generated token lists, in request order."""
        if self.max_batch < 2:
            return [self.generate_one(i, s, l, max_new_tokens, eos_token_id) for (i, s, l) in requests]
        eos = set(eos_token_id if isinstance(eos_token_id, (list, tuple, set)) else
                  ([] if eos_token_id is None else [eos_token_id]))
        B = self.buf
        results = [None] * len(requests)
        waiting = list(range(len(requests)))[::-1]
        slots = [None] * self.max_batch              # request index per slot
        produced = [0] * self.max_batch
        B["pos_all"].fill_(-1); B["step_all"].zero_(); B["token_all"].zero_()
        steps_since_sync = 0

        def harvest(final=False):
            toks_all = B["out_tokens_all"].cpu()
            if self._tp is not None and self._tp.status():
                raise RuntimeError("tensor-parallel exchange timed out (a peer rank did not reach the collective): tokens are invalid")
            for s, r in enumerate(slots):
                if r is None:
                    continue
                toks = toks_all[s, :produced[s]].tolist()
                cut = next((i + 1 for i, t in enumerate(toks) if t in eos), None) if eos else None
                if cut is not None or produced[s] >= max_new_tokens or final:
                    results[r] = toks[:cut] if cut is not None else toks[:max_new_tokens]
                    slots[s] = None
                    B["pos_all"][s] = -1         # parked: the batched step neither attends nor writes this slot's cache
                    self.note_generated(s, results[r])

        while waiting or any(r is not None for r in slots):
            while waiting and any(v is None for v in slots):
                free = [i for i, v in enumerate(slots) if v is None]
                cands = waiting[::-1][:len(free)]            # the next requests in arrival order
                pack = self.plan_pack([requests[r] + (max_new_tokens,) for r in cands], free)
                if pack and self._kv_dynamic and not self.kv_fits(
                        [self.request_tokens(*requests[cands[j]]) + max_new_tokens for j in pack]):
                    pack = []                                # the block pool cannot take them all at once: one at a time
                if pack:                                     # several short prompts: one packed prefill pass
                    items, members = [], []
                    for j in pack:
                        r = cands[j]
                        ids, series, lengths = requests[r]
                        s = self.pick_slot(free, self._request_idents(ids, series, lengths))
                        free.remove(s)
                        items.append((s, ids, series, lengths, max_new_tokens))
                        members.append((s, r))
                    try:
                        self._admit_packed(items)
                    except RuntimeError as e:                # the pool could not cover the pack after all (a protected prefix
                        if type(e).__name__ != "KvPoolExhausted" or all(v is None for v in slots):     # source): the touched slots forgot their resident prefixes
                            raise
                        break
                    for s, r in members:
                        slots[s], produced[s] = r, 1
                        waiting.remove(r)
                    continue
                ids, series, lengths = requests[waiting[-1]]
                if self._kv_dynamic and not self.kv_fits([self.request_tokens(ids, series, lengths) + max_new_tokens]):
                    if all(v is None for v in slots):        # nothing left to finish and free blocks: it can never fit
                        from .kv_blocks import KvPoolExhausted
                        raise KvPoolExhausted(f"request {waiting[-1]} needs more KV blocks than the pool holds ({self.kv_stats()})")
                    break                                    # wait until a running sequence finishes and its blocks become evictable
                s = self.pick_slot(free, self._request_idents(ids, series, lengths))
                try:
                    self._admit(s, ids, series, lengths, max_new_tokens)
                except RuntimeError as e:                    # (the reservation is _admit's first change: nothing to undo)
                    if type(e).__name__ != "KvPoolExhausted" or all(v is None for v in slots):
                        raise
                    break
                r = waiting.pop()
                slots[s], produced[s] = r, 1
            if all(r is None or produced[s] >= max_new_tokens for s, r in enumerate(slots)):
                harvest()
                continue
            self.batched_step()
            for s, r in enumerate(slots):
                if r is not None and produced[s] < max_new_tokens:
                    produced[s] += 1
            steps_since_sync += 1
            if steps_since_sync >= sync_every or all(r is None or produced[s] >= max_new_tokens for s, r in enumerate(slots)):
                harvest()
                steps_since_sync = 0
        return results

    # ---------------------------------------------------------------------------------------------
    # HF surface
    # ---------------------------------------------------------------------------------------------
    def expand_input_ids(self, ids, patch_counts):
        """Replace each [<ts>, <ts/>] pair by patch_cnt <ts> placeholders (chatts_vllm.py:402-415, 441)."""
        ts0 = self.config.ts_token_start_index
        out, k, i, n = [], 0, 0, len(ids)
        while i < n:
            if ids[i] == ts0 and i + 1 < n and ids[i + 1] == ts0 + 1:
                if k >= len(patch_counts):
                    raise ValueError("more <ts><ts/> pairs than time series")
                out.extend([ts0] * int(patch_counts[k]))
                k += 1
                i += 2
            else:
                out.append(ids[i])
                i += 1
        if k != len(patch_counts):
            raise ValueError(f"{len(patch_counts)} time series but {k} <ts><ts/> pairs in the prompt")
        return out

    @torch.no_grad()
    def generate_one(self, ids, series=None, lengths=None, max_new_tokens=64, eos_token_id=None, sync_every=16,
                     return_logits=False):
        """ids: un-expanded prompt ids (host list); series: [n, 2*Lmax, 1] tensor of this prompt's series."""
        self._prefill_request(ids, series, lengths, max_new_tokens)
        logits0 = self.buf["logits"].clone() if return_logits else None
        eos = set(eos_token_id if isinstance(eos_token_id, (list, tuple, set)) else
                  ([] if eos_token_id is None else [eos_token_id]))
        produced = 1
        toks = None
        while produced < max_new_tokens:
            self.decode_step()
            produced += 1
            if eos and (produced % sync_every == 0):
                toks = self.buf["out_tokens"][:produced].tolist()
                if any(t in eos for t in toks):
                    break
        toks = self.buf["out_tokens"][:produced].tolist()
        if self._tp is not None and self._tp.status():
            raise RuntimeError("tensor-parallel exchange timed out (a peer rank did not reach the collective): tokens are invalid")
        if eos:
            for i, t in enumerate(toks):
                if t in eos:
                    toks = toks[:i + 1]
                    break
        self.note_generated(0, toks)
        return (toks, logits0) if return_logits else toks

    @torch.no_grad()
    def generate_stream(self, ids, series=None, lengths=None, max_new_tokens=64, eos_token_id=None, chunk=1):
        """Generator over the new tokens of ONE request: yields a list of token ids every `chunk` decode steps (chunk=1: token by
        token, what an SSE stream / HF TextStreamer consumes; each yield costs one device->host read of the tokens so far).
        Stops after an EOS token (which is yielded) or max_new_tokens.  Same kernels and tokens as generate_one."""
        eos = set(eos_token_id if isinstance(eos_token_id, (list, tuple, set)) else ([] if eos_token_id is None else [eos_token_id]))
        self._prefill_request(ids, series, lengths, max_new_tokens)
        produced, sent = 1, 0
        while True:
            if produced - sent >= chunk or produced >= max_new_tokens:
                toks = self.buf["out_tokens"][sent:produced].tolist()
                if self._tp is not None and self._tp.status():
                    raise RuntimeError("tensor-parallel exchange timed out (a peer rank did not reach the collective): tokens are invalid")
                hit = next((i for i, t in enumerate(toks) if t in eos), None)
                if hit is not None:
                    yield toks[:hit + 1]
                    return
                yield toks
                sent = produced
            if produced >= max_new_tokens:
                return
            self.decode_step()
            produced += 1

    def _prefill_request(self, ids, series, lengths, max_new_tokens):
        """TS encode + merge + prefill + first token of one request on the single-sequence path -> prompt length T."""
        cfg = self.config
        ps = cfg.ts["patch_size"]
        mm, counts = None, []
        if series is not None and series.shape[0] > 0:
            series = series.to(self.device, dtype=torch.float32)
            if lengths is None:
                lengths = self.ts_encoder.get_patch_cnt(series)[0].tolist()
            counts = [(int(v) + ps - 1) // ps for v in lengths]
            mm = self.get_multimodal_embeddings(timeseries=series, valid_lengths=lengths)
        full = self.expand_input_ids(list(ids), counts)
        T = len(full)
        if T + max_new_tokens > self.max_ctx:
            raise ValueError(f"prompt ({T}) + max_new_tokens ({max_new_tokens}) exceeds max_ctx={self.max_ctx}")
        emb = self.get_input_embeddings(torch.tensor(full, dtype=torch.int64), mm)
        idents = self._token_idents(full, series, lengths, counts, cfg.ts_token_start_index)
        self.reserve_kv(0, T + max_new_tokens, idents)
        n0 = self._reuse_prefix(0, idents, T)
        self._slot_idents[0] = idents
        self.prefix_stats["tokens_prefilled"] += T - n0
        self.reset()
        last = self.prefill(emb[n0:], n0, for_next_token=True)
        self.buf["pos"].fill_(T)
        self._first_token(last)
        return T

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, timeseries=None, max_new_tokens=64, max_length=None,
                 streamer=None, eos_token_id=None, do_sample=False, temperature=None, top_k=None, top_p=None,
                 seed=0, synced_gpus=False, valid_lengths=None, **kw):
        """model.generate(**inputs, max_new_tokens=...) -> LongTensor [B, T_in + new] (README.md:102, demo_hf.ipynb
        cell 5).  Greedy unless do_sample=True, like HF; with do_sample the HF GenerationConfig defaults apply to what is
        not given (temperature 1.0, top_k 50, top_p 1.0; the reference's deepspeed driver passes temperature=0.2,
        inference_tsmllm_deepspeed.py:95-100).  Rows start with the un-expanded input_ids; shorter rows are
        right-padded with pad_token_id.  A flat `timeseries` tensor is consumed across the batch in prompt order."""
        if do_sample:
            self.set_sampling(1.0 if temperature is None else temperature, 50 if top_k is None else top_k,
                              1.0 if top_p is None else top_p, seed)
        else:
            self.set_sampling(0.0)
        cfg = self.config
        ids = input_ids if isinstance(input_ids, torch.Tensor) else torch.tensor(input_ids)
        if ids.dim() == 1:
            ids = ids[None]
        ids = ids.cpu()
        mask = attention_mask.cpu() if attention_mask is not None else torch.ones_like(ids)
        if eos_token_id is None:
            eos_token_id = cfg.eos_token_id
        ts0 = cfg.ts_token_start_index
        cursor, rows, reqs = 0, [], []
        for b in range(ids.shape[0]):
            seq = ids[b][mask[b].bool()].tolist()
            n_ts = sum(1 for i in range(len(seq) - 1) if seq[i] == ts0 and seq[i + 1] == ts0 + 1)
            ser = timeseries[cursor:cursor + n_ts] if (timeseries is not None and n_ts) else None
            lens = valid_lengths[cursor:cursor + n_ts] if valid_lengths is not None else None
            cursor += n_ts
            reqs.append((seq, ser, lens))
        budget = max_new_tokens if max_length is None else max(1, max_length - max(len(r[0]) for r in reqs))
        if self.max_batch > 1 and len(reqs) > 1 and streamer is None and (self.plan.world == 1 or self._tp is not None):
            outs = self.generate_batch(reqs, budget, eos_token_id)          # continuous batching over the cache slots
        else:
            outs = []
            if streamer is not None:          # HF generate(): the prompt goes to the streamer first, then every token as it is made
                streamer.put(ids)
            for seq, ser, lens in reqs:
                if streamer is None:
                    toks = self.generate_one(seq, ser, lens, budget, eos_token_id)
                else:
                    toks = []
                    for new in self.generate_stream(seq, ser, lens, budget, eos_token_id, chunk=1):
                        for t in new:
                            streamer.put(torch.tensor([t]))
                        toks += new
                outs.append(toks)
        rows = [ids[b].tolist() + outs[b] for b in range(ids.shape[0])]
        if streamer is not None:
            streamer.end()
        n = max(len(r) for r in rows)
        pad = cfg.pad_token_id
        return torch.tensor([r + [pad] * (n - len(r)) for r in rows], dtype=torch.long)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
