"""ctypes binding of libchatts_amd.so (the C-ABI declared in include/chatts_amd.h).

There is NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

OK, E_BADARG, E_SHAPE, E_COUNT_MISMATCH, E_LAUNCH, E_WORKSPACE = 0, -1, -2, -3, -4, -5
EPI_NONE, EPI_GELU, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3

c_void_p, c_int, c_int32, c_int64, c_float, c_size_t, c_uint32 = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint32)


class PatchifyArgs(C.Structure):
    _fields_ = [("series", c_void_p), ("row_off", c_void_p), ("valid_len", c_void_p), ("pos_table", c_void_p),
                ("out", c_void_p), ("n_series", c_int), ("lmax", c_int), ("patch_size", c_int), ("mode", c_int),
                ("emb_dim", c_int), ("max_seq_len", c_int), ("max_valid_len", c_int), ("total_patches", c_int),
                ("ld_out", c_int), ("out_hi", c_void_p), ("out_lo", c_void_p)]


class TsWeights(C.Structure):
    _fields_ = [("patch_size", c_int), ("num_layers", c_int), ("hidden", c_int), ("mode", c_int), ("emb_dim", c_int),
                ("max_seq_len", c_int), ("in_features_pad", c_int), ("pos_table", c_void_p), ("w", c_void_p * 8),
                ("b", c_void_p * 8)]


ABI_VERSION = 9            # CHATTS_ABI_VERSION of include/chatts_amd.h this binding was written against (checked by load())
W8_FP8, W8_INT8 = 0, 1     # ChattsLinearArgs.w8_format


class LinearArgs(C.Structure):
    _fields_ = [("a", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("resid", c_void_p), ("c", c_void_p),
                ("norm_w", c_void_p), ("norm_eps", c_float), ("m", c_int), ("n", c_int), ("k", c_int),
                ("lda", c_int), ("ldw", c_int), ("ldc", c_int), ("epilogue", c_int), ("workspace", c_void_p),
                ("workspace_bytes", c_size_t), ("w8", c_void_p), ("w8_scale", c_void_p), ("ldw8", c_int),
                ("a_hi", c_void_p), ("a_lo", c_void_p), ("ld_planes", c_int),
                ("c_hi", c_void_p), ("c_lo", c_void_p), ("ld_cplanes", c_int),
                ("post_norm_w", c_void_p), ("post_norm_eps", c_float), ("post_hi", c_void_p), ("post_lo", c_void_p), ("ld_post", c_int),
                ("w4", c_void_p), ("w4_sz", c_void_p), ("ldw4", c_int), ("w4_group", c_int),
                ("w8_format", c_int), ("tp_reduce", c_void_p), ("w_tiled", c_void_p), ("planes_tiled", c_int)]


class LinearFp8Args(C.Structure):
    _fields_ = [("a8", c_void_p), ("a_scale", c_void_p), ("w8", c_void_p), ("w_scale", c_void_p), ("bias", c_void_p), ("resid", c_void_p),
                ("c", c_void_p), ("m", c_int), ("n", c_int), ("k", c_int), ("lda8", c_int), ("ldw8", c_int), ("ldc", c_int),
                ("epilogue", c_int)]


class LinearF16qArgs(C.Structure):
    """ChattsLinearF16qArgs: the prefill projection on f16q planes (csrc/gemm_f16q.hip)"""
    _fields_ = [("a_hi", c_void_p), ("a_lo8", c_void_p), ("a_scale", c_void_p), ("ld_a", c_int), ("ld_scale", c_int),
                ("w16", c_void_p), ("w8", c_void_p), ("w8_exp", c_void_p), ("ldw", c_int),
                ("bias", c_void_p), ("resid", c_void_p), ("c", c_void_p), ("m", c_int), ("n", c_int), ("k", c_int), ("ldc", c_int),
                ("epilogue", c_int), ("c_hi", c_void_p), ("c_lo8", c_void_p), ("c_scale", c_void_p), ("ld_cplanes", c_int),
                ("ld_cscale", c_int), ("post_norm_w", c_void_p), ("post_norm_eps", c_float), ("post_hi", c_void_p),
                ("post_lo8", c_void_p), ("post_scale", c_void_p), ("ld_post", c_int), ("ld_pscale", c_int),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("planes_tiled", c_int), ("w_tiled", c_int)]


class KvCache(C.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("max_ctx", c_int),
                ("block_table", c_void_p), ("block_size", c_int), ("table_stride", c_int)]      # paged form: block_table != NULL


class LayerWeights(C.Structure):
    _fields_ = [("input_norm", c_void_p), ("qkv", c_void_p), ("qkv_bias", c_void_p), ("q_norm", c_void_p),
                ("k_norm", c_void_p), ("o", c_void_p), ("post_norm", c_void_p), ("gate_up", c_void_p),
                ("down", c_void_p), ("qkv8", c_void_p), ("qkv8_scale", c_void_p), ("o8", c_void_p),
                ("o8_scale", c_void_p), ("gate_up8", c_void_p), ("gate_up8_scale", c_void_p), ("down8", c_void_p),
                ("down8_scale", c_void_p),
                ("qkv4", c_void_p), ("qkv4_sz", c_void_p), ("o4", c_void_p), ("o4_sz", c_void_p), ("gate_up4", c_void_p),
                ("gate_up4_sz", c_void_p), ("down4", c_void_p), ("down4_sz", c_void_p), ("w4_group", c_int),
                ("qkv_t", c_void_p), ("o_t", c_void_p), ("gate_up_t", c_void_p), ("down_t", c_void_p),
                ("qkv16", c_void_p), ("qkv_q8", c_void_p), ("qkv_q8e", c_void_p), ("o16", c_void_p), ("o_q8", c_void_p), ("o_q8e", c_void_p),
                ("gate_up16", c_void_p), ("gate_up_q8", c_void_p), ("gate_up_q8e", c_void_p),
                ("down16", c_void_p), ("down_q8", c_void_p), ("down_q8e", c_void_p)]


class PrefillSegment(C.Structure):
    _fields_ = [("row0", c_int), ("t", c_int), ("pos0", c_int), ("slot", c_int)]


class SamplingArgs(C.Structure):
    _fields_ = [("temperature", c_float), ("top_k", c_int), ("top_p", c_float), ("seed", c_uint32), ("n_kept", c_void_p),
                ("kept_mass", c_void_p), ("temperature_rows", c_void_p), ("top_k_rows", c_void_p), ("top_p_rows", c_void_p),
                ("seed_rows", c_void_p)]


class DecoderConfig(C.Structure):
    _fields_ = [("hidden", c_int), ("n_layers", c_int), ("n_q", c_int), ("n_kv", c_int), ("head_dim", c_int),
                ("inter", c_int), ("vocab_local", c_int64), ("vocab_offset", c_int64), ("rms_eps", c_float),
                ("max_ctx", c_int), ("max_pos", c_int), ("tp_world", c_int), ("embed_rows", c_int64), ("embed_offset", c_int64), ("w8_format", c_int)]


class DecoderWeights(C.Structure):
    _fields_ = [("layers", C.POINTER(LayerWeights)), ("final_norm", c_void_p), ("lm_head", c_void_p),
                ("lm_head8", c_void_p), ("lm_head8_scale", c_void_p),
                ("embed", c_void_p), ("cos_tab", c_void_p), ("sin_tab", c_void_p)]


class DecoderBuffers(C.Structure):
    _fields_ = [("kv_k", c_void_p), ("kv_v", c_void_p), ("x", c_void_p), ("xn", c_void_p), ("qkv", c_void_p),
                ("attn", c_void_p), ("act", c_void_p), ("delta", c_void_p), ("logits", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("t_max", c_int), ("max_batch", c_int),
                ("planes_hi", c_void_p), ("planes_lo", c_void_p), ("planes2_hi", c_void_p), ("planes2_lo", c_void_p),
                ("tp_pair_logit", c_void_p), ("tp_pair_token", c_void_p), ("logits_full", c_void_p),
                ("kv_block_table", c_void_p), ("kv_block_size", c_int), ("kv_table_stride", c_int), ("kv_pool_blocks", c_int)]


# name -> (restype, argtypes); this table IS the list of symbols include/chatts_amd.h declares
SIGNATURES = {
    "chatts_last_error": (C.c_char_p, []),
    "chatts_abi_version": (c_int, []),
    "chatts_device_cus": (c_int, []),
    "chatts_set_option": (c_int, [C.c_char_p, c_int]),
    "chatts_unset_option": (c_int, [C.c_char_p]),
    "chatts_get_option": (c_int, [C.c_char_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "chatts_option_name": (C.c_char_p, [c_int]),
    "chatts_fill_hash": (c_int, [c_void_p, c_int, c_uint32, c_float, c_int, c_int64, c_int64, c_int64, c_int64,
                                 c_int64, c_int64, c_void_p]),
    "chatts_ts_normalise": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "chatts_ts_patch_cnt": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "chatts_ts_patchify": (c_int, [C.POINTER(PatchifyArgs), c_void_p]),
    "chatts_ts_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, C.POINTER(TsWeights), c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "chatts_linear_workspace": (c_size_t, [c_int, c_int, c_int]),
    "chatts_gemv_ksplit": (c_int, [c_int, c_int, c_int, c_int]),
    "chatts_linear": (c_int, [C.POINTER(LinearArgs), c_void_p]),
    "chatts_quantize_rows_fp8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "chatts_linear_fp8": (c_int, [C.POINTER(LinearFp8Args), c_void_p]),
    "chatts_split_f16q": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "chatts_weights_f16q": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "chatts_rmsnorm_f16q": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "chatts_tile_e4m3_bytes": (c_size_t, [c_int, c_int]),
    "chatts_tile_e4m3": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chatts_linear_f16q_workspace": (c_size_t, [c_int, c_int, c_int]),
    "chatts_linear_f16q": (c_int, [C.POINTER(LinearF16qArgs), c_void_p]),
    "chatts_decoder_set_prefill_fp8": (c_int, [c_void_p, c_int]),
    "chatts_decoder_set_prefill_f16q": (c_int, [c_void_p, c_int]),
    "chatts_tile_bf16_elems": (c_size_t, [c_int, c_int]),
    "chatts_tile_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "chatts_split_bf16x2": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "chatts_embed_merge": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "chatts_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "chatts_rmsnorm_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "chatts_rope_kv_write": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                     c_int, c_void_p, C.POINTER(KvCache), c_void_p]),
    "chatts_attn_workspace": (c_size_t, [c_int, c_int, c_int]),
    "chatts_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, C.POINTER(KvCache), c_void_p, c_int,
                                 c_void_p, c_size_t, c_void_p]),
    "chatts_attention_decode_fused": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                              c_int, c_void_p, C.POINTER(KvCache), c_void_p, c_int, c_void_p,
                                              c_size_t, c_void_p]),
    "chatts_attention_decode_batched": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                                c_void_p, c_int, c_void_p, C.POINTER(KvCache), c_size_t, c_void_p, c_int,
                                                c_void_p, c_size_t, c_void_p]),
    "chatts_argmax_batched": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_int, c_void_p]),
    "chatts_argmax_workspace": (c_size_t, [c_int]),
    "chatts_argmax_batched_ws": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "chatts_sample_batched": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, C.POINTER(SamplingArgs), c_void_p, c_void_p,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p]),
    "chatts_embed_token_batched": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "chatts_decoder_select_sequence": (c_int, [c_void_p, c_int]),
    "chatts_decoder_set_sampling": (c_int, [c_void_p, C.POINTER(SamplingArgs)]),
    "chatts_decoder_layer_part_batched": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "chatts_decoder_decode_step_batched": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_int64, c_void_p, c_int, c_void_p]),
    "chatts_argmax": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "chatts_embed_token": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "chatts_decoder_create": (c_void_p, [C.POINTER(DecoderConfig), C.POINTER(DecoderWeights),
                                         C.POINTER(DecoderBuffers)]),
    "chatts_decoder_destroy": (None, [c_void_p]),
    "chatts_decoder_workspace": (c_size_t, [C.POINTER(DecoderConfig), c_int, c_int]),
    "chatts_decoder_layer_part": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "chatts_residual_add": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "chatts_decoder_layer_part_add": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "chatts_decoder_prefill": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "chatts_decoder_prefill_last": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "chatts_decoder_prefill_packed": (c_int, [c_void_p, C.POINTER(PrefillSegment), c_int, c_void_p]),
    "chatts_decoder_logits": (c_int, [c_void_p, c_int, c_void_p]),
    "chatts_decoder_logits_batched": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "chatts_decoder_decode_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_void_p]),
    "chatts_decoder_set_tp": (c_int, [c_void_p, c_void_p]),
    "chatts_decoder_select_tokens": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                             c_void_p, c_int, C.POINTER(SamplingArgs), c_void_p]),
    # tensor-parallel exchange (tp.hip)
    "chatts_tp_buffer_bytes": (c_size_t, [c_int, c_int64]),
    "chatts_tp_buffer_bytes_bulk": (c_size_t, [c_int, c_int64, c_int64]),
    "chatts_tp_bulk_elems": (c_int64, [c_void_p]),
    "chatts_allreduce_bulk": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "chatts_tp_buffer_alloc": (c_int, [c_size_t, C.POINTER(c_void_p), c_void_p]),
    "chatts_tp_buffer_free": (c_int, [c_void_p]),
    "chatts_tp_init": (c_void_p, [c_int, c_int, c_void_p, c_void_p, c_size_t, c_int64]),
    "chatts_tp_init_local": (c_void_p, [c_int, c_int, C.POINTER(c_void_p), c_size_t, c_int64]),
    "chatts_tp_init_loopback": (c_void_p, [c_int, c_int, c_void_p, c_size_t, c_int64]),
    "chatts_tp_destroy": (None, [c_void_p]),
    "chatts_tp_rank": (c_int, [c_void_p]),
    "chatts_tp_world": (c_int, [c_void_p]),
    "chatts_tp_max_elems": (c_int64, [c_void_p]),
    "chatts_tp_cross_device": (c_int, [c_void_p]),
    "chatts_tp_set_cross_device": (c_int, [c_void_p, c_int]),
    "chatts_tp_set_bulk_release": (c_int, [c_void_p, c_int]),
    "chatts_tp_bulk_release": (c_int, [c_void_p]),
    "chatts_tp_status": (c_int, [c_void_p]),
    "chatts_tp_reset": (c_int, [c_void_p, c_void_p]),
    "chatts_tp_flush_epochs": (c_int, [c_void_p, c_void_p]),
    "chatts_tp_pending": (c_int, [c_void_p]),
    "chatts_allreduce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "chatts_allgather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "chatts_tp_argmax": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_int, c_void_p]),
}
TP_HANDLE_BYTES = 64


class ChattsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libchatts_amd error {code}: {msg}")
        self.code = code
        self.msg = msg


def lib_path():
    """the in-tree library; CHATTS_AMD_LIB points at another build of it (A/B runs of compile-time knobs) - still no fallback"""
    return os.environ.get("CHATTS_AMD_LIB") or _build.LIB


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built. Run `python -m chatts_amd.build` "
            "(or __graft_entry__.build()). chatts_amd has no CPU fallback.")
    # torch must be imported BEFORE the .so: both link libamdhip64, and torch ships its own copy.  Whichever is loaded
    # first serves the whole process; if ours pulled in /opt/rocm's runtime first, torch's allocations and our launches
    # would live in two HIP runtimes ("no ROCm-capable device is detected" on the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.chatts_abi_version() != ABI_VERSION:      # e.g. CHATTS_AMD_LIB pointing at a stale A/B build: struct layouts would not match
        raise RuntimeError(f"{path} has ABI version {lib.chatts_abi_version()}, this package binds version {ABI_VERSION} "
                           "(include/chatts_amd.h: CHATTS_ABI_VERSION): rebuild it with `python -m chatts_amd.build --force`")
    _LIB = lib
    sync_env()
    return lib


def option_names():
    """the tuning options the library knows (chatts_option_name)"""
    lib, out, i = load(), [], 0
    while True:
        n = lib.chatts_option_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def set_option(name, value):
    """chatts_set_option: select an alternative path / launch geometry (None = back to the shipped choice)"""
    lib = load()
    if value is None:
        check(lib.chatts_unset_option(name.encode()))
    else:
        check(lib.chatts_set_option(name.encode(), int(value)))


def get_option(name):
    v, isset = c_int(0), c_int(0)
    check(load().chatts_get_option(name.encode(), C.byref(v), C.byref(isset)))
    return v.value if isset.value else None


class options:
    """with _lib.options(GEMM_SK=2, ROPE_FUSE=0): ...   - set for the block, restored afterwards"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


_ENV_ALIASES = {"bf16x2": 0, "bf16": 1}      # CHATTS_GEMM_PRECISION=bf16


_ENV_SET = {}      # option name -> value this module last mirrored from the environment


def sync_env():
    """Make the options mirror the process environment: every known option NAME takes the value of CHATTS_<NAME> when that variable is
    set; an option this function set EARLIER goes back to unset when its variable has disappeared.  Options set programmatically
    (set_option, e.g. ChatTSForCausalLM(precision='bf16') -> GEMM_PRECISION) are left alone unless their variable is set (ADVICE r5:
    the earlier form cleared the whole table first).  Called once by load(); tools that flip os.environ between calls call it again.
    This is the ONLY place the environment reaches the library: the C side never calls getenv."""
    lib = _LIB
    if lib is None:
        return
    i = 0
    while True:
        n = lib.chatts_option_name(i)
        if n is None:
            break
        i += 1
        name = n.decode()
        v = os.environ.get("CHATTS_" + name)
        if v is None or v == "":
            if name in _ENV_SET:
                del _ENV_SET[name]
                check(lib.chatts_unset_option(n))
            continue
        try:
            iv = _ENV_ALIASES[v] if v in _ENV_ALIASES else int(v)
        except ValueError:
            raise RuntimeError(f"CHATTS_{name}={v!r}: tuning options are integers")
        check(lib.chatts_set_option(n, iv))
        _ENV_SET[name] = iv


def check(rc):
    """Translate a negative return code into the exception type the reference raises in that situation."""
    if rc == OK:
        return
    msg = load().chatts_last_error().decode("utf-8", "replace")
    if rc == E_COUNT_MISMATCH:
        raise ValueError(msg)                 # vLLM merge_multimodal_embeddings raises ValueError
    raise ChattsError(rc, msg)


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
