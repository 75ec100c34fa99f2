"""ChatTSProcessor - host-side mirror of the reference's AutoProcessor for the <ts><ts/> protocol.

Reference interface (NetManAIOps/ChatTS):
  processor(text=[prompt...], timeseries=[np.ndarray...], padding=True, return_tensors="pt")
      -> {input_ids, attention_mask, timeseries}          README.md:98-100, demo/demo_hf.ipynb cell 5
  The HF-hub implementation (processing_qwen2_ts.py) is NOT under /root/reference; its behaviour is
  restated in the repo by chatts/utils/encoding_utils.py:
      sp_encoding              :23-37   value-preserved normalisation + "[Value Offset|Value Scaling]" prefix
      eval_prompt_to_encoding  :65-86   splice one prefix per "<ts><ts/>", zero-pad and stack the series
  and its richer prefix format is pinned by a stored notebook output (demo/demo_lora.ipynb:147):
      [offset=6.0772|scaling=3.6917|length=256|max=4.9979|min=-15.0000|left=0.0000|right=-8.2047]<ts><ts/>
  vLLM flavour (``vllm_flag=True``, chatts_vllm.py:319-348): one (ts_tokens, encoded_ts [1,2L,1]) pair per
  series, later expanded by _get_prompt_updates (:369-444).

The normalisation statistics stay on the host in float64 (they feed the prompt TEXT, which the
tokenizer sees), exactly as the reference computes them; the (value, mask) tensor is emitted in
float32 for the HIP encoder.
"""
import numpy as np

PLACEHOLDER = "<ts><ts/>"


def sp_stats(series):
    """(mean, scale_factor) of encoding_utils.py:25-31, float64."""
    x = np.asarray(series, dtype=np.float64)
    mean = np.mean(x)
    dev = x - mean
    factor = 1.0
    if np.any(np.abs(dev) >= 3.0):
        factor = np.max(np.abs(dev)) / 3.0
    return mean, factor


def sp_normalise(series):
    """-> (scaled float64 [L], mean, factor); scaled = (x-mean)/factor only when some |x-mean| >= 3."""
    x = np.asarray(series, dtype=np.float64)
    mean = np.mean(x)
    scaled = x - mean
    peak = np.max(np.abs(scaled)) if x.size else 0.0      # one pass: any(|dev| >= 3) <=> max|dev| >= 3 (NaN compares false both ways)
    factor = 1.0
    if peak >= 3.0:
        factor = peak / 3.0
        scaled = scaled / factor
    return scaled, mean, factor


def sp_prefix(mean, factor):
    return f"[Value Offset: {-mean:.4f}|Value Scaling: {factor:.4f}]" + PLACEHOLDER


def hf_prefix(series, mean, factor):
    x = np.asarray(series, dtype=np.float64)
    return (f"[offset={-mean:.4f}|scaling={factor:.4f}|length={len(x)}|max={np.max(x):.4f}|"
            f"min={np.min(x):.4f}|left={x[0]:.4f}|right={x[-1]:.4f}]" + PLACEHOLDER)


class BatchFeature(dict):
    """dict with ``.to(device)`` like transformers.BatchFeature (callers also do {k: v.to(0)})."""

    def to(self, device):
        return BatchFeature({k: (v.to(device) if hasattr(v, "to") else v) for k, v in self.items()})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ChatTSProcessor:
    """``AutoProcessor.from_pretrained(path, trust_remote_code=True, tokenizer=tok)`` equivalent."""

    def __init__(self, tokenizer, config=None, prefix_format="hf"):
        self.tokenizer = tokenizer
        self.config = config
        self.patch_size = int(config.ts["patch_size"]) if config is not None else 16
        assert prefix_format in ("hf", "sp")
        self.prefix_format = prefix_format

    @classmethod
    def register_for_auto_class(cls, auto_class="AutoProcessor"):
        """called by transformers on a class resolved through auto_map; nothing to register"""

    @classmethod
    def from_pretrained(cls, path_or_config, tokenizer=None, trust_remote_code=True, prefix_format="hf", **kw):
        """``AutoProcessor.from_pretrained(path, trust_remote_code=True, tokenizer=tokenizer)`` (README.md:90) lands here through
        the auto_map; transformers' own keywords (cache_dir, revision, ...) are accepted and ignored."""
        from .config import ChatTSConfig
        from .tokenizer import SyntheticTokenizer
        cfg = path_or_config if isinstance(path_or_config, ChatTSConfig) else ChatTSConfig.from_pretrained(path_or_config)
        return cls(tokenizer or SyntheticTokenizer.for_config(cfg), cfg, prefix_format=prefix_format)

    # ---- series handling ----------------------------------------------------------------------
    @staticmethod
    def _as_series(ts):
        if isinstance(ts, np.ndarray):
            a = ts.astype(np.float64)
        elif isinstance(ts, (list, tuple)):
            a = np.asarray(ts, dtype=np.float64)
        elif hasattr(ts, "detach"):
            a = ts.detach().cpu().numpy().astype(np.float64)
        else:      # chatts_vllm.py:277-279 raises TypeError for unsupported containers
            raise TypeError(f"Unsupported time series type: {type(ts)}")
        if a.ndim != 1:
            raise ValueError(f"each time series must be 1-D, got shape {a.shape}")
        return a

    def encode_series(self, ts):
        """-> (encoded [1, 2L, 1] float32 interleaved (value, 1.0), prefix text, meta)."""
        x = self._as_series(ts)
        if x.size == 0:
            enc = np.zeros((1, 0, 1), dtype=np.float32)
            return enc, PLACEHOLDER, {"offset": 0.0, "scale_factor": 1.0, "length": 0}
        scaled, mean, factor = sp_normalise(x)
        enc = np.stack([scaled, np.ones_like(scaled)], axis=-1).reshape(1, -1, 1).astype(np.float32)
        pfx = hf_prefix(x, mean, factor) if self.prefix_format == "hf" else sp_prefix(mean, factor)
        return enc, pfx, {"offset": float(-mean), "scale_factor": float(factor), "length": int(x.size)}

    def encode_batch_on_device(self, series_list, device="cuda"):
        """The batch form of encode_series with the statistics computed on the GPU (chatts_ts_normalise; SURVEY.md section 8f item
        4): raw series go up once as float64, the padded (value, mask) tensor [N, 2*Lmax, 1] stays on the device, and 48 bytes of
        statistics per series come back for the prompt prefixes.  -> (device tensor, [prefix...], [length...]).
        Same numbers as encode_series except that the mean is summed in the kernel's own fixed order (may differ from np.mean in
        the last bit: invisible at the prefix's %.4f and in the float32 values save for exact rounding ties)."""
        import torch
        from . import _lib
        lib = _lib.load()
        xs = [self._as_series(ts) for ts in series_list]
        n = len(xs)
        lens = [int(x.size) for x in xs]
        lmax = max(lens) if lens else 0
        if n == 0 or lmax == 0:
            return torch.zeros((n, 0, 1), dtype=torch.float32, device=device), [PLACEHOLDER] * n, lens
        raw = np.zeros((n, lmax), dtype=np.float64)
        for i, x in enumerate(xs):
            raw[i, :x.size] = x
        raw_d = torch.from_numpy(raw).to(device)
        len_d = torch.tensor(lens, dtype=torch.int32, device=device)
        enc = torch.empty((n, 2 * lmax, 1), dtype=torch.float32, device=device)
        stats = torch.empty((n, 6), dtype=torch.float64, device=device)
        _lib.check(lib.chatts_ts_normalise(raw_d.data_ptr(), len_d.data_ptr(), n, lmax, enc.data_ptr(), stats.data_ptr(), _lib.stream_ptr()))
        st = stats.cpu().numpy()
        prefixes = []
        for i in range(n):
            if lens[i] == 0:
                prefixes.append(PLACEHOLDER)
                continue
            mean, factor, mx, mn, left, right = (float(v) for v in st[i])
            if self.prefix_format == "hf":
                prefixes.append(f"[offset={-mean:.4f}|scaling={factor:.4f}|length={lens[i]}|max={mx:.4f}|min={mn:.4f}|"
                                f"left={left:.4f}|right={right:.4f}]" + PLACEHOLDER)
            else:
                prefixes.append(sp_prefix(mean, factor))
        return enc, prefixes, lens

    def splice(self, prompt, series_list):
        """eval_prompt_to_encoding (encoding_utils.py:65-86) for one prompt -> (text, [encoded...], lengths)."""
        parts = prompt.split(PLACEHOLDER)
        if len(series_list) != len(parts) - 1:
            raise ValueError(f"prompt has {len(parts) - 1} <ts><ts/> placeholders but {len(series_list)} "
                             "time series were given")
        out, encs, lens = parts[0], [], []
        for i, ts in enumerate(series_list):
            enc, pfx, meta = self.encode_series(ts)
            out += pfx + parts[i + 1]
            encs.append(enc)
            lens.append(meta["length"])
        return out, encs, lens

    @staticmethod
    def pad_stack(encs):
        """zero-pad [1, 2L_i, 1] to 2*Lmax and concatenate (encoding_utils.py:78-84) -> [N, 2*Lmax, 1]."""
        if not encs:
            return np.zeros((0, 0, 1), dtype=np.float32)
        lmax = max(e.shape[1] for e in encs)
        out = np.zeros((len(encs), lmax, 1), dtype=np.float32)
        for i, e in enumerate(encs):
            out[i, :e.shape[1]] = e[0]
        return out

    # ---- the call surface -----------------------------------------------------------------------
    def __call__(self, text=None, timeseries=None, padding=True, return_tensors="pt", vllm_flag=False, device_stats=False, **kw):
        """device_stats=True: normalisation statistics + the padded tensor are produced on the GPU (encode_batch_on_device);
        the returned `timeseries` is then already a device tensor."""
        import torch
        if text is None:
            raise ValueError("text is required")
        texts = [text] if isinstance(text, str) else list(text)
        series = list(timeseries) if timeseries is not None else []
        if device_stats and not vllm_flag:
            enc_dev, prefixes, lens = self.encode_batch_on_device(series)
            cursor, new_texts, per_prompt = 0, [], []
            for t in texts:
                parts = t.split(PLACEHOLDER)
                n = len(parts) - 1
                if cursor + n > len(series):
                    raise ValueError("not enough time series for the <ts><ts/> placeholders in the batch")
                new_texts.append("".join(parts[i] + prefixes[cursor + i] for i in range(n)) + parts[-1])
                cursor += n
                per_prompt.append(n)
            if cursor != len(series):
                raise ValueError(f"{len(series)} time series given but the prompts hold {cursor} placeholders")
            out = BatchFeature(dict(self.tokenizer(new_texts, padding=padding, return_tensors=return_tensors)))
            if series:
                out["timeseries"] = enc_dev
            self.last_lengths, self.last_series_per_prompt = lens, per_prompt
            return out
        if vllm_flag:
            # one (ts_tokens, encoded_ts) tuple per series (chatts_vllm.py:319-348)
            items = []
            for ts in series:
                enc, pfx, _ = self.encode_series(ts)
                toks = self.tokenizer.encode(pfx[:-len(PLACEHOLDER)]) if len(pfx) > len(PLACEHOLDER) else []
                items.append((toks, enc))
            ids = [self.tokenizer.encode(t) for t in texts]
            return BatchFeature({"input_ids": ids, "timeseries": items})
        # one flat series list is consumed across the batch in prompt order (inference_tsmllm_deepspeed.py:75-89)
        cursor, new_texts, encs, lens, per_prompt = 0, [], [], [], []
        for t in texts:
            n = t.count(PLACEHOLDER)
            if cursor + n > len(series):
                raise ValueError("not enough time series for the <ts><ts/> placeholders in the batch")
            nt, e, l = self.splice(t, series[cursor:cursor + n])
            cursor += n
            new_texts.append(nt); encs += e; lens += l; per_prompt.append(n)
        if cursor != len(series):
            raise ValueError(f"{len(series)} time series given but the prompts hold {cursor} placeholders")
        tok = self.tokenizer(new_texts, padding=padding, return_tensors=return_tensors)
        out = BatchFeature(dict(tok))
        arr = self.pad_stack(encs)
        if series:
            out["timeseries"] = torch.from_numpy(arr) if return_tensors == "pt" else arr
        self.last_lengths = lens          # host-side lengths let the encoder skip its one D2H sync
        self.last_series_per_prompt = per_prompt
        return out

    def batch_decode(self, *a, **k):
        return self.tokenizer.batch_decode(*a, **k)

    def decode(self, *a, **k):
        return self.tokenizer.decode(*a, **k)
