"""TimeSeriesEmbedding on MI355X - same constructor / forward contract as the reference class
(NetManAIOps/ChatTS chatts/vllm/chatts_vllm.py:61-193), executed by the HIP kernels behind the C-ABI:

    forward(x [N, 2*Lmax, 1]) -> (features [P, hidden] float32, patch_cnt [N] int64)

  chatts_ts_patch_cnt   mask reduction, one wave per series      (:94-100)
  chatts_ts_patchify    patch rows + last-value pad + position-embedding gather -> [P, Kpad]   (:107-183)
  chatts_linear x n     MLP, bias + exact-erf GELU fused in the epilogue                       (:83-91,188)

Differences that are deliberate: one optional D2H sync per call instead of 2 per series
(none when the processor's host-side lengths are passed); with use_position_embedding=False the
reference crashes on ragged tails (AttributeError) - here those modes work.
"""
import torch

from . import _lib


def _pad64(k):
    """K of the first Linear, zero padded: whole 128-byte bf16 lines, the K-step of the LDS-DMA GEMM kernels."""
    return (k + 63) // 64 * 64


class TimeSeriesEmbedding:
    def __init__(self, config, device="cuda"):
        self.patch_size = config["patch_size"]
        self.num_layers = config["num_layers"]
        self.hidden_size = config["hidden_size"]
        self.num_features = config["num_features"]
        self.max_sequence_length = config["max_sequence_length"]
        self.use_position_embedding = config.get("use_position_embedding", False)
        self.use_position_idx = config.get("use_position_idx", False)
        self.embedding_dim = config.get("embedding_dim", 16)
        if self.num_features != 2:
            raise ValueError("the sp encoding has exactly 2 features (value, mask)")
        if self.use_position_embedding:
            self.mode, self.in_features = 1, self.patch_size * (1 + self.embedding_dim)
        elif self.use_position_idx:
            self.mode, self.in_features = 2, 2 * self.patch_size
        else:
            self.mode, self.in_features = 0, self.patch_size
        self.k0 = _pad64(self.in_features)           # K of the first Linear, zero padded for the MFMA GEMM
        if self.hidden_size % 32:
            raise ValueError("ts hidden_size must be a multiple of 32")
        self.device = torch.device(device)
        self.precision = "bf16x2"                     # "fp8": SPEED mode, see set_precision
        self._w8 = None
        self.position_embedding = None               # f32 [max_seq+1, emb]
        self.weights = [None] * self.num_layers      # bf16 [H, Kpad_l]
        self.biases = [None] * self.num_layers       # f32 [H]

    # ---- weights ---------------------------------------------------------------------------------
    def layer_in_features(self, l):
        return self.in_features if l == 0 else self.hidden_size

    def layer_k(self, l):
        return self.k0 if l == 0 else self.hidden_size

    def state_names(self):
        names = ["position_embedding.weight"] if self.mode == 1 else []
        for l in range(self.num_layers):
            names += [f"mlp.{2 * l}.weight", f"mlp.{2 * l}.bias"]
        return names

    def load_tensor(self, name, tensor):
        """name without the 'ts_encoder.' prefix; tensor: any float dtype, host or device."""
        self._tsw = None
        t = tensor.to(self.device)
        if name == "position_embedding.weight":
            if t.dim() != 2 or t.shape[0] != self.max_sequence_length + 1:      # row max_sequence_length is the padding_idx row
                raise ValueError(f"position_embedding.weight has {tuple(t.shape)} rows, config ts.max_sequence_length="
                                 f"{self.max_sequence_length} needs {self.max_sequence_length + 1}")
            self.position_embedding = t.float().contiguous()
            return
        l, kind = int(name.split(".")[1]) // 2, name.split(".")[2]
        if kind == "bias":
            self.biases[l] = t.float().contiguous()
        else:
            w = torch.zeros((self.hidden_size, self.layer_k(l)), dtype=torch.bfloat16, device=self.device)
            w[:, :t.shape[1]] = t.to(torch.bfloat16)
            self.weights[l] = w

    def load_synthetic(self, specs, seed):
        from . import synth
        self._tsw = None
        for s in specs:
            name = s.name[len("ts_encoder."):]
            if name == "position_embedding.weight":
                self.position_embedding = synth.fill_device(
                    torch.empty((s.rows, s.cols), dtype=torch.float32, device=self.device), s, seed)
                continue
            l, kind = int(name.split(".")[1]) // 2, name.split(".")[2]
            if kind == "bias":
                self.biases[l] = synth.fill_device(torch.empty(s.cols, dtype=torch.float32, device=self.device), s, seed)
            else:
                ld = self.layer_k(l)
                w = torch.zeros((s.rows, ld), dtype=torch.bfloat16, device=self.device)
                self.weights[l] = synth.fill_device(w, s, seed, ld=ld)

    def set_precision(self, precision):
        """"bf16x2" (default, parity grade: float32 activations as bf16 hi + lo planes) or "fp8" - the SPEED mode of
        ChatTSForCausalLM(precision="fp8"): calls with >= 64 patches quantise activations per row to e4m3 and run the MLP as fp8 x fp8
        GEMMs (chatts_linear_fp8, v_mfma_scale_f32_16x16x128_f8f6f4) on per-row power-of-two-scaled e4m3 copies of the weights; features
        ~1e-2 from the default.  Fewer patches than 64 keep the default path (weight-streaming bound: nothing to gain)."""
        if precision not in ("bf16x2", "fp8"):
            raise ValueError("TimeSeriesEmbedding precision must be 'bf16x2' or 'fp8'")
        if precision == "fp8" and self.hidden_size % 128 != 0:
            # layers after the first quantise their activations with row length H but multiply over ceil128(H): the pad bytes of the
            # e4m3 buffer would be uninitialised (NaN x 0 = NaN) - the fp8 GEMM's K-step is 128, so H must be whole steps
            raise ValueError(f"TimeSeriesEmbedding precision='fp8' needs hidden_size % 128 == 0 (got {self.hidden_size})")
        self.precision = precision
        self._w8 = None

    def _fp8_weights(self):
        if self._w8 is None:
            from .modeling import quantize_fp8_rows
            self._w8 = []
            for l, w in enumerate(self.weights):
                k = (w.shape[1] + 127) // 128 * 128           # K of the fp8 GEMM: whole 128-byte MFMA steps (zero padded)
                wp = torch.zeros((w.shape[0], k), dtype=torch.bfloat16, device=w.device)
                wp[:, :w.shape[1]] = w
                q, sc, _ = quantize_fp8_rows(wp)
                self._w8.append((q, sc, k))
        return self._w8

    def _replay_fp8(self):
        """the fp8 speed-mode form of replay_last: patchify -> [quantise rows -> fp8 GEMM (+bias, GELU)] x layers"""
        import ctypes as C
        lib = _lib.load()
        x, row_off_dev, vl_dev, n, lmax, maxvl, P, out = self._last
        w8 = self._fp8_weights()
        H, dev, st = self.hidden_size, x.device, _lib.stream_ptr()
        k0 = w8[0][2]
        key = ("fp8", P, dev)
        if getattr(self, "_buf8_key", None) != key:
            self._bufs8 = dict(feat=torch.empty((P, k0), dtype=torch.float32, device=dev), h=torch.empty((P, H), dtype=torch.float32, device=dev),
                               a8=torch.empty((P, max(k0, H)), dtype=torch.uint8, device=dev), sa=torch.empty(P, dtype=torch.float32, device=dev))
            self._buf8_key = key
        B = self._bufs8
        pa = _lib.PatchifyArgs(series=_lib.ptr(x), row_off=_lib.ptr(row_off_dev), valid_len=_lib.ptr(vl_dev),
                               pos_table=_lib.ptr(self.position_embedding), out=_lib.ptr(B["feat"]), n_series=n, lmax=lmax,
                               patch_size=self.patch_size, mode=self.mode, emb_dim=self.embedding_dim, max_seq_len=self.max_sequence_length,
                               max_valid_len=maxvl, total_patches=P, ld_out=k0, out_hi=None, out_lo=None)
        _lib.check(lib.chatts_ts_patchify(C.byref(pa), st))
        src, k = B["feat"], k0
        for l in range(self.num_layers):
            _lib.check(lib.chatts_quantize_rows_fp8(_lib.ptr(src), P, k, k, None, 0.0, _lib.ptr(B["a8"]), k, _lib.ptr(B["sa"]), st))
            last = l == self.num_layers - 1
            q, sc, kq = w8[l]
            dst = out if last else B["h"]
            fa = _lib.LinearFp8Args(a8=_lib.ptr(B["a8"]), a_scale=_lib.ptr(B["sa"]), w8=_lib.ptr(q), w_scale=_lib.ptr(sc),
                                    bias=_lib.ptr(self.biases[l]), resid=None, c=_lib.ptr(dst), m=P, n=H, k=kq, lda8=k, ldw8=kq, ldc=H,
                                    epilogue=_lib.EPI_NONE if last else _lib.EPI_GELU)
            _lib.check(lib.chatts_linear_fp8(C.byref(fa), st))
            src, k = B["h"], H
        return out

    def weight_bytes(self):
        n = sum(w.numel() * 2 for w in self.weights) + sum(b.numel() * 4 for b in self.biases)
        return n

    # ---- forward -----------------------------------------------------------------------------------
    def get_patch_cnt(self, x):
        """chatts_vllm.py:198-207 on the device: [N, 2*Lmax, 1] -> (valid_len int32 [N], patch_cnt int64 [N])."""
        lib = _lib.load()
        n = x.shape[0]
        x = x.reshape(n, -1)
        lmax = x.shape[1] // 2
        vl = torch.empty(n, dtype=torch.int32, device=x.device)
        pc = torch.empty(n, dtype=torch.int64, device=x.device)
        _lib.check(lib.chatts_ts_patch_cnt(_lib.ptr(x), n, lmax, self.patch_size, _lib.ptr(vl), _lib.ptr(pc),
                                          _lib.stream_ptr()))
        return vl, pc

    def forward(self, x, valid_lengths=None):
        """x: [N, 2*Lmax, 1] (value, mask) interleaved, any float dtype, on the GPU.

        valid_lengths: optional host list of the true lengths (from the processor); when given no
        device->host copy happens at all.
        """
        lib = _lib.load()
        if not x.is_cuda:
            raise RuntimeError("chatts_amd has no CPU path: move the timeseries tensor to the GPU")
        n = x.shape[0]
        x = x.reshape(n, -1).float().contiguous()
        lmax = x.shape[1] // 2
        dev = x.device
        vl_dev, pc_dev = self.get_patch_cnt(x) if n else (torch.empty(0, dtype=torch.int32, device=dev),
                                                          torch.empty(0, dtype=torch.int64, device=dev))
        host_given = valid_lengths is not None
        if not host_given:
            valid_lengths = vl_dev.tolist()          # the single D2H sync of this path
        vl_host = [int(v) for v in valid_lengths]
        if len(vl_host) != n:
            raise ValueError("valid_lengths does not match the number of series")
        if any(v < 0 or v > lmax for v in vl_host):
            raise ValueError(f"valid_lengths {vl_host} exceed the padded series length {lmax}")
        if any(v > self.max_sequence_length for v in vl_host) and self.mode == 1:
            raise IndexError("index out of range in self")     # nn.Embedding lookup, chatts_vllm.py:165
        ps = self.patch_size
        pcs = [(v + ps - 1) // ps for v in vl_host]
        row_off = [0]
        for c in pcs:
            row_off.append(row_off[-1] + c)
        P = row_off[-1]
        if P == 0:
            return torch.empty((0, self.hidden_size), dtype=torch.float32, device=dev), pc_dev
        row_off_dev = torch.tensor(row_off, dtype=torch.int32).to(dev, non_blocking=True)
        if host_given:
            vl_dev = torch.tensor(vl_host, dtype=torch.int32).to(dev, non_blocking=True)
        # one C call: patchify + the whole MLP (chatts_ts_encode); buffers are cached per patch count
        key = (P, dev)
        if getattr(self, "_buf_key", None) != key:
            H = self.hidden_size
            ws_bytes = max(int(lib.chatts_linear_workspace(P, H, self.layer_k(l))) for l in range(self.num_layers))
            self._bufs = dict(feat=torch.empty((P, self.k0), dtype=torch.float32, device=dev),
                              h0=torch.empty((P, H), dtype=torch.float32, device=dev),
                              h1=torch.empty((P, H), dtype=torch.float32, device=dev),
                              ws=torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev), ws_bytes=ws_bytes)
            self._buf_key = key
        if getattr(self, "_tsw", None) is None:
            if self.num_layers > 8:
                raise ValueError("ts num_layers > 8 is not supported")
            tw = _lib.TsWeights(patch_size=ps, num_layers=self.num_layers, hidden=self.hidden_size, mode=self.mode,
                                emb_dim=self.embedding_dim, max_seq_len=self.max_sequence_length, in_features_pad=self.k0,
                                pos_table=_lib.ptr(self.position_embedding))
            for l in range(self.num_layers):
                tw.w[l] = _lib.ptr(self.weights[l])
                tw.b[l] = _lib.ptr(self.biases[l])
            self._tsw = tw
        B = self._bufs
        out = torch.empty((P, self.hidden_size), dtype=torch.float32, device=dev)
        self._last = (x, row_off_dev, vl_dev, n, lmax, max(vl_host) if vl_host else 0, P, out)
        self.replay_last()
        return out, pc_dev

    def replay_last(self):
        """Enqueue the kernels of the most recent forward() again (same device inputs and buffers): what bench.py times
        between HIP events for the encoder's roofline line - no host-side staging in the timed region."""
        import ctypes as C
        if self.precision == "fp8" and self._last[6] >= 64:
            return self._replay_fp8()
        lib, B = _lib.load(), self._bufs
        x, row_off_dev, vl_dev, n, lmax, maxvl, P, out = self._last
        _lib.check(lib.chatts_ts_encode(_lib.ptr(x), _lib.ptr(row_off_dev), _lib.ptr(vl_dev), n, lmax, maxvl, P,
                                        C.byref(self._tsw), _lib.ptr(B["feat"]), _lib.ptr(B["h0"]), _lib.ptr(B["h1"]),
                                        _lib.ptr(out), _lib.ptr(B["ws"]), B["ws_bytes"], _lib.stream_ptr()))
        return out

    __call__ = forward
