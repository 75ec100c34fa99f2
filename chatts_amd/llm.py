"""vLLM-style offline surface: ``LLM(...).generate([{"prompt", "multi_modal_data": {"timeseries": [...]}}], SamplingParams)``.

Mirrors how the reference drives vLLM (NetManAIOps/ChatTS demo/demo_vllm.py:30-63, chatts/utils/llm_utils.py:154-182):
same constructor keywords, same dict input schema, results expose ``.outputs[0].text``.  vLLM itself is not a
dependency; requests are served one after another on the HIP engine (continuous batching: SURVEY.md section 8f).
"""
import os

import numpy as np


class SamplingParams:
    def __init__(self, max_tokens=16, temperature=0.0, top_p=1.0, top_k=-1, seed=None, stop_token_ids=None,
                 ignore_eos=False, **kw):
        self.max_tokens = int(max_tokens)
        self.temperature = temperature          # 0 = greedy (this engine's default; the reference's drivers pass 0.2 / 0.5)
        self.top_p = top_p
        self.top_k = top_k                      # -1 / 0 = off, like vLLM
        self.seed = seed
        self.stop_token_ids = list(stop_token_ids or [])
        self.ignore_eos = ignore_eos


class CompletionOutput:
    def __init__(self, text, token_ids):
        self.text, self.token_ids, self.index = text, token_ids, 0


class RequestOutput:
    def __init__(self, prompt, prompt_token_ids, outputs):
        self.prompt, self.prompt_token_ids, self.outputs = prompt, prompt_token_ids, outputs


def _load_checkpoint_tokenizer(path):
    """The HF tokenizer stored in a checkpoint directory (demo/demo_vllm.py builds prompts with it).  A directory written
    by ChatTSConfig.save_pretrained for the synthetic-weight tests marks itself with `synthetic_tokenizer` in config.json."""
    import json
    with open(os.path.join(path, "config.json")) as f:
        if json.load(f).get("synthetic_tokenizer"):
            return None
    has_files = any(os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "vocab.json", "tokenizer.model"))
    if not has_files:
        raise ValueError(f"{path} holds no tokenizer files (tokenizer.json / vocab.json): pass tokenizer=... explicitly; "
                         "the synthetic stand-in tokenizer would feed meaningless ids to real weights")
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, trust_remote_code=False)


def _build_llm(model, kw):
    """factory of chatts_amd.tp_spawn.TpGroup: every spawned rank (and the leader) builds the same LLM"""
    return LLM(model, **kw)


class LLM:
    def __init__(self, model, tensor_parallel_size=1, max_model_len=6000, limit_mm_per_prompt=None,
                 trust_remote_code=True, gpu_memory_utilization=None, seed=0, tokenizer=None, comm=None,
                 max_num_seqs=1, block_size=None, num_gpu_blocks_override=None, **kw):
        from .config import PRESETS, ChatTSConfig, preset
        from .modeling import ChatTSForCausalLM
        from .processing import ChatTSProcessor
        from .tp import Comm, LocalComm
        self._tp_group = None
        if tensor_parallel_size > 1 and comm is None and "WORLD_SIZE" not in os.environ:
            import torch.distributed as dist
            if not dist.is_initialized():
                # the reference's call shape (demo/demo_vllm.py:30, llm_utils.py:154): ONE plain process asks for k GPUs and the
                # engine spawns its own workers, like vLLM does.  This process becomes rank 0; the followers replay every generate()
                from .tp_spawn import TpGroup
                args = dict(tensor_parallel_size=tensor_parallel_size, max_model_len=max_model_len,
                            limit_mm_per_prompt=limit_mm_per_prompt, trust_remote_code=trust_remote_code,
                            gpu_memory_utilization=gpu_memory_utilization, seed=seed, tokenizer=tokenizer,
                            max_num_seqs=max_num_seqs, block_size=block_size, num_gpu_blocks_override=num_gpu_blocks_override, **kw)
                group, inner = TpGroup.launch(tensor_parallel_size, _build_llm, (model, args))
                self.__dict__.update(inner.__dict__)
                self._tp_group = group
                return
        self._seed = int(seed)              # also the default seed of sampled generation (SamplingParams.seed overrides)
        if comm is None:
            comm = Comm() if tensor_parallel_size > 1 else LocalComm()
        if comm.world != tensor_parallel_size:
            raise ValueError(f"tensor_parallel_size={tensor_parallel_size} needs {tensor_parallel_size} ranks launched "
                             f"one per GPU (torchrun); this process group has {comm.world}")
        # max_num_seqs (vLLM's name): cache slots decoded together (continuous batching); 1 = one request at a time
        # block_size / num_gpu_blocks_override (vLLM's names): block-paged KV cache; fewer blocks than
        # max_num_seqs x max_model_len / block_size oversubscribes the slots (requests then wait for blocks at admission)
        mk = dict(comm=comm, max_ctx=max_model_len, max_prefill_tokens=min(2048, max_model_len),
                  max_batch=max(1, int(max_num_seqs)), kv_block_size=block_size, kv_pool_blocks=num_gpu_blocks_override)
        if isinstance(model, ChatTSConfig):
            self.model = ChatTSForCausalLM.from_synthetic(model, seed=seed, **mk)
        elif isinstance(model, str) and model in PRESETS:
            self.model = ChatTSForCausalLM.from_synthetic(preset(model), seed=seed, **mk)
        elif isinstance(model, str) and os.path.isdir(model):
            self.model = ChatTSForCausalLM.from_pretrained(model, **mk)
            if tokenizer is None:           # real weights need the checkpoint's own tokenizer, never the synthetic stand-in
                tokenizer = _load_checkpoint_tokenizer(model)
        else:
            raise ValueError(f"model must be a checkpoint directory, a ChatTSConfig or one of {sorted(PRESETS)}")
        self.config = self.model.config
        self.processor = ChatTSProcessor.from_pretrained(self.config, tokenizer=tokenizer)
        self.limit = (limit_mm_per_prompt or {}).get("timeseries", 50)      # chatts_vllm.py:220 caps at 50

    def get_tokenizer(self):
        return self.processor.tokenizer

    def shutdown(self):
        """stop the tensor-parallel workers this object spawned (no-op otherwise)"""
        if self._tp_group is not None:
            self._tp_group.shutdown()
            self._tp_group = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    def generate(self, prompts, sampling_params=None, use_tqdm=False):
        if self._tp_group is not None:      # spawned followers run the same call now; all ranks meet inside the exchange kernels
            self._tp_group.call("generate", prompts, sampling_params)
        sp = sampling_params or SamplingParams()
        self.model.set_sampling(sp.temperature or 0.0, max(int(sp.top_k or 0), 0), sp.top_p,
                                self._seed if sp.seed is None else sp.seed)
        if isinstance(prompts, (str, dict)):
            prompts = [prompts]
        import torch
        reqs, metas = [], []
        for req in prompts:
            if isinstance(req, str):
                req = {"prompt": req}
            series = list((req.get("multi_modal_data") or {}).get("timeseries", []))
            if len(series) > self.limit:
                raise ValueError(f"At most {self.limit} timeseries may be provided in one prompt, got {len(series)}")
            text, encs, lens = self.processor.splice(req["prompt"], [np.asarray(s, dtype=np.float64) for s in series])
            ids = self.processor.tokenizer.encode(text)
            ser = torch.from_numpy(self.processor.pad_stack(encs)) if encs else None
            reqs.append((ids, ser, lens))
            metas.append((req["prompt"], ids))
        e = self.config.eos_token_id
        eos = None if sp.ignore_eos else ((list(e) if isinstance(e, (list, tuple)) else [e]) + sp.stop_token_ids)
        toks_all = self.model.generate_batch(reqs, sp.max_tokens, eos)      # sequential when max_num_seqs == 1
        tok = self.processor.tokenizer
        return [RequestOutput(p, ids, [CompletionOutput(tok.decode(t, skip_special_tokens=True), t)])
                for (p, ids), t in zip(metas, toks_all)]
