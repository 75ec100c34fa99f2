"""OpenAI-compatible HTTP server for the MI355X-native ChatTS engine.

Replaces what the reference deploys with ``vllm serve ./ckpt --served-model-name chatts --limit-mm-per-prompt timeseries=15
--port 12345`` (NetManAIOps/ChatTS scripts/start_vllm_server.sh:2-12) and talks to with the openai client
(demo/vllm_api.py:43-55):

    client.chat.completions.create(model="chatts", messages=[{"role": "user", "content":
        [{"type": "text", "text": prompt}] + [{"timeseries": ts} for ts in ts_list]}])

i.e. ``POST /v1/chat/completions`` whose message content may carry ``{"timeseries": [floats]}`` parts next to the text
parts; series are matched to the ``<ts><ts/>`` placeholders of the conversation in order, over ALL turns (multi-turn
history accumulates its series like chatts/utils/vllm_stream_qa.py:41-107 does).  ``stream: true`` answers with
server-sent ``chat.completion.chunk`` events, token by token.  Also: ``POST /v1/completions`` (raw prompt +
``multi_modal_data.timeseries``), ``GET /v1/models``, ``GET /health``.

    python -m chatts_amd.server --model chatts-14b --port 12345 [--max-num-seqs 16] [--tensor-parallel-size 1]
"""
import argparse
import asyncio
import json
import time
import uuid

DEFAULT_SYSTEM = "You are a helpful assistant."


# ---------------------------------------------------------------------------------------------------------------
# request -> (prompt text, series list): pure host logic, unit-tested on CPU
# ---------------------------------------------------------------------------------------------------------------
def split_content(content):
    """message content (str | list of parts) -> (text, [series...]).  Parts: {"type": "text", "text": ...},
    {"timeseries": [...]} (demo/vllm_api.py:50) or {"type": "timeseries", "timeseries": [...]}."""
    if content is None:
        return "", []
    if isinstance(content, str):
        return content, []
    text, series = [], []
    for part in content:
        if isinstance(part, str):
            text.append(part)
        elif "timeseries" in part:
            series.append([float(v) for v in part["timeseries"]])
        elif part.get("type") in (None, "text", "input_text"):
            text.append(part.get("text", ""))
        else:
            raise ValueError(f"unsupported content part type {part.get('type')!r} (text and timeseries are supported)")
    return "".join(text), series


def render_chat(messages, default_system=DEFAULT_SYSTEM):
    """ChatML rendering (the Qwen template the reference builds by hand, vllm_stream_qa.py:91-94, demo_hf.ipynb cell 5).
    A message whose text already is a complete ChatML prompt (starts with <|im_start|>, as demo/vllm_api.py sends it) is
    passed through verbatim.  -> (prompt, [series...] in placeholder order)."""
    if not messages:
        raise ValueError("messages must not be empty")
    series, parts = [], []
    texts = []
    for m in messages:
        t, s = split_content(m.get("content"))
        texts.append((m.get("role", "user"), t))
        series += s
    if len(texts) == 1 and texts[0][1].lstrip().startswith("<|im_start|>"):
        prompt = texts[0][1]
    else:
        if texts[0][0] != "system":
            parts.append(f"<|im_start|>system\n{default_system}<|im_end|>\n")
        for role, t in texts:
            parts.append(f"<|im_start|>{role}\n{t}<|im_end|>\n")
        parts.append("<|im_start|>assistant\n")
        prompt = "".join(parts)
    n_ph = prompt.count("<ts><ts/>")
    if n_ph != len(series):
        raise ValueError(f"the conversation holds {n_ph} <ts><ts/> placeholders but {len(series)} timeseries parts")
    return prompt, series


def sampling_from_body(body, default_max_tokens=512):
    mt = body.get("max_completion_tokens") or body.get("max_tokens") or default_max_tokens
    temp = body.get("temperature")
    top_p = body.get("top_p")
    try:
        out = dict(max_tokens=int(mt), temperature=0.0 if temp is None else float(temp), top_p=1.0 if top_p is None else float(top_p),
                   top_k=int(body.get("top_k") or 0), seed=None if body.get("seed") is None else int(body["seed"]),
                   stop_token_ids=[int(t) for t in (body.get("stop_token_ids") or [])], ignore_eos=bool(body.get("ignore_eos", False)))
    except (TypeError, ValueError) as e:
        raise ValueError(f"malformed sampling parameter: {e}")
    from .engine import validate_sampling
    validate_sampling(**out)             # ValueError -> 400 (never reaches the engine thread)
    return out


class IncrementalDecoder:
    """token ids -> text deltas.  Decodes the whole sequence each time and emits the new suffix, holding back a trailing
    replacement character (an incomplete multi-byte piece), like vLLM's incremental detokenizer."""

    def __init__(self, tokenizer):
        self.tok, self.ids, self.text = tokenizer, [], ""

    def push(self, new_ids, final=False):
        self.ids += list(new_ids)
        full = self.tok.decode(self.ids, skip_special_tokens=True)
        if not final and full.endswith("�"):
            full = full[:-1]
        delta = full[len(self.text):] if full.startswith(self.text) else full
        self.text = full if full.startswith(self.text) else self.text + delta
        return delta


# ---------------------------------------------------------------------------------------------------------------
# the app
# ---------------------------------------------------------------------------------------------------------------
def create_app(engine_thread, tokenizer, served_model_name="chatts", limit_timeseries=15, default_max_tokens=512):
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse, StreamingResponse
    app = FastAPI(title="chatts_amd")
    created = int(time.time())

    def error(status, msg, kind="invalid_request_error"):
        return JSONResponse(status_code=status, content={"error": {"message": msg, "type": kind, "code": status}})

    async def run(prompt, series, sp, loop):
        """submit to the engine thread; -> (request holder, asyncio queue of (new_token_ids, finished))."""
        q = asyncio.Queue()
        holder = []

        def on_tokens(r, new, finished):              # called on the engine thread
            loop.call_soon_threadsafe(q.put_nowait, (list(new), finished, r))
        engine_thread.submit(prompt=prompt, timeseries=series, on_tokens=on_tokens, holder=holder, **sp)
        return holder, q

    async def next_event(q):
        """next (new_tokens, finished, request) of a request; a dead engine thread surfaces as an error instead of a hang"""
        while True:
            try:
                return await asyncio.wait_for(q.get(), timeout=5.0)
            except asyncio.TimeoutError:
                err = getattr(engine_thread, "error", None)
                if err is not None:
                    raise RuntimeError(f"engine thread died: {err}")

    @app.get("/health")
    async def health():
        return {"status": "ok"}

    @app.get("/v1/models")
    async def models():
        return {"object": "list", "data": [{"id": served_model_name, "object": "model", "created": created, "owned_by": "chatts_amd"}]}

    async def complete(body, chat):
        loop = asyncio.get_running_loop()
        try:
            if chat:
                prompt, series = render_chat(body.get("messages") or [])
            else:
                prompt = body.get("prompt")
                if isinstance(prompt, list):
                    prompt = prompt[0]
                series = [[float(v) for v in s] for s in ((body.get("multi_modal_data") or {}).get("timeseries") or [])]
                if not isinstance(prompt, str):
                    raise ValueError("prompt must be a string")
                if prompt.count("<ts><ts/>") != len(series):
                    raise ValueError(f"prompt holds {prompt.count('<ts><ts/>')} <ts><ts/> placeholders but {len(series)} timeseries")
            if len(series) > limit_timeseries:
                raise ValueError(f"At most {limit_timeseries} timeseries may be provided in one prompt, got {len(series)}")
            sp = sampling_from_body(body, default_max_tokens)
        except (ValueError, TypeError, KeyError) as e:
            return error(400, str(e))
        rid = ("chatcmpl-" if chat else "cmpl-") + uuid.uuid4().hex[:24]
        model_name = body.get("model") or served_model_name
        obj = "chat.completion" if chat else "text_completion"
        try:
            holder, q = await run(prompt, series, sp, loop)
        except RuntimeError as e:
            return error(500, str(e), "server_error")
        dec = IncrementalDecoder(tokenizer)

        def chunk(delta_text, finish=None, first=False):
            if chat:
                delta = {"content": delta_text} if delta_text or not first else {}
                if first:
                    delta = {"role": "assistant", "content": delta_text}
                ch = {"index": 0, "delta": delta, "finish_reason": finish}
            else:
                ch = {"index": 0, "text": delta_text, "finish_reason": finish}
            return {"id": rid, "object": obj + ".chunk" if chat else obj, "created": int(time.time()), "model": model_name, "choices": [ch]}

        if body.get("stream"):
            async def gen():
                first = True
                while True:
                    try:
                        new, finished, r = await next_event(q)
                    except RuntimeError as e:
                        yield "data: " + json.dumps({"error": {"message": str(e), "type": "server_error"}}) + "\n\n"
                        break
                    if r.error is not None:
                        yield "data: " + json.dumps({"error": {"message": str(r.error), "type": "invalid_request_error"}}) + "\n\n"
                        break
                    text = dec.push(new, final=finished)
                    if text or first:
                        yield "data: " + json.dumps(chunk(text, None, first)) + "\n\n"
                        first = False
                    if finished:
                        yield "data: " + json.dumps(chunk("", r.finish_reason)) + "\n\n"
                        break
                yield "data: [DONE]\n\n"
            return StreamingResponse(gen(), media_type="text/event-stream")
        while True:
            try:
                new, finished, r = await next_event(q)
            except RuntimeError as e:
                return error(500, str(e), "server_error")
            if r.error is not None:
                return error(400, str(r.error))
            dec.push(new, final=finished)
            if finished:
                break
        usage = {"prompt_tokens": r.prompt_tokens, "completion_tokens": len(r.tokens), "total_tokens": r.prompt_tokens + len(r.tokens)}
        if chat:
            choice = {"index": 0, "message": {"role": "assistant", "content": dec.text}, "finish_reason": r.finish_reason}
        else:
            choice = {"index": 0, "text": dec.text, "finish_reason": r.finish_reason}
        return {"id": rid, "object": obj, "created": int(time.time()), "model": model_name, "choices": [choice], "usage": usage,
                "token_ids": list(r.tokens)}

    @app.post("/v1/chat/completions")
    async def chat_completions(request: Request):
        try:
            body = await request.json()
        except Exception:
            return error(400, "request body is not valid JSON")
        return await complete(body, chat=True)

    @app.post("/v1/completions")
    async def completions(request: Request):
        try:
            body = await request.json()
        except Exception:
            return error(400, "request body is not valid JSON")
        return await complete(body, chat=False)

    return app


def build_server(model, tensor_parallel_size=1, max_model_len=6000, max_num_seqs=1, seed=0, tokenizer=None,
                 served_model_name="chatts", limit_timeseries=15, block_size=None, num_gpu_blocks_override=None,
                 enable_chunked_prefill=False, max_num_batched_tokens=512):
    """LLM (model + processor) -> Engine -> EngineThread -> FastAPI app.
    Tensor parallel (one process per GPU under torchrun, process group initialised): rank 0 gets the app, the other ranks get
    None after serving as followers until the leader shuts down (engine.follow)."""
    from .engine import ControlPlane, Engine, EngineThread, follow
    from .llm import LLM
    llm = LLM(model, tensor_parallel_size=tensor_parallel_size, max_model_len=max_model_len, max_num_seqs=max_num_seqs, seed=seed,
              tokenizer=tokenizer, limit_mm_per_prompt={"timeseries": limit_timeseries}, block_size=block_size,
              num_gpu_blocks_override=num_gpu_blocks_override)
    control = ControlPlane.create() if tensor_parallel_size > 1 else None
    engine = Engine(llm.model, llm.processor, prefill_chunk_tokens=max_num_batched_tokens if enable_chunked_prefill else None)
    if control is not None and control.rank != 0:
        follow(engine, control)                  # returns when the leader publishes the shutdown
        return None
    et = EngineThread(engine, device=llm.model.device, control=control)
    app = create_app(et, llm.processor.tokenizer, served_model_name, limit_timeseries)
    app.state.engine_thread, app.state.llm = et, llm
    return app


def main():
    ap = argparse.ArgumentParser(description="OpenAI-compatible ChatTS server on the MI355X-native engine")
    ap.add_argument("--model", required=True, help="checkpoint directory or a preset name (chatts-14b, chatts-8b: synthetic weights)")
    ap.add_argument("--served-model-name", default="chatts")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=12345)
    ap.add_argument("--max-model-len", type=int, default=6000)
    ap.add_argument("--max-num-seqs", type=int, default=1)
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--limit-mm-per-prompt", default="timeseries=15")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--enable-chunked-prefill", action="store_true", help="while other sequences decode, prefill a long prompt "
                    "--max-num-batched-tokens rows at a time between their decode steps")
    ap.add_argument("--max-num-batched-tokens", type=int, default=512)
    ap.add_argument("--block-size", type=int, default=None, help="block-paged KV cache: positions per block (power of two >= 64); "
                    "default: one contiguous cache per sequence slot")
    ap.add_argument("--num-gpu-blocks-override", type=int, default=None, help="with --block-size: blocks in the pool (fewer than "
                    "max-num-seqs x max-model-len / block-size oversubscribes the slots)")
    args = ap.parse_args()
    limit = int(dict(kv.split("=") for kv in args.limit_mm_per_prompt.split(",")).get("timeseries", 15))
    import uvicorn
    if args.tensor_parallel_size > 1:            # torchrun --nproc-per-node N -m chatts_amd.server --tensor-parallel-size N ...
        import os
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = int(os.environ.get("CHATTS_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        backend = os.environ.get("CHATTS_DIST_BACKEND", "nccl")
        dist.init_process_group(backend=backend, **({"device_id": torch.device(f"cuda:{dev}")} if backend == "nccl" else {}))
    app = build_server(args.model, args.tensor_parallel_size, args.max_model_len, args.max_num_seqs, args.seed,
                       served_model_name=args.served_model_name, limit_timeseries=limit, block_size=args.block_size,
                       num_gpu_blocks_override=args.num_gpu_blocks_override, enable_chunked_prefill=args.enable_chunked_prefill,
                       max_num_batched_tokens=args.max_num_batched_tokens)
    if app is None:
        return                                   # a follower rank: the leader has shut down
    try:
        uvicorn.run(app, host=args.host, port=args.port, log_level="info")
    finally:
        app.state.engine_thread.close()          # tells the follower ranks to leave


if __name__ == "__main__":
    main()
