"""GPU: the OpenAI-compatible server (chatts_amd/server.py) on the real engine.  The request body is the one the
reference's client sends (NetManAIOps/ChatTS demo/vllm_api.py:43-55: text part + {"timeseries": [...]} parts); the
answers must be the tokens LLM.generate / the CPU oracle produce for the same prompt, blocking and streamed, one request
at a time and with several requests in flight (continuous batching over the cache slots)."""
import json
import threading

import numpy as np
import pytest

from chatts_amd import config as cfgmod, server, synth
from oracle import pipeline, synth as osynth
from tests.util import chat_prompt, random_walk_series

pytestmark = pytest.mark.gpu


def _oracle_tokens(cfg, llm, prompt, series, n, seed):
    sd = osynth.state_dict(synth.all_specs(cfg), seed)
    inputs = llm.processor(text=[prompt], timeseries=series if series else None, return_tensors="pt")
    ts = inputs["timeseries"].numpy() if series else None
    return pipeline.generate(cfg, sd, inputs["input_ids"][0].tolist(), ts, n)["tokens"]


def _sse(resp):
    events = [line[6:] for line in resp.iter_lines() if line.startswith("data: ")]
    assert events[-1] == "[DONE]"
    return [json.loads(e) for e in events[:-1]]


@pytest.mark.parametrize("max_num_seqs", [1, 3])
def test_chat_completions_match_oracle_blocking_streaming_concurrent(max_num_seqs):
    from starlette.testclient import TestClient
    cfg = cfgmod.preset("tiny-qwen2")
    seed, n = 3, 10
    app = server.build_server(cfg, max_model_len=512, max_num_seqs=max_num_seqs, seed=seed)
    llm = app.state.llm
    try:
        client = TestClient(app)
        rng = np.random.default_rng(11)
        cases = []
        for lengths in ([64, 30], [256], [17]):
            series = [random_walk_series(rng, L) for L in lengths]
            body_text = f"I have {len(lengths)} time series. " + " ".join(f"TS{i} is of length {L}: <ts><ts/>;" for i, L in enumerate(lengths))
            cases.append((body_text, series))
        # (1) the reference client's body: a full ChatML prompt as the text part, series as {"timeseries": ...} parts
        text, series = cases[0]
        raw = chat_prompt([len(s) for s in series])
        body = {"model": "chatts", "max_tokens": n, "ignore_eos": True,
                "messages": [{"role": "user", "content": [{"type": "text", "text": raw}] + [{"timeseries": s.tolist()} for s in series]}]}
        want = _oracle_tokens(cfg, llm, raw, series, n, seed)
        d = client.post("/v1/chat/completions", json=body).json()
        assert d["token_ids"] == want and d["choices"][0]["finish_reason"] == "length"
        assert d["choices"][0]["message"]["content"] == llm.processor.tokenizer.decode(want, skip_special_tokens=True)
        assert d["usage"]["completion_tokens"] == n
        # (2) streamed: same text, token by token
        with client.stream("POST", "/v1/chat/completions", json=dict(body, stream=True)) as resp:
            chunks = _sse(resp)
        streamed = "".join(c["choices"][0]["delta"].get("content", "") for c in chunks)
        assert streamed == d["choices"][0]["message"]["content"] and chunks[-1]["choices"][0]["finish_reason"] == "length"
        # (3) plain messages (the server renders ChatML + default system prompt), several requests in flight
        results = {}

        def post(i, text, series):
            b = {"max_tokens": n, "ignore_eos": True,
                 "messages": [{"role": "user", "content": [{"type": "text", "text": text}] + [{"timeseries": s.tolist()} for s in series]}]}
            results[i] = client.post("/v1/chat/completions", json=b).json()
        th = [threading.Thread(target=post, args=(i, t, s)) for i, (t, s) in enumerate(cases)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for i, (text, series) in enumerate(cases):
            prompt = f"<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n{text}<|im_end|>\n<|im_start|>assistant\n"
            assert results[i]["token_ids"] == _oracle_tokens(cfg, llm, prompt, series, n, seed), i
        # (4) a request that cannot fit is a 400, and the engine keeps serving
        big = {"max_tokens": 600, "messages": [{"role": "user", "content": "hello"}]}
        r = client.post("/v1/chat/completions", json=big)
        assert r.status_code == 400 and "max_ctx" in r.json()["error"]["message"]
        assert client.post("/v1/chat/completions", json=body).json()["token_ids"] == want
        # (5) sampled requests are reproducible per seed and differ from greedy
        sb = dict(body, temperature=0.8, top_p=0.95, seed=5)
        a = client.post("/v1/chat/completions", json=sb).json()["token_ids"]
        b = client.post("/v1/chat/completions", json=sb).json()["token_ids"]
        assert a == b and a != want
    finally:
        app.state.engine_thread.close()


def test_hf_generate_streams_token_by_token():
    """model.generate(**inputs, streamer=...) (demo/demo_hf.ipynb:157-165): the prompt first, then one put() per token."""
    import torch
    from chatts_amd.modeling import ChatTSForCausalLM
    from chatts_amd.processing import ChatTSProcessor
    cfg = cfgmod.preset("tiny-qwen2")
    proc = ChatTSProcessor.from_pretrained(cfg)
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=3, max_ctx=512, max_prefill_tokens=512)
    rng = np.random.default_rng(2)
    series = [random_walk_series(rng, 64)]
    inputs = proc(text=[chat_prompt([64])], timeseries=series, return_tensors="pt")

    class Streamer:
        def __init__(self):
            self.puts, self.ended = [], False

        def put(self, v):
            self.puts.append(v.clone())

        def end(self):
            self.ended = True
    st = Streamer()
    out = model.generate(**inputs.to("cuda"), max_new_tokens=7, eos_token_id=[], streamer=st)
    ref = model.generate(**inputs.to("cuda"), max_new_tokens=7, eos_token_id=[])
    assert torch.equal(out, ref) and st.ended
    assert st.puts[0].shape == inputs["input_ids"].shape                   # the prompt
    assert [int(p) for p in st.puts[1:]] == out[0, inputs["input_ids"].shape[1]:].tolist() and len(st.puts) == 8
