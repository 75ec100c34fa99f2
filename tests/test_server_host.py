"""CPU: the OpenAI-compatible server's host logic (chatts_amd/server.py) with a scripted engine: request parsing
(the content-part schema of NetManAIOps/ChatTS demo/vllm_api.py:43-55), ChatML rendering / multi-turn series accumulation
(chatts/utils/vllm_stream_qa.py:41-107), SSE framing, error mapping.  The real engine behind the same app runs in
tests/test_gpu_server.py."""
import json
import threading

import pytest

from chatts_amd import config as cfgmod, server
from chatts_amd.tokenizer import SyntheticTokenizer


def test_split_content_and_render_chat():
    ts1, ts2 = [1.0, 2.0, 3.5], [0.5] * 20
    text, series = server.split_content([{"type": "text", "text": "A <ts><ts/> B"}, {"timeseries": ts1}])
    assert text == "A <ts><ts/> B" and series == [ts1]
    # multi-turn: series of ALL turns, in placeholder order; default system prompt is added
    msgs = [{"role": "user", "content": [{"type": "text", "text": "TS1: <ts><ts/>"}, {"timeseries": ts1}]},
            {"role": "assistant", "content": "It rises."},
            {"role": "user", "content": [{"type": "timeseries", "timeseries": ts2}, {"type": "text", "text": "and this <ts><ts/>?"}]}]
    prompt, series = server.render_chat(msgs)
    assert series == [ts1, ts2]
    assert prompt == ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\nTS1: <ts><ts/><|im_end|>\n"
                      "<|im_start|>assistant\nIt rises.<|im_end|>\n<|im_start|>user\nand this <ts><ts/>?<|im_end|>\n<|im_start|>assistant\n")
    # the reference client sends a complete ChatML prompt as the user text: passed through verbatim
    raw = "<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\nX <ts><ts/><|im_end|><|im_start|>assistant\n"
    p2, s2 = server.render_chat([{"role": "user", "content": [{"type": "text", "text": raw}, {"timeseries": ts1}]}])
    assert p2 == raw and s2 == [ts1]
    with pytest.raises(ValueError, match="placeholders"):
        server.render_chat([{"role": "user", "content": [{"type": "text", "text": "no placeholder"}, {"timeseries": ts1}]}])
    with pytest.raises(ValueError, match="unsupported content part"):
        server.split_content([{"type": "image_url", "image_url": {"url": "x"}}])
    sp = server.sampling_from_body({"max_tokens": 7, "temperature": 0.5, "top_p": 0.95})
    assert sp["max_tokens"] == 7 and sp["temperature"] == 0.5 and sp["top_p"] == 0.95 and sp["top_k"] == 0
    assert server.sampling_from_body({})["temperature"] == 0.0            # greedy unless asked otherwise


class ScriptedEngineThread:
    """stands in for EngineThread: answers every request with the token ids of a fixed sentence, two tokens at a time"""

    def __init__(self, tok, answer):
        self.tok, self.answer, self.seen = tok, answer, []

    def submit(self, prompt, timeseries, on_tokens, holder, max_tokens, **kw):
        self.seen.append((prompt, timeseries, max_tokens, kw))
        ids = self.tok.encode(self.answer)[:max_tokens]

        class R:
            error, prompt_tokens, tokens, finish_reason = None, len(self.tok.encode(prompt)), ids, "stop"
        r = R()
        if "BAD" in prompt:
            r.error = ValueError("prompt (9999) + max_new_tokens exceeds max_ctx")

        def run():
            if r.error is not None:
                on_tokens(r, [], True)
                return
            for i in range(0, len(ids), 2):
                on_tokens(r, ids[i:i + 2], i + 2 >= len(ids))
        threading.Thread(target=run).start()


def _client(answer="The series rises, then drops sharply near point 103."):
    from starlette.testclient import TestClient
    tok = SyntheticTokenizer.for_config(cfgmod.preset("tiny-qwen2"))
    tok.encode(answer)                       # the synthetic tokenizer can decode the pieces it has seen
    et = ScriptedEngineThread(tok, answer)
    return TestClient(server.create_app(et, tok, "chatts", limit_timeseries=3)), et, answer


def test_chat_completions_blocking_and_streaming():
    client, et, answer = _client()
    ts = [float(i) for i in range(40)]
    body = {"model": "chatts", "max_tokens": 64,
            "messages": [{"role": "user", "content": [{"type": "text", "text": "Describe <ts><ts/>"}, {"timeseries": ts}]}]}
    r = client.post("/v1/chat/completions", json=body)
    assert r.status_code == 200, r.text
    d = r.json()
    assert d["object"] == "chat.completion" and d["model"] == "chatts"
    assert d["choices"][0]["message"] == {"role": "assistant", "content": answer}
    assert d["choices"][0]["finish_reason"] == "stop" and d["usage"]["completion_tokens"] == len(d["token_ids"])
    assert et.seen[-1][1] == [ts] and et.seen[-1][0].endswith("<|im_start|>assistant\n")
    # streaming: SSE chat.completion.chunk events, role first, [DONE] last; the deltas concatenate to the same text
    with client.stream("POST", "/v1/chat/completions", json=dict(body, stream=True)) as resp:
        assert resp.status_code == 200 and resp.headers["content-type"].startswith("text/event-stream")
        events = [line[6:] for line in resp.iter_lines() if line.startswith("data: ")]
    assert events[-1] == "[DONE]"
    chunks = [json.loads(e) for e in events[:-1]]
    assert all(c["object"] == "chat.completion.chunk" for c in chunks)
    assert chunks[0]["choices"][0]["delta"]["role"] == "assistant"
    assert "".join(c["choices"][0]["delta"].get("content", "") for c in chunks) == answer
    assert chunks[-1]["choices"][0]["finish_reason"] == "stop" and len(chunks) > 3
    assert client.get("/v1/models").json()["data"][0]["id"] == "chatts" and client.get("/health").json() == {"status": "ok"}


def test_errors_and_completions_endpoint():
    client, et, answer = _client()
    ts = [1.0, 2.0]
    # placeholder / series mismatch, too many series, engine-side rejection -> 400 with an OpenAI-style error object
    r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": [{"type": "text", "text": "none"}, {"timeseries": ts}]}]})
    assert r.status_code == 400 and "placeholders" in r.json()["error"]["message"]
    many = {"messages": [{"role": "user", "content": [{"type": "text", "text": "<ts><ts/>" * 4}] + [{"timeseries": ts}] * 4}]}
    r = client.post("/v1/chat/completions", json=many)
    assert r.status_code == 400 and "At most 3 timeseries" in r.json()["error"]["message"]
    r = client.post("/v1/chat/completions", json={"messages": [{"role": "user", "content": "BAD"}]})
    assert r.status_code == 400 and "max_ctx" in r.json()["error"]["message"]
    r = client.post("/v1/chat/completions", content=b"{not json", headers={"content-type": "application/json"})
    assert r.status_code == 400
    # out-of-range sampling parameters never reach the engine thread ({"temperature": -1} used to kill it for everybody)
    n_seen = len(et.seen)
    for bad in ({"temperature": -1}, {"top_p": 1.5}, {"top_p": 0}, {"top_k": -7}, {"max_tokens": -2}, {"temperature": "x"}):
        r = client.post("/v1/chat/completions", json=dict({"messages": [{"role": "user", "content": "hi"}]}, **bad))
        assert r.status_code == 400 and r.json()["error"]["message"], bad
    assert len(et.seen) == n_seen
    # raw completions: prompt + multi_modal_data like the offline LLM.generate schema
    r = client.post("/v1/completions", json={"prompt": "X <ts><ts/>", "multi_modal_data": {"timeseries": [ts]}, "max_tokens": 4})
    assert r.status_code == 200 and r.json()["object"] == "text_completion" and len(r.json()["token_ids"]) == 4
    assert et.seen[-1][0] == "X <ts><ts/>" and et.seen[-1][2] == 4


def test_incremental_decoder_holds_back_incomplete_pieces():
    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return "".join({1: "He", 2: "llo", 3: "�", 4: "é"}[i] for i in ids).replace("�é", "é")
    d = server.IncrementalDecoder(Tok())
    assert d.push([1]) == "He" and d.push([2, 3]) == "llo"      # the dangling replacement char is held back
    assert d.push([4], final=True) == "é" and d.text == "Helloé"
