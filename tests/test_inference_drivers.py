"""Host logic of the batch-inference drivers and their on-disk formats (chatts_amd/inference.py), no GPU: the engine is a stub.
Reference behaviour: chatts/utils/inference_tsmllm_vllm.py, inference_tsmllm_deepspeed.py, llm_utils.py:235-341,
evaluation/evaluate_tsmllm_models.py:35-42."""
import json
import os

import numpy as np
import pytest
import torch

from chatts_amd import inference as inf


def _dataset(tmp_path, n=5):
    recs = []
    for i in range(n):
        k = i % 3                        # 0, 1 or 2 series; record 0 is text-only
        recs.append({"question": f"Q{i}: " + " ".join("<ts><ts/>" for _ in range(k)) + " ünïcode?",
                     "timeseries": [[float(i + j + t) for t in range(16 * (j + 1) + i)] for j in range(k)] if k else None,
                     "cols": [f"c{j}" for j in range(k)], "attributes": [], "ability_types": [], "answer": "a"})
    p = tmp_path / "dataset.json"
    p.write_text(json.dumps(recs, ensure_ascii=False))
    return str(p), recs


class _StubLLM:
    """records the requests; answers with a digest of prompt + series"""

    def __init__(self):
        self.calls = []

    def generate(self, reqs, sp, use_tqdm=False):
        from chatts_amd.llm import CompletionOutput, RequestOutput
        self.calls.append((reqs, sp))
        outs = []
        for r in reqs:
            series = (r.get("multi_modal_data") or {}).get("timeseries", [])
            outs.append(RequestOutput(r["prompt"], [], [CompletionOutput(f"{len(r['prompt'])}|{[len(s) for s in series]}", [])]))
        return outs


def test_dataset_validation(tmp_path):
    path, recs = _dataset(tmp_path)
    assert inf.load_eval_dataset(path) == recs
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps([{"question": "one <ts><ts/>", "timeseries": []}]))
    with pytest.raises(ValueError, match="placeholders"):
        inf.load_eval_dataset(str(bad))
    bad.write_text(json.dumps({"question": "x"}))
    with pytest.raises(ValueError, match="JSON list"):
        inf.load_eval_dataset(str(bad))
    bad.write_text(json.dumps([{"timeseries": []}]))
    with pytest.raises(ValueError, match="question"):
        inf.load_eval_dataset(str(bad))


def test_prompts_and_shards():
    assert inf.chat_prompt("hi") == "<|im_start|>system\nYou are a helpful assistant.<|im_end|><|im_start|>user\nhi<|im_end|><|im_start|>assistant\n"
    assert inf.qwen_chat_prompt("hi", "S") == "<|im_start|>system\nS<|im_end|>\n<|im_start|>user\nhi<|im_end|>\n<|im_start|>assistant\n"
    assert inf.shard_indices(7, 3, 1) == [1, 4]
    assert sorted(sum((inf.shard_indices(10, 4, r) for r in range(4)), [])) == list(range(10))
    with pytest.raises(ValueError):
        inf.shard_indices(3, 2, 2)


def test_llm_driver_writes_reference_format(tmp_path):
    path, recs = _dataset(tmp_path)
    stub = _StubLLM()
    out = inf.run("unused", path, "expA", workdir=str(tmp_path), surface="llm", llm=stub, max_tokens=7, temperature=0.2,
                  log=lambda *_: None)
    assert out == os.path.join(str(tmp_path), "exp", "expA", "generated_answer.json")
    got = json.load(open(out))
    assert [g["idx"] for g in got] == list(range(5))
    assert all(set(g) == {"idx", "question_text", "response"} for g in got)
    assert got[1]["question_text"] == recs[1]["question"]            # the vLLM driver stores the raw question
    reqs = [r for call in stub.calls for r in call[0]]
    assert len(reqs) == 5 and "multi_modal_data" not in reqs[0]
    assert reqs[2]["prompt"] == inf.qwen_chat_prompt(recs[2]["question"])
    assert [len(s) for s in reqs[2]["multi_modal_data"]["timeseries"]] == [18, 34]
    assert got[2]["response"] == f"{len(reqs[2]['prompt'])}|[18, 34]"
    sp = stub.calls[0][1]
    assert sp.max_tokens == 7 and sp.temperature == 0.2
    assert "ünïcode" in open(out, encoding="utf-8").read()           # ensure_ascii=False like the reference


def test_replicas_split_and_merge(tmp_path):
    path, recs = _dataset(tmp_path, 7)
    for r in range(3):
        inf.run("unused", path, "expB", workdir=str(tmp_path), surface="llm", llm=_StubLLM(), world=3, rank=r, log=lambda *_: None)
    exp_dir = os.path.join(str(tmp_path), "exp", "expB")
    assert sorted(os.listdir(exp_dir)) == [f"generated_answer_3_{r}.json" for r in range(3)]
    assert [a["idx"] for a in json.load(open(os.path.join(exp_dir, "generated_answer_3_1.json")))] == [1, 4]
    merged = inf.merge_answer_files(exp_dir, 7)
    assert [m["idx"] for m in merged] == list(range(7))
    os.remove(os.path.join(exp_dir, "generated_answer_3_2.json"))
    merged = inf.merge_answer_files(exp_dir, 7)
    assert merged[2] == {} and merged[5] == {} and merged[3]["idx"] == 3


def test_client_surface():
    c = inf.LLMClient(engine="dryrun")
    assert c.wait_for_ready()
    assert c.llm_batch_generate(["a", "b"], None, dryrun_outputs=["x", "y"]) == ["x", "y"]
    with pytest.raises(AssertionError):
        inf.LLMClient(llm=_StubLLM()).llm_batch_generate(["a"], [[], []])
    with pytest.raises(NotImplementedError):
        inf.LLMClient(engine="llama")
    c = inf.LLMClient(llm=_StubLLM(), replica=1, replicas=2)
    ans = c.llm_batch_generate(["p0", "p1", "p2"], [[np.ones(4)], [np.ones(5)], None], use_chat_template=False)
    assert ans[0] is None and ans[2] is None and ans[1] == "2|[5]"
    sp = c.default_sampling_params()
    assert (sp.temperature, sp.top_p, sp.stop_token_ids) == (0.5, 0.95, [151643, 151645])
    c.kill()


class _StubTok:
    def decode(self, ids, skip_special_tokens=False):
        assert skip_special_tokens
        return " ".join(str(i) for i in ids)


class _StubProc:
    tokenizer = _StubTok()

    def __call__(self, text, timeseries, padding=True, return_tensors="pt"):
        self.seen = (text, [len(s) for s in timeseries])
        n = 6
        ids = torch.arange(n).repeat(len(text), 1)
        mask = torch.ones_like(ids)
        mask[:, 0] = 0                                   # one pad column
        return {"input_ids": ids, "attention_mask": mask, "timeseries": torch.zeros(len(timeseries), 4, 1)}


class _StubModel:
    class config:
        ts = {"patch_size": 16}

    def generate(self, input_ids=None, attention_mask=None, timeseries=None, max_length=None, temperature=None, **kw):
        self.kw = dict(max_length=max_length, temperature=temperature)
        return torch.cat([input_ids, torch.full((input_ids.shape[0], 3), 9)], dim=1)


def test_hf_driver_counts_tokens_like_the_reference(tmp_path):
    path, recs = _dataset(tmp_path, 4)
    model, proc = _StubModel(), _StubProc()
    out = inf.run("unused", path, "expC", workdir=str(tmp_path), surface="hf", hf_model=model, hf_processor=proc, world=2, rank=1,
                  max_tokens=1024, log=lambda *_: None)
    assert out.endswith("generated_answer_2_1.json")
    got = json.load(open(out))
    assert [g["idx"] for g in got] == [1, 3]
    assert got[0]["question_text"] == inf.chat_prompt(recs[1]["question"])     # the deepspeed driver stores the wrapped prompt
    assert got[0]["response"] == "9 9 9"
    assert got[0]["num_tokens"] == (16 + 1) // 16 + 5                         # sum(len)//patch + attended prompt tokens
    assert model.kw == {"max_length": 6 + 1024, "temperature": 0.2}
    assert proc.seen[1] == []                                                 # record 3 is text-only (k = 0)


def test_training_jsonl_round_trip(tmp_path):
    recs = [{"input": "q <ts><ts/>", "output": "a", "timeseries": [np.arange(4.0)]},
            {"input": "q2", "output": "ü", "timeseries": []}]
    p = str(tmp_path / "train.jsonl")
    inf.write_training_jsonl(p, recs)
    back = inf.read_training_jsonl(p)
    assert back[0] == {"input": "q <ts><ts/>", "output": "a", "timeseries": [[0.0, 1.0, 2.0, 3.0]]}
    assert back[1]["output"] == "ü"
    with open(p, "a") as f:
        f.write("\n" + json.dumps({"input": "x"}) + "\n")
    with pytest.raises(ValueError, match="output"):
        inf.read_training_jsonl(p)
