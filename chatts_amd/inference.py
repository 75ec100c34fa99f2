"""Batch-inference drivers and the on-disk formats either side of the hot path.

What the reference's two evaluation drivers do, on this engine:

* ``chatts/utils/inference_tsmllm_vllm.py:52-95``: read ``evaluation/dataset/*.json`` (records ``{question, timeseries, cols,
  attributes, ability_types, answer}``), push every question through ``LLMClient.llm_batch_generate`` (chat template applied,
  ``{"prompt", "multi_modal_data": {"timeseries": [...]}}`` requests, ``SamplingParams(max_tokens=512, temperature=0.2)``), write
  ``exp/<EXP>/generated_answer.json`` = ``[{idx, question_text, response}]`` (``ensure_ascii=False, indent=4``).
* ``chatts/utils/inference_tsmllm_deepspeed.py:62-147``: one process per GPU, sample ``i`` belongs to rank ``i % world_size``,
  HF ``processor`` + ``model.generate(max_length = prompt + 1024, temperature=0.2)``, the continuation decoded with
  ``skip_special_tokens``; writes ``exp/<EXP>/generated_answer_<world>_<rank>.json`` = ``[{idx, question_text, response,
  num_tokens}]`` with ``num_tokens = sum(len(series)) // patch_size + prompt tokens``.

``evaluation/evaluate_tsmllm_models.py:35-42`` then merges every ``*generated_answer*.json`` of the directory by ``idx``
(`merge_answer_files` here).  Training records (``chatts/align/uts_template_qa.py:128-133``) are JSON lines ``{input, output,
timeseries}`` (`read_training_jsonl` / `write_training_jsonl`).

The data-parallel replicas of ``LLMClient`` (``llm_utils.py:251-266``: one worker process per ``gpus_per_model`` GPUs fed from a
queue) are plain process-level replicas: here every torchrun replica group takes the strided share ``i % replicas == replica``
of the questions and writes its own answer file; nothing is exchanged on the data path.

    python -m chatts_amd.inference --model ckpt --dataset evaluation/dataset/dataset_a.json --exp chatts_dataset_a
"""
import argparse
import json
import os

import numpy as np

DEFAULT_SYSTEM = "You are a helpful assistant."
PLACEHOLDER = "<ts><ts/>"


# ---------------------------------------------------------------------------------------------------------------
# formats
# ---------------------------------------------------------------------------------------------------------------
def load_eval_dataset(path):
    """-> list of records; every record needs `question` (str) and `timeseries` (list of 1-D number lists, or None/absent for a
    text-only question: inference_tsmllm_deepspeed.py:76-84 skips those).  Other keys (cols, attributes, ability_types, answer)
    travel untouched for the evaluation side."""
    with open(path) as f:
        data = json.load(f)
    if not isinstance(data, list):
        raise ValueError(f"{path}: an evaluation set is a JSON list of records, got {type(data).__name__}")
    for i, rec in enumerate(data):
        if not isinstance(rec, dict) or not isinstance(rec.get("question"), str):
            raise ValueError(f"{path}: record {i} has no `question` string")
        ts = rec.get("timeseries")
        if ts is not None:
            if not isinstance(ts, list) or any(not isinstance(s, list) for s in ts):
                raise ValueError(f"{path}: record {i}: `timeseries` must be a list of series (lists of numbers)")
            n_ph = rec["question"].count(PLACEHOLDER)
            if n_ph != len(ts):
                raise ValueError(f"{path}: record {i}: {n_ph} <ts><ts/> placeholders but {len(ts)} series")
    return data


def chat_prompt(question, system=DEFAULT_SYSTEM):
    """The ChatML wrapping both drivers end up with (inference_tsmllm_deepspeed.py:133 literally; llm_utils.py:274-279 through
    the tokenizer's Qwen chat template, whose rendering differs only by a newline after each <|im_end|>: `qwen_template=True`
    in `LLMClient` below)."""
    return f"<|im_start|>system\n{system}<|im_end|><|im_start|>user\n{question}<|im_end|><|im_start|>assistant\n"


def qwen_chat_prompt(question, system=DEFAULT_SYSTEM):
    """tokenizer.apply_chat_template([system, user], add_generation_prompt=True) of the Qwen2/Qwen3 tokenizers, rendered."""
    return f"<|im_start|>system\n{system}<|im_end|>\n<|im_start|>user\n{question}<|im_end|>\n<|im_start|>assistant\n"


def shard_indices(n, world=1, rank=0):
    """the strided split of inference_tsmllm_deepspeed.py:64-66"""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return [i for i in range(n) if i % world == rank]


def answer_file(exp_dir, world=None, rank=None):
    name = "generated_answer.json" if world is None else f"generated_answer_{int(world)}_{int(rank)}.json"
    return os.path.join(exp_dir, name)


def write_answers(path, answers, questions):
    """answers: {idx: {"response": str[, "num_tokens": int]}} -> the reference's list-of-records file"""
    out = []
    for idx in sorted(answers):
        rec = {"idx": int(idx), "question_text": questions[idx], "response": answers[idx]["response"]}
        if "num_tokens" in answers[idx]:
            rec["num_tokens"] = int(answers[idx]["num_tokens"])
        out.append(rec)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wt") as f:
        json.dump(out, f, ensure_ascii=False, indent=4)
    return out


def merge_answer_files(exp_dir, n):
    """evaluate_tsmllm_models.py:35-42: every file of the directory whose name holds `generated_answer` and ends in .json, later
    files overwrite earlier ones per idx; unanswered positions stay {}."""
    merged = [{} for _ in range(n)]
    for name in sorted(os.listdir(exp_dir)):
        if "generated_answer" in name and name.endswith(".json"):
            with open(os.path.join(exp_dir, name)) as f:
                for ans in json.load(f):
                    if not 0 <= ans["idx"] < n:
                        raise ValueError(f"{name}: idx {ans['idx']} outside the {n}-record dataset")
                    merged[ans["idx"]] = ans
    return merged


def read_training_jsonl(path):
    """{input, output, timeseries} per line (uts_template_qa.py:128-133); blank lines are skipped"""
    out = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            line = line.strip()
            if not line:
                continue
            rec = json.loads(line)
            for k in ("input", "output", "timeseries"):
                if k not in rec:
                    raise ValueError(f"{path}:{ln}: training record without `{k}`")
            out.append(rec)
    return out


def write_training_jsonl(path, records):
    with open(path, "wt") as f:
        for rec in records:
            f.write(json.dumps({"input": rec["input"], "output": rec["output"],
                                "timeseries": [np.asarray(s, dtype=np.float64).tolist() for s in rec["timeseries"]]},
                               ensure_ascii=False) + "\n")


# ---------------------------------------------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------------------------------------------
class LLMClient:
    """`chatts.utils.llm_utils.LLMClient` (llm_utils.py:235-341) over this engine: same constructor keywords and the same
    `wait_for_ready / llm_batch_generate / kill` calls.  The reference starts one vLLM worker process per `gpus_per_model` GPUs and
    feeds them from a queue; here the engine lives in the calling process (TP = `gpus_per_model` when launched one process per
    GPU), and throughput replicas are separate launches that each take a strided share of the questions (`replica`, `replicas`).
    `engine` may be 'vllm-ts' (time series) or 'dryrun' (echoes `dryrun_outputs`, llm_utils.py:187-224)."""

    def __init__(self, model_path=None, engine="vllm-ts", num_gpus=1, gpu_range=None, gpus_per_model=1, batch_size=16, sample_n=1,
                 chat_template=None, system_prompt=DEFAULT_SYSTEM, llm=None, max_model_len=6000, replica=0, replicas=1,
                 qwen_template=True, **llm_kw):
        if engine not in ("vllm-ts", "dryrun"):
            raise NotImplementedError(f"Unrecognized inference engine: {engine}")
        if sample_n != 1:
            raise NotImplementedError("sample_n > 1 (several samples per prompt) is not served by this engine")
        if chat_template is not None:
            raise NotImplementedError("custom jinja chat templates are not rendered here: pass ready prompts with use_chat_template=False")
        self.engine, self.system_prompt, self.batch_size = engine, system_prompt, int(batch_size)
        self.replica, self.replicas, self.qwen_template = int(replica), int(replicas), bool(qwen_template)
        shard_indices(0, self.replicas, self.replica)
        self.llm = llm
        if llm is None and engine == "vllm-ts":
            from .llm import LLM
            self.llm = LLM(model=model_path, tensor_parallel_size=int(gpus_per_model), max_model_len=max_model_len,
                           limit_mm_per_prompt={"timeseries": 50}, max_num_seqs=self.batch_size, **llm_kw)

    def wait_for_ready(self):
        return True                              # the engine is built synchronously in the constructor

    def _apply_chat_template(self, prompt):
        return (qwen_chat_prompt if self.qwen_template else chat_prompt)(prompt, self.system_prompt)

    def default_sampling_params(self):
        """the worker's own default (llm_utils.py:153): temperature 0.5, top_p 0.95, stop at <|endoftext|> / <|im_end|>"""
        from .llm import SamplingParams
        return SamplingParams(temperature=0.5, top_p=0.95, max_tokens=512, stop_token_ids=[151643, 151645])

    def llm_batch_generate(self, batch_prompts, batch_timeseries=None, dryrun_outputs=None, use_chat_template=True,
                           sampling_params=None):
        """-> one answer string per prompt, in prompt order; None for the prompts another replica owns (llm_utils.py:330-336
        returns None for unanswered positions)."""
        if batch_timeseries is not None and len(batch_prompts) != len(batch_timeseries):
            raise AssertionError(f"len(batch_prompts) != len(batch_timeseries): {len(batch_prompts)} != {len(batch_timeseries)}")
        mine = shard_indices(len(batch_prompts), self.replicas, self.replica)
        answers = [None] * len(batch_prompts)
        if dryrun_outputs is not None or self.engine == "dryrun":
            if dryrun_outputs is None:
                raise ValueError("the dryrun engine needs dryrun_outputs")
            for i in mine:
                answers[i] = dryrun_outputs[i]
            return answers
        sp = sampling_params or self.default_sampling_params()
        for b0 in range(0, len(mine), max(1, self.batch_size) * 4):          # a few slot-fulls per engine call keeps the batch full
            idxs = mine[b0:b0 + max(1, self.batch_size) * 4]
            reqs = []
            for i in idxs:
                text = self._apply_chat_template(batch_prompts[i]) if use_chat_template else batch_prompts[i]
                if batch_timeseries is not None and batch_timeseries[i] is not None:
                    reqs.append({"prompt": text, "multi_modal_data": {"timeseries": list(batch_timeseries[i])}})
                else:
                    reqs.append({"prompt": text})
            for i, out in zip(idxs, self.llm.generate(reqs, sp, use_tqdm=False)):
                answers[i] = out.outputs[0].text
        return answers

    def kill(self):
        self.llm = None


def answer_question_list_llm(client, question_list, ts_list, sampling_params=None):
    """inference_tsmllm_vllm.py:48-61 -> {idx: {"response": str}} for the questions this replica owns"""
    answers = client.llm_batch_generate(question_list, ts_list, sampling_params=sampling_params)
    return {i: {"response": a} for i, a in enumerate(answers) if a is not None}


def answer_question_list_hf(model, processor, question_list, ts_list, world=1, rank=0, batch_size=1, max_new_tokens=1024,
                            temperature=0.2, log=None, **gen_kw):
    """inference_tsmllm_deepspeed.py:62-118 on the HF-style surface: strided shard, processor(text=, timeseries=, padding=True),
    model.generate(max_length = prompt + 1024, temperature=0.2), decode the continuation, count tokens the reference's way.
    -> {idx: {"response", "num_tokens"}}"""
    patch = int(model.config.ts["patch_size"])
    local = shard_indices(len(question_list), world, rank)
    out = {}
    for b0 in range(0, len(local), batch_size):
        idxs = local[b0:b0 + batch_size]
        texts = [question_list[i] for i in idxs]
        flat, ts_tokens = [], []
        for i in idxs:
            series = ts_list[i] or []
            flat += [np.asarray(s, dtype=np.float64) for s in series]
            ts_tokens.append(sum(len(s) for s in series) // patch)
        inputs = processor(text=texts, timeseries=flat, padding=True, return_tensors="pt")
        n_in = inputs["input_ids"].shape[-1]
        seqs = model.generate(**inputs, max_length=n_in + max_new_tokens, temperature=temperature, **gen_kw)
        for j, i in enumerate(idxs):
            input_len = int(inputs["attention_mask"][j].sum().item())
            # generate() returns [left-padded prompt | continuation]: the continuation starts at column n_in for every row
            # (the reference slices at input_len, which equals n_in at its fixed batch size of 1)
            text = processor.tokenizer.decode([int(t) for t in seqs[j][n_in:]], skip_special_tokens=True)
            out[i] = {"response": text, "num_tokens": int(ts_tokens[j] + input_len)}
        if log:
            log(f"[worker {rank}] {len(out)}/{len(local)} finished.")
    return out


def run(model, dataset, exp, workdir=".", surface="llm", world=1, rank=0, tensor_parallel_size=1, max_tokens=512, temperature=0.2,
        batch_size=16, limit=None, llm=None, hf_model=None, hf_processor=None, log=print):
    """One driver run -> the answer file's path.  surface 'llm' = inference_tsmllm_vllm.py (answers in generated_answer.json, or
    generated_answer_<world>_<rank>.json when world > 1 replicas split the set); 'hf' = inference_tsmllm_deepspeed.py."""
    data = load_eval_dataset(dataset) if isinstance(dataset, str) else dataset
    if limit is not None:
        data = data[:limit]
    exp_dir = os.path.join(workdir, "exp", exp)
    os.makedirs(exp_dir, exist_ok=True)
    log(f"Experiment directory: {exp_dir}")
    ts_list = [rec.get("timeseries") for rec in data]
    if surface == "llm":
        from .llm import SamplingParams
        questions = [rec["question"] for rec in data]
        client = LLMClient(model_path=model, engine="vllm-ts", gpus_per_model=tensor_parallel_size, batch_size=batch_size,
                           replica=rank, replicas=world, llm=llm)
        client.wait_for_ready()
        ans = answer_question_list_llm(client, questions, [None if t is None else [np.asarray(s) for s in t] for t in ts_list],
                                       SamplingParams(max_tokens=max_tokens, temperature=temperature))
        client.kill()
        path = answer_file(exp_dir) if world == 1 else answer_file(exp_dir, world, rank)
    elif surface == "hf":
        questions = [chat_prompt(rec["question"]) for rec in data]
        if hf_model is None:
            from .modeling import ChatTSForCausalLM
            from .processing import ChatTSProcessor
            hf_model = ChatTSForCausalLM.from_pretrained(model)
            hf_processor = ChatTSProcessor.from_pretrained(model)
        ans = answer_question_list_hf(hf_model, hf_processor, questions, ts_list, world, rank, batch_size=1, temperature=temperature,
                                      max_new_tokens=max_tokens, log=log)
        path = answer_file(exp_dir, world, rank)
    else:
        raise ValueError(f"surface must be 'llm' or 'hf', got {surface!r}")
    write_answers(path, ans, questions)
    log(f"Results saved to {path}")
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(description="answer an evaluation set with the MI355X engine (generated_answer*.json)")
    ap.add_argument("--model", required=True, help="checkpoint directory or a preset name (synthetic weights)")
    ap.add_argument("--dataset", required=True)
    ap.add_argument("--exp", default="chatts_dataset_a")
    ap.add_argument("--workdir", default=".")
    ap.add_argument("--surface", choices=("llm", "hf"), default="llm")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--replicas", type=int, default=int(os.environ.get("CHATTS_REPLICAS", "1")))
    ap.add_argument("--replica", type=int, default=int(os.environ.get("CHATTS_REPLICA", "0")))
    ap.add_argument("--max-tokens", type=int, default=512)
    ap.add_argument("--temperature", type=float, default=0.2)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args(argv)
    run(a.model, a.dataset, a.exp, a.workdir, a.surface, a.replicas, a.replica, a.tensor_parallel_size, a.max_tokens, a.temperature,
        a.batch_size, a.limit)


if __name__ == "__main__":
    main()
