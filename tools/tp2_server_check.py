#!/usr/bin/env python
"""GPU-box check of the tensor-parallel SERVER: two processes (TP=2) on the one GPU, rank 0 serves HTTP, rank 1 follows the
control plane (chatts_amd.engine.ControlPlane); a chat completion with a time series must return the tokens of a TP=1 engine
(= the oracle's, tests/test_gpu_server.py), blocking and streamed, two requests in flight.
    python tools/tp2_server_check.py            (writes gpurun_out/r2_tp2_server.json)"""
import json
import os
import signal
import subprocess
import sys
import threading
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PORT = 12399


def post(body, stream=False):
    req = urllib.request.Request(f"http://127.0.0.1:{PORT}/v1/chat/completions", data=json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=120) as r:
        raw = r.read().decode()
    if not stream:
        return json.loads(raw)
    chunks = [json.loads(l[6:]) for l in raw.splitlines() if l.startswith("data: ") and l != "data: [DONE]"]
    return "".join(c["choices"][0]["delta"].get("content", "") for c in chunks)


def main():
    import numpy as np
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CHATTS_FORCE_DEVICE="0", CHATTS_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "-m", "chatts_amd.server", "--model", "tiny-qwen3", "--tensor-parallel-size", "2",
           "--max-model-len", "512", "--max-num-seqs", "3", "--port", str(PORT), "--host", "127.0.0.1", "--seed", "3"]
    log = open(os.path.join(ROOT, "gpurun_out", "r2_tp2_server.log"), "w")
    srv = subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT, start_new_session=True)
    res = {"what": "OpenAI-compatible server under tensor parallelism: TP=2 as two processes on one GPU vs a TP=1 engine"}
    try:
        t0 = time.time()
        while True:
            try:
                urllib.request.urlopen(f"http://127.0.0.1:{PORT}/health", timeout=2).read()
                break
            except Exception:
                if srv.poll() is not None or time.time() - t0 > 240:
                    raise RuntimeError("server did not come up")
                time.sleep(1.0)
        res["startup_s"] = time.time() - t0
        rng = np.random.default_rng(11)
        bodies = []
        for lengths in ([64, 30], [100]):
            series = [(50 + 2 * np.cumsum(rng.standard_normal(L))).tolist() for L in lengths]
            text = f"I have {len(lengths)} time series. " + " ".join(f"TS{i} is of length {L}: <ts><ts/>;" for i, L in enumerate(lengths))
            bodies.append({"model": "chatts", "max_tokens": 10, "ignore_eos": True,
                           "messages": [{"role": "user", "content": [{"type": "text", "text": text}] + [{"timeseries": s} for s in series]}]})
        out = {}

        def one(i):
            out[i] = post(bodies[i])
        th = [threading.Thread(target=one, args=(i,)) for i in range(len(bodies))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        streamed = post(dict(bodies[0], stream=True), stream=True)
        res["tp2_tokens"] = [out[i]["token_ids"] for i in range(len(bodies))]
        res["streamed_equals_blocking"] = streamed == out[0]["choices"][0]["message"]["content"]
    finally:
        try:
            os.killpg(srv.pid, signal.SIGINT)          # exactly the process group this script started
            srv.wait(timeout=30)
        except Exception:
            try:
                os.killpg(srv.pid, signal.SIGKILL)
            except Exception:
                pass
        log.close()
    # TP=1 reference in this process
    from chatts_amd import server as srvmod
    from starlette.testclient import TestClient
    app = srvmod.build_server("tiny-qwen3", max_model_len=512, max_num_seqs=3, seed=3)
    try:
        c = TestClient(app)
        res["tp1_tokens"] = [c.post("/v1/chat/completions", json=b).json()["token_ids"] for b in bodies]
    finally:
        app.state.engine_thread.close()
    res["tokens_match"] = res["tp1_tokens"] == res["tp2_tokens"]
    res["passed"] = bool(res["tokens_match"] and res["streamed_equals_blocking"])
    with open(os.path.join(ROOT, "gpurun_out", "r2_tp2_server.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    return 0 if res["passed"] else 1


if __name__ == "__main__":
    sys.exit(main())
