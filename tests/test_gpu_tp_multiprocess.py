"""GPU: tensor parallelism with ONE PROCESS PER RANK, the product's shape (vLLM tensor_parallel_size=k: NetManAIOps/ChatTS
demo/demo_vllm.py:30) - eight rank processes share the box's one GPU, exchange hipIpc handles, map each other's exchange buffers and
run the real TP path at ChatTS-14B widths (tools/tp_parity_worker.py): IPC-mapped peer-to-peer exchange, the o_proj / down_proj
GEMVs carrying the exchange in their own launch, the (max, idx) token agreement, one captured hipGraph per rank.  Rank 0 checks the
gathered logits and tokens against the unsharded float32 oracle and the residual streams of all ranks against each other.
tests/test_gpu_tp_shards.py covers the same shard shapes without the transport (and config 4); tests/test_gpu_tp_p2p.py the
exchange kernels with two in-process ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(flow, world, tmp_path, extra_env=None):
    from chatts_amd.tp_spawn import free_port
    out = str(tmp_path / f"tp{world}_{flow}.json")
    env = dict(os.environ)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", CHATTS_FORCE_DEVICE="0", CHATTS_DIST_BACKEND="gloo", CHATTS_TP_FUSE_BLOCKS="48",
               CHATTS_TP_BULK_BLOCKS="16", CHATTS_TP_AR_BLOCKS="16", OMP_NUM_THREADS="16")      # (grid caps: W ranks' waiting launches must be resident together on ONE device)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tools", "tp_parity_worker.py"), "--flow", flow, "--out", out]
    for attempt in range(2):          # W processes must make progress side by side on one device; a stalled attempt is repeated once
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        if os.path.exists(out):
            res = json.load(open(out))
            if res["pass"] or not any(res["exchange_status_per_rank"]):
                return res, r
            os.remove(out)
        why = "\n".join(l for l in r.stderr.splitlines() if "tp_parity_worker rank" in l or "Error" in l)[-3000:]
        print(f"[tp multiprocess] attempt {attempt} rc={r.returncode}\n{r.stdout[-1500:]}\n{why}", flush=True)
    pytest.fail(f"tools/tp_parity_worker.py --flow {flow} did not produce a result (rc {r.returncode})")


@pytest.mark.parametrize("flow", ["headline", "config5"])
def test_tp8_one_process_per_rank_on_one_device_matches_oracle(flow, tmp_path):
    res, r = _run(flow, 8, tmp_path)
    assert res["tensor_parallel_size"] == 8 and res["processes"] == 8
    assert res["tokens_identical_on_all_ranks"] and res["residual_streams_bit_identical_on_all_ranks"]
    assert not any(res["exchange_status_per_rank"])
    for s, v in res["slots"].items():
        assert v["tokens_match"], (s, v["tokens_hip"], v["tokens_oracle"])
    assert res["max_logits_rel_err"] < 1e-3 and res["max_abs_err_over_max_logit"] < 1e-3
    assert res["max_logits_rel_err"] < 2e-4
    assert res["pass"]
    assert "tp_reduce" in res["exchange"] and res["decode_graph"]
