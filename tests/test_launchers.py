"""The two ways N ranks come into being without the user typing torchrun:
`python bench.py --gpus N` re-executes itself under torch.distributed.run (the driver's call shape), and
`LLM(model, tensor_parallel_size=k)` spawns its followers like vLLM does for the reference
(NetManAIOps/ChatTS demo/demo_vllm.py:30, chatts/utils/llm_utils.py:154,251-266).  CPU: gloo, world size 2."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CHATTS_DIST_BACKEND"] = "gloo"
    return env


@pytest.mark.timeout(180)
def test_bench_gpus_n_launches_its_own_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_env(),
                         capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_summed"] == 2 and line["rccl_world_size"] == 2


@pytest.mark.timeout(180)
def test_bench_refuses_a_launcher_with_another_world_size():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=170)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


_SPAWN_SCRIPT = r"""
import sys, json
sys.path.insert(0, {root!r})
from chatts_amd.tp_spawn import TpGroup
from tests import tp_spawn_stub
group, obj = TpGroup.launch(2, tp_spawn_stub.build, ({out!r},), {{"scale": 2.0}}, use_cuda=False)
res = []
for tag, vals in (("a", [1, 2, 3]), ("boom", [1]), ("b", [10])):
    group.call("generate", vals, tag=tag)
    try:
        res.append(obj.generate(vals, tag=tag))
    except ValueError:
        res.append("raised")
group.shutdown()
print(json.dumps(res))
"""


@pytest.mark.timeout(180)
def test_spawned_followers_replay_the_leaders_calls(tmp_path):
    """leader in a plain process, one spawned follower: every announced call runs on both ranks (they meet in the all-reduce),
    an argument error on all ranks leaves the follower alive for the next call, shutdown joins it"""
    script = _SPAWN_SCRIPT.format(root=ROOT, out=str(tmp_path))
    out = subprocess.run([sys.executable, "-c", script], env=_env(), capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert res == [36.0, "raised", 60.0]                 # (1 + 2) ranks x scale 2 x sum
    for tag, want in (("a", 36.0), ("boom", 6.0), ("b", 60.0)):
        for r in (0, 1):
            assert float(open(tmp_path / f"rank{r}_{tag}.txt").read()) == want
