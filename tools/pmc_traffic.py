#!/usr/bin/env python
"""profiles/pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE passes (the `traffic` leg of bench.py's roofline objects).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fs -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --ttft-runs 1
    python tools/pmc_traffic.py headline /tmp/fs/.../p_results.db "<the command above>" profiles/r4_bench_pmc_fetch_size.txt
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fst -o p -- python tools/pmc_traffic.py ts-run
    python tools/pmc_traffic.py ts /tmp/fst/.../p_results.db "<command>" profiles/r4_ts_encoder_pmc_fetch_size.txt

Every entry records a digest of the kernel sources it was measured on (code_digest below); bench.py prints `traffic: null` when
the digest of the checked-out sources differs - a counter leg must describe the code it is quoted for.
gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream -> bytes = KiB x 1024 x 2 (MI355X_MICROARCH.md, HBM)."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PMC_TRAFFIC_OUT") or os.path.join(ROOT, "profiles", "pmc_traffic.json")      # (a GPU-box job writes under gpurun_out/)
GEMV_FILES = ["chatts_amd/csrc/gemv.hip", "chatts_amd/csrc/gemv_common.h", "chatts_amd/csrc/tp_common.h", "chatts_amd/csrc/common.h"]
TS_FILES = ["chatts_amd/csrc/gemm.hip", "chatts_amd/csrc/gemm_ring.hip", "chatts_amd/csrc/gemm_common.h", "chatts_amd/csrc/ts_frontend.hip",
            "chatts_amd/csrc/common.h"]
BATCHED_FILES = ["chatts_amd/csrc/gemm.hip", "chatts_amd/csrc/gemm_common.h", "chatts_amd/csrc/attention.hip", "chatts_amd/csrc/attn_decode.h",
                 "chatts_amd/csrc/common.h"]


def code_digest(files):
    h = hashlib.sha256()
    for f in files:
        h.update(f.encode())
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def counter_rows(db):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("""select kernel_name, grid_size, workgroup_size, count(*), avg(value), avg(duration) / 1000.0
                          from counters_collection where counter_name = 'FETCH_SIZE' and kernel_name like '%chatts%'
                          group by kernel_name, grid_size order by avg(value) desc""").fetchall()


def load():
    for path in (OUT, os.path.join(ROOT, "profiles", "pmc_traffic.json")):
        try:
            with open(path) as f:
                return json.load(f)
        except (OSError, ValueError):
            continue
    return {}


def save(d):
    with open(OUT, "w") as f:
        json.dump(d, f, indent=1)
        f.write("\n")


def headline(db, cmd, src):
    rows = [r for r in counter_rows(db) if "gemv_ldsx_kernel<2, 2, 3, true>" in r[0]]
    name, grid, wg, n, kib, us = max(rows, key=lambda r: r[4])            # gate_up: the SwiGLU GEMV with the most bytes
    inter = 13824
    d = load()
    d.update({"source": f"{src} ({cmd}, MI355X, round 6)", "code_digest": code_digest(GEMV_FILES), "code_files": GEMV_FILES,
              "kernel": f"gemv_ldsx_kernel<2,2,SWIGLU,NORM> (gate_up_proj, grid {grid} = {grid // wg} workgroups x {wg} threads)",
              "launches_counted": n, "fetch_size_kib_per_launch": round(kib, 1), "gfx950_fetch_correction": 2.0,
              "write_bytes_per_launch": inter * 4, "hbm_bytes_per_launch": int(kib * 1024 * 2 + inter * 4), "avg_us_under_pmc": round(us, 2)})
    save(d)
    print(json.dumps({k: d[k] for k in ("kernel", "fetch_size_kib_per_launch", "hbm_bytes_per_launch", "code_digest")}))


def ts(db, cmd, src):
    rows = counter_rows(db)
    calls = int(os.environ.get("TS_CALLS", "20")) + 1
    per_kernel, total = [], 0.0
    for name, grid, wg, n, kib, us in rows:
        if "fill_hash" in name:
            continue
        b = kib * 1024 * 2 * n / calls
        total += b
        per_kernel.append({"kernel": name.split("(")[0].replace("void chatts::", "")[:80], "grid": grid, "launches_per_call": round(n / calls, 2),
                           "fetch_bytes_per_call": int(b), "avg_us_under_pmc": round(us, 2)})
    d = load()
    d["ts_encoder"] = {"source": f"{src} ({cmd}, MI355X, round 6)", "code_digest": code_digest(TS_FILES), "code_files": TS_FILES,
                       "workload": "8 series x 256 steps (P = 128 patches), chatts_ts_encode", "calls_counted": calls,
                       "hbm_fetch_bytes_per_call": int(total), "algorithmic_bytes_per_call": 215257152,
                       "ratio": round(total / 215257152, 3), "per_kernel": per_kernel}
    save(d)
    print(json.dumps({k: d["ts_encoder"][k] for k in ("hbm_fetch_bytes_per_call", "ratio")}))
    for k in per_kernel:
        print(k)


def batched(db, cmd, src, key="14b_8x1024_fp8_b16"):
    rows = counter_rows(db)
    # gate_up: the per-layer weight stream with the most bytes (>= 40 launches: lm_head runs once per step and is larger)
    gu = max((r for r in rows if "gemm_stream_kernel" in r[0] and r[3] >= 40), key=lambda r: r[4])
    at = max((r for r in rows if "attn_decode_kernel" in r[0]), key=lambda r: r[4])
    d = load()
    inter, B = 13824, 16
    d.setdefault("batched", {})[key] = {
        "source": f"{src} ({cmd}, MI355X, round 6)", "code_digest": code_digest(BATCHED_FILES), "code_files": BATCHED_FILES,
        "kernel": f"{gu[0].split('(')[0].replace('void chatts::', '')} (gate_up_proj + SwiGLU, M = {B}, grid {gu[1]})",
        "fetch_size_kib_per_launch": round(gu[4], 1), "gfx950_fetch_correction": 2.0, "write_bytes_per_launch": B * inter * 2 * 2,
        "hbm_bytes_per_launch": int(gu[4] * 1024 * 2 + B * inter * 4), "algorithmic_bytes_per_launch": 143208448,
        "attn_decode_kernel": {"fetch_size_kib_per_launch": round(at[4], 1), "hbm_bytes_per_launch": int(at[4] * 1024 * 2), "avg_us_under_pmc": round(at[5], 2)}}
    save(d)
    print(json.dumps(d["batched"][key]))


def ts_run():
    """the TS encoder alone, N calls (what the ts pass profiles): bench.py's inputs, a 1-layer decoder to keep the build short"""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from chatts_amd import config as cfgmod
    from chatts_amd.modeling import ChatTSForCausalLM
    cfg = cfgmod.preset("chatts-14b", num_hidden_layers=1)
    proc, prompt, series, lengths = bench.build_inputs(cfg, 8, 256)
    inputs = proc(text=[prompt], timeseries=series, padding=True, return_tensors="pt")
    model = ChatTSForCausalLM.from_synthetic(cfg, seed=0, max_ctx=256, max_prefill_tokens=64)
    ser = inputs["timeseries"].cuda()
    model.get_multimodal_embeddings(timeseries=ser, valid_lengths=proc.last_lengths)
    for _ in range(int(os.environ.get("TS_CALLS", "20"))):
        model.ts_encoder.replay_last()
    torch.cuda.synchronize()


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "ts-run":
        ts_run()
    elif what == "headline":
        headline(*sys.argv[2:5])
    elif what == "ts":
        ts(*sys.argv[2:5])
    elif what == "batched":
        batched(*sys.argv[2:5])
    elif what == "digest":
        print(code_digest(GEMV_FILES), code_digest(TS_FILES), code_digest(BATCHED_FILES))
    else:
        raise SystemExit(__doc__)
