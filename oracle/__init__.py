"""CPU oracle for the ChatTS inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``chatts_amd/`` may import, call or
execute anything in this package.  The only legal importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` - and
there only as the checker, never as the thing measured or shipped.

What it is: a plain numpy / torch-CPU float32 restatement of the reference
algorithm for the path SURVEY.md section 8 names:

  sp_encoding.py   <- /root/reference/chatts/utils/encoding_utils.py:23-37,65-86
  ts_embedding.py  <- /root/reference/chatts/vllm/chatts_vllm.py:61-207
  protocol.py      <- /root/reference/chatts/vllm/chatts_vllm.py:369-444,564-574
  qwen_decoder.py  <- transformers 4.52.4/5.x models/qwen2/modeling_qwen2.py and
                      models/qwen3/modeling_qwen3.py (third-party dependency of the
                      reference, requirements.txt:7; NOT under /root/reference)
  synth.py         <- our own counter-hash weight definition (no reference twin)
  pipeline.py      <- the assembled greedy-generate path

Parity pinning: the reference ships no tests.  The oracle is
therefore pinned against the reference ITSELF, run in the build container:
``tests/golden/make_golden.py`` imports ``encoding_utils`` from /root/reference,
AST-slices ``TimeSeriesEmbedding`` out of ``chatts_vllm.py`` and runs stock
``transformers`` Qwen2/Qwen3 on CPU float32, and commits the resulting vectors
under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every oracle
function against them, plus the one known-answer the reference pins in a stored
notebook output (demo/demo_lora.ipynb:147: offset=6.0772, scaling=3.6917 ...).
"""
