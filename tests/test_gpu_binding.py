"""The north star's drop-in claim at the level of the extension module: `zhilight.C` -- the reference's src/py_export/*.cpp (the pybind11 surface
zhilight.llama / zhilight.dynamic_batch import) and src/generator/batch_generator.cpp (the dynamic-batch scheduler), every unit compiled UNMODIFIED
from /root/reference by zhilight_amd.build.build_binding and linked on libzhilight_amd_host.so -- runs on the MI355X: C.Engine, C.LLaMA,
load_state_dict, C.BatchGenerator on its own thread, C.SearchTask, batch_search; the reference's own scheduler schedules, the boundary computes.
The generated tokens are compared with the CPU oracle's greedy continuation.  tests/_binding_worker.py is the child process (a scheduler thread
that stopped answering must not take the suite down); the module travels prebuilt, nothing here reads /root/reference at run time."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding(dev):
    from zhilight_amd import build
    if not os.path.exists(build.binding_target()):
        pytest.skip("zhilight.C was not built (no reference tree at build time)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_binding_worker.py")], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("BINDING_RESULT ")]
    assert lines, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    out = json.loads(lines[-1][len("BINDING_RESULT "):])
    assert not out["errors"], out["errors"]
    return out


def test_reference_binding_generates_the_oracles_greedy_tokens(binding):
    """one task through BatchGenerator.submit / SearchTask.get_result: prompt encode (the scheduler's chunking), bos / eos masking at the first step
    (scatter_update), log_softmax_bias + TopK per step (sampling_ops.hip), the generated tokens = the CPU oracle model's greedy continuation"""
    g = binding["greedy"]
    assert g["margin"] > 2e-3, g                # (the oracle's top-1 / top-2 gap at every step: far above the 1e-3 logit bar, so argmax is decided)
    assert g["got"] is not None and g["got"][-len(g["oracle"]):] == g["oracle"], g
    assert g["first_token_delay_ms"] is not None and g["first_token_delay_ms"] > 0


def test_reference_binding_dynamic_batch_of_three_tasks(binding):
    """batch_search with three prompts of 5 / 40 / 23 tokens: the scheduler admits them one prompt per step while the earlier ones decode
    (SearcherImplV1::batch_search), ragged KV buffers per task; every task's tokens = its own oracle continuation"""
    assert len(binding["batch"]) == 3
    for b in binding["batch"]:
        if b["margin"] > 2e-3:
            assert b["got"][-len(b["oracle"]):] == b["oracle"], b
        else:                                   # a near-tie in the oracle: the prefix before it must still agree
            assert b["got"][-len(b["oracle"])] == b["oracle"][0], b


def test_reference_binding_sampling_and_beam_search_run(binding):
    """the sampling path (top_p < 1: softmax -> random_sampler_gpu's host restatement on the counter-based generator) is reproducible for a seed and
    stays in the vocabulary; two beams score at least what greedy scored (beam search keeps the greedy hypothesis unless it finds a better one)"""
    s = binding["sampling"]
    assert s["draws"][0] is not None and s["draws"][0] == s["draws"][1], s
    assert all(0 <= t < s["vocab"] for t in s["draws"][0])
    b = binding["beam2"]
    assert b["got"] is not None and b["score"] >= b["greedy_score"] - 1e-3, b
