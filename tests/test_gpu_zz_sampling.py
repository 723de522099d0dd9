"""csrc/sampling_ops.hip -- the logit post-processing the reference's batch generator runs between decode steps (src/generator/beam_util.cu,
bmengine functions/{softmax,topk}.cu) -- against the numpy restatements in oracle/zl_oracle.py.  The reference reduces in fp32 over 1024 threads in
its own tree order; the bars are one rounding of T plus that reduction noise (1e-5 relative of a row's log-sum), positions and integer plumbing exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_CASES = [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7), (torch.float32, 2.0 ** -20)]


def _logits(rng, rows, n, dtype, dev):
    x = (rng.standard_normal((rows, n)) * 4.0).astype(np.float32)
    return torch.from_numpy(x).to(dtype).to(dev)


@pytest.mark.parametrize("dtype,ulp", _CASES)
@pytest.mark.parametrize("rows,n", [(1, 128256), (3, 1000), (8, 37)])
@pytest.mark.parametrize("temperature", [0.0, 0.7])
def test_log_softmax_bias(oracle, dev, dtype, ulp, rows, n, temperature):
    from zhilight_amd import ops
    rng = np.random.default_rng(rows * n)
    x = _logits(rng, rows, n, dtype, dev)
    bias = torch.from_numpy(rng.standard_normal(rows).astype(np.float32)).to(dev)
    got = ops.log_softmax_bias(x, bias, temperature).float().cpu().numpy().astype(np.float64)
    want = oracle.log_softmax_bias_ref(x.float().cpu().numpy(), bias.cpu().numpy(), temperature)
    assert (np.abs(got - want) <= ulp * np.abs(want) + 2e-5 * np.abs(want).max() + 1e-6).all(), float(np.abs(got - want).max())
    # the probabilities it encodes sum to one (bias removed)
    p = np.exp(got - bias.cpu().numpy().astype(np.float64).reshape(-1, 1)).sum(axis=1)
    assert np.allclose(p, 1.0, rtol=max(64 * ulp, 1e-3))


@pytest.mark.parametrize("dtype,ulp", _CASES)
def test_softmax_rows(oracle, dev, dtype, ulp):
    from zhilight_amd import ops
    rng = np.random.default_rng(5)
    x = _logits(rng, 4, 5000, dtype, dev)
    got = ops.softmax_rows(x, 0.8).float().cpu().numpy().astype(np.float64)
    want = oracle.softmax_rows_ref(x.float().cpu().numpy(), 0.8)
    assert (np.abs(got - want) <= ulp * want + 2e-5 * want.max(axis=1, keepdims=True) + 1e-12).all(), float(np.abs(got - want).max())


@pytest.mark.parametrize("dtype,ulp", _CASES)
@pytest.mark.parametrize("rows,n,top", [(1, 128256, 8), (5, 300, 16), (2, 64, 64), (3, 1000, 1)])
def test_topk_rows_values_positions_and_ties(oracle, dev, dtype, ulp, rows, n, top):
    from zhilight_amd import ops
    rng = np.random.default_rng(n + top)
    x = _logits(rng, rows, n, dtype, dev)
    x[:, n // 3] = x[:, n // 2]                                   # a tie inside every row: the lower index must come first
    x[0, 1] = float("-inf")
    v, i = ops.topk_rows(x, top)
    wv, wi = oracle.topk_rows_ref(x.float().cpu().numpy(), top)
    assert i.dtype == torch.int32 and np.array_equal(i.cpu().numpy(), wi)
    assert np.array_equal(v.float().cpu().numpy(), wv)             # the values are the row's own elements: bit-exact


def test_topk_rows_nan_never_wins(dev):
    from zhilight_amd import ops
    x = torch.tensor([[1.0, float("nan"), 3.0, 2.0]], dtype=torch.float16, device=dev)
    v, i = ops.topk_rows(x, 3)
    assert i.cpu().tolist() == [[2, 3, 0]] and v.cpu().tolist() == [[3.0, 2.0, 1.0]]


@pytest.mark.parametrize("dtype,ulp", _CASES)
def test_gather_scatter_and_penalties(oracle, dev, dtype, ulp):
    from zhilight_amd import ops
    rng = np.random.default_rng(9)
    rows, n = 4, 512
    x = _logits(rng, rows, n, dtype, dev)
    ref = x.float().cpu().numpy().astype(np.float64)
    rnd = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dtype).float().numpy().astype(np.float64)      # one rounding to T
    # gather: exact
    idx = torch.from_numpy(rng.integers(0, rows * n, 33).astype(np.int32)).to(dev)
    assert np.array_equal(ops.gather_logits(idx, x).cpu().numpy().astype(np.float64), ref.reshape(-1)[idx.cpu().numpy()])
    # scatter (set, then add): what SearcherImplV1::apply_repetition_penalty does to bos / eos at the first step
    tok = np.array([1, 2, 1, 2], np.int32)
    bid = np.array([0, 0, 3, 3], np.int32)
    val = np.array([-50000.0, -50000.0, 1.5, -2.25], np.float32)
    y = x.clone()
    ops.scatter_logits(torch.from_numpy(val).to(dev), torch.from_numpy(tok).to(dev), torch.from_numpy(bid).to(dev), y)
    want = ref.copy()
    want[bid, tok] = rnd(val)
    assert np.array_equal(y.float().cpu().numpy().astype(np.float64), want)
    ops.scatter_logits(torch.from_numpy(val).to(dev), torch.from_numpy(tok).to(dev), torch.from_numpy(bid).to(dev), y, add=True)
    want[bid, tok] = rnd(want[bid, tok] + rnd(val))
    assert np.array_equal(y.float().cpu().numpy().astype(np.float64), want)
    # repetition / presence penalties on distinct (row, token) pairs
    tok = rng.permutation(n)[:20].astype(np.int32)
    bid = rng.integers(0, rows, 20).astype(np.int32)
    fac = np.full(20, 1.3, np.float32)
    pres = np.where(np.arange(20) % 3 == 0, 0.5, 0.0).astype(np.float32)
    z = x.clone()
    ops.repetition_penalty(torch.from_numpy(fac).to(dev), torch.from_numpy(tok).to(dev), torch.from_numpy(bid).to(dev), z, presence=torch.from_numpy(pres).to(dev))
    want = oracle.repetition_penalty_ref(ref, rnd(fac), rnd(pres), tok, bid, rnd)
    assert np.array_equal(z.float().cpu().numpy().astype(np.float64), want)
