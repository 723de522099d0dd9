"""Decode attention with the split merge INSIDE the launch (zl_decode_attn_la, attention.hip attn_tail_la): the last-arriving
workgroup of a (task, kv head) pair merges the pair's split records.  SURVEY 8a row a15; replaces
KERNEL_mqa_rag_buffer_split_kv + KERNEL_mqa_combine (src/nn/attention/attention_kernel.cu:729-923).

  * against the fp64 oracle (oracle.mqa_rag_buffer(exact=True)) over ragged lengths around the 32-key chunk / split boundaries,
    every split length the launcher may take (1 / 2 / 3 / 4 waves per workgroup), fp16 / bf16, both buffer layouts, NaN in the
    never-visible tail of V;
  * bit-identical to the two-launch path (zl_decode_attn: split kernel + merge kernel) at that path's split length;
  * the cross-workgroup hand-off itself: hundreds of back-to-back launches on ONE workspace (the arrival words must come back to
    zero every time), replayed from a hipGraph, next to a copy stream that keeps the memory system busy, every output word
    compared (MI355X_MICROARCH.md: "test every hand-off under uneven load, consumer L1-warm, checking every word").
"""
import numpy as np
import pytest
import torch

from test_gpu_ops import _bits, _make_kv, _t, _to_bits, _tt

pytestmark = pytest.mark.gpu

LENS = [64, 64, 64, 160, 160, 160, 1088, 1088, 640]
VALID = [1, 31, 33, 127, 128, 129, 1025, 517, 640]


def _la(ops, q, lens, valid, dk, dv, scale, hkv, dev, dtype, bshd=True, split_len=0, half=False, ws=None):
    b, h = q.shape[0], q.shape[-2]
    if ws is None:
        ws = ops.decode_attn_la_workspace(b, h, hkv, max(lens), dev)
    return ops.decode_attention_la(_tt(q, dev, dtype), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk), ops.make_ptr_table(dv),
                                   _t(np.array(valid, np.int32), dev), scale, max(lens), hkv, ws, bshd=bshd, split_len=split_len, half=half), ws


@pytest.mark.parametrize("h,hkv", [(32, 8), (32, 32), (16, 1), (28, 4)])
@pytest.mark.parametrize("bshd", [True, False])
@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("split_len", [0, 32, 64, 96, 128, 256, 288, 544, 1088])     # 1088: no split at all (the workgroup writes the rows itself)
def test_last_arriver_attention_against_the_oracle(oracle, dev, h, hkv, bshd, dtype, split_len):
    from zhilight_amd import ops
    rng = np.random.default_rng(330 + split_len)
    d, b = 128, len(LENS)
    kb, vb, dk, dv = _make_kv(rng, LENS, hkv, d, bshd, dev, dtype, oracle)
    tdt = torch.bfloat16 if dtype else torch.float16
    for bi, (L, v) in enumerate(zip(LENS, VALID)):       # poison (device copy only) what must never reach the result
        if v < L:
            pv = vb[bi].copy()
            if bshd:
                pv[v:] = np.uint16(0x7fc0 if dtype else 0x7e00)
            else:
                pv[:, v:] = np.uint16(0x7fc0 if dtype else 0x7e00)
            dv[bi].copy_(_t(pv.view(np.int16), dev, tdt))
    q = _to_bits(rng.standard_normal((b, 1, h, d)), dtype, oracle)
    scale = 1.0 / np.sqrt(d)
    mask = np.concatenate([(np.arange(L) < v).astype(np.int8) for L, v in zip(LENS, VALID)])
    exact = oracle.mqa_rag_buffer(q, np.array(LENS, np.int32), kb, vb, mask, hkv, scale, bshd, dtype=dtype, exact=True)
    rel = 5e-3 if dtype else 1e-3                        # bf16 output rounding is 2^-9 relative
    for half in ((False, True) if dtype == 0 else (False,)):
        got, ws = _la(ops, q, LENS, VALID, dk, dv, scale, hkv, dev, dtype, bshd, split_len, half)
        g = oracle.to_f32(_bits(got), dtype).astype(np.float64)
        assert np.isfinite(g).all()
        assert np.abs(g - exact).max() < rel * max(1.0, np.abs(exact).max()), (half, np.abs(g - exact).max())
        torch.cuda.synchronize()
        nb = b * hkv
        assert int(ws.view(torch.int32)[:nb].abs().sum()) == 0      # every pair's arrival word is back to zero


@pytest.mark.parametrize("b", [1, 3, 8])
def test_last_arriver_equals_the_two_launch_path_bitwise(oracle, dev, b):
    """fp32 records, the two-launch path's split length: same records, same merge arithmetic and order -> the same bits."""
    from zhilight_amd import ops
    rng = np.random.default_rng(77 + b)
    h, hkv, d = 32, 8, 128
    lens = [1088] * b
    valid = [1025, 700, 129, 1088, 1, 33, 1024, 513][:b]
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, True, dev, 0, oracle)
    q = _to_bits(rng.standard_normal((b, 1, h, d)), 0, oracle)
    scale = 1.0 / np.sqrt(d)
    two = ops.multi_query_attention_rag_buffer(_tt(q, dev, 0), _t(np.array(lens, np.int32), dev), ops.make_ptr_table(dk), ops.make_ptr_table(dv),
                                               None, scale, max(lens), hkv, valid_lens=_t(np.array(valid, np.int32), dev))
    sl = ops.decode_attn_split_len(b, hkv, max(lens))
    assert (max(lens) + sl - 1) // sl <= 16
    got, _ = _la(ops, q, lens, valid, dk, dv, scale, hkv, dev, 0, True, sl, False)
    assert np.array_equal(_bits(got), _bits(two))


@pytest.mark.parametrize("b,split_len", [(1, 32), (8, 64), (32, 128), (16, 0), (32, 0)])     # 0: the launcher's choice (8-wave workgroups, 2 / 1 splits)
def test_last_arriver_hand_off_under_load(oracle, dev, b, split_len):
    """300 launches back to back on one workspace, eager and replayed from a graph, with a copy stream hammering HBM next to
    them: every launch's output equals the first one's word for word, and the arrival words end at zero."""
    from zhilight_amd import ops
    rng = np.random.default_rng(5 + b)
    h, hkv, d = 32, 8, 128
    lens = [1088] * b
    valid = [int(v) for v in rng.integers(900, 1089, b)]
    kb, vb, dk, dv = _make_kv(rng, lens, hkv, d, True, dev, 0, oracle)
    q = _to_bits(rng.standard_normal((b, 1, h, d)), 0, oracle)
    scale = 1.0 / np.sqrt(d)
    qd, bl, vl = _tt(q, dev, 0), _t(np.array(lens, np.int32), dev), _t(np.array(valid, np.int32), dev)
    kt, vt = ops.make_ptr_table(dk), ops.make_ptr_table(dv)
    ws = ops.decode_attn_la_workspace(b, h, hkv, max(lens), dev)
    outs = [torch.empty_like(qd) for _ in range(4)]

    def launch(o):
        ops.decode_attention_la(qd, bl, kt, vt, vl, scale, max(lens), hkv, ws, out=o, split_len=split_len)
    launch(outs[0])
    torch.cuda.synchronize()
    first = outs[0].clone()
    mask = np.concatenate([(np.arange(L) < v).astype(np.int8) for L, v in zip(lens, valid)])
    exact = oracle.mqa_rag_buffer(q, np.array(lens, np.int32), kb, vb, mask, hkv, scale, True, dtype=0, exact=True)
    assert np.abs(oracle.to_f32(_bits(first), 0).astype(np.float64) - exact).max() < 1e-3 * max(1.0, np.abs(exact).max())
    # a copy stream that keeps the fabric busy (uneven load: the copies do not line up with the launches)
    side = torch.cuda.Stream()
    big_a, big_b = torch.empty(64 << 20, dtype=torch.uint8, device=dev), torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(40):
            big_b.copy_(big_a, non_blocking=True)
    bad = 0
    for it in range(300):
        o = outs[1 + it % 3]
        o.zero_()
        launch(o)
        if it % 25 == 24:
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                for _ in range(40):
                    big_b.copy_(big_a, non_blocking=True)
        bad += int((o != first).sum())
    torch.cuda.synchronize()
    assert bad == 0
    assert int(ws.view(torch.int32)[:b * hkv].abs().sum()) == 0
    # hipGraph replay of a chain of launches (what a captured decode step does, one per layer)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(8):
            launch(outs[1 + i % 3])
    for _ in range(30):
        for o in outs[1:]:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert all(bool((o == first).all()) for o in outs[1:])
    assert int(ws.view(torch.int32)[:b * hkv].abs().sum()) == 0
    del stop


def test_last_arriver_argument_checks(dev):
    from zhilight_amd import ops
    h, hkv, d = 32, 8, 128
    q = torch.zeros((1, 1, h, d), dtype=torch.float16, device=dev)
    kv = [torch.zeros((8192, hkv, d), dtype=torch.float16, device=dev)]
    bl = torch.tensor([8192], dtype=torch.int32, device=dev)
    ws = ops.decode_attn_la_workspace(1, h, hkv, 8192, dev)
    kt, vt = ops.make_ptr_table(kv), ops.make_ptr_table(kv)
    with pytest.raises(ops.ZLError):                      # 256 splits of 32 keys: more than the merge holds
        ops.decode_attention_la(q, bl, kt, vt, bl, 1.0, 8192, hkv, ws, split_len=32)
    with pytest.raises(ops.ZLError):                      # not a multiple of 32 keys
        ops.decode_attention_la(q, bl, kt, vt, bl, 1.0, 8192, hkv, ws, split_len=100)
    assert ops.decode_attn_la_split_len(1, hkv, 8192) % 32 == 0
    out = ops.decode_attention_la(q, bl, kt, vt, bl, 1.0, 8192, hkv, ws)     # the launcher's own choice fits
    torch.cuda.synchronize()
    assert out.shape == q.shape
