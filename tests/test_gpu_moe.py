"""Fused MoE GEMVs (include/zhilight_amd.h f3: zl_w4a16_moe_up / zl_w4a16_moe_down) against the oracle's restatement of
KERNEL_gemm_moe_up / KERNEL_gemm_moe_down (src/nn/quant/gptq/q_gemm_k_major.cu:243-390): bit-exact, like the decode GEMV they
are built from."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _t(a, dev, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dt is None else t.view(dt)


def _experts(oracle, rng, e, n, k, g):
    """e experts of an (n, k) matrix as k-major tensors: per-expert lists + (E, ...) stacks for the oracle"""
    kms = []
    for _ in range(e):
        qw, qz, sc = synth.gptq_hf(rng, k, (n + 7) // 8 * 8, g)
        kms.append(tuple(np.ascontiguousarray(a[:n]) for a in oracle.gptq_prepare_k_major(qw, qz, sc, g)))
    return kms, tuple(np.stack([km[i] for km in kms]) for i in range(3))


@pytest.mark.parametrize("m,top_k,n_shared,e,n_ff,k,g,ep", [(1, 2, 0, 4, 96, 1024, 128, None), (3, 4, 1, 6, 136, 2048, 128, None),
                                                         (2, 8, 0, 16, 768, 2048, 128, None), (4, 3, 2, 6, 64, 1152, 128, (2, 1)),
                                                         (5, 2, 0, 4, 40, 256, 64, (2, 0))])
def test_moe_up_and_down_bit_exact(oracle, dev, m, top_k, n_shared, e, n_ff, k, g, ep):
    from zhilight_amd import ops
    rng = np.random.default_rng(100 * m + top_k)
    et = e + n_shared                                    # routed experts stored locally + shared ones at the end of the stack
    exp_parallel, world, rank = (True, ep[0], ep[1]) if ep else (False, 1, 0)
    gate_l, gate_s = _experts(oracle, rng, et, n_ff, k, g)
    up_l, up_s = _experts(oracle, rng, et, n_ff, k, g)
    dim = 264                                            # down: (dim, kd) per expert; kd = n_ff rounded up to whole groups
    kd = (n_ff + g - 1) // g * g
    down_l, down_s = _experts(oracle, rng, et, dim, kd, g)
    # token -> expert ids: global ids under expert parallelism (local = id % world == rank, stored at id / world)
    n_global = e * world
    ids = np.stack([rng.choice(n_global, top_k, replace=False) for _ in range(m)]).astype(np.int32)
    wts = rng.random((m, top_k)).astype(np.float32)
    x = synth.act(rng, m, k)

    wu = ops.W4MoEWeight.from_k_major(
        [_t(np.concatenate([a[0], b[0]]).view(np.int32), dev) for a, b in zip(gate_l, up_l)],
        [_t(np.concatenate([a[1], b[1]]), dev) for a, b in zip(gate_l, up_l)],
        [_t(np.concatenate([a[2], b[2]]), dev, torch.float16) for a, b in zip(gate_l, up_l)], g, row_interleave=True)
    got_up = ops.moe_up(_t(x, dev), wu, _t(ids, dev), n_shared, exp_parallel, world, rank)
    ref_up = oracle.gptq_moe_up(oracle.h2u(x), gate_s, up_s, ids, n_shared, e, False, exp_parallel, world, rank)
    gu = got_up.view(torch.int16).cpu().numpy().view(np.uint16)
    assert gu.shape == ref_up.shape == (m, top_k + n_shared, n_ff)
    assert np.array_equal(gu, ref_up), int((gu != ref_up).sum())
    if ep:   # some expert was skipped somewhere: its rows are zero
        local = (ids % world) == rank
        assert (gu[:, :top_k][~local] == 0).all()

    wd = ops.W4MoEWeight.from_k_major([_t(a[0].view(np.int32), dev) for a in down_l], [_t(a[1], dev) for a in down_l],
                                      [_t(a[2], dev, torch.float16) for a in down_l], g)
    # the activations of the down projection (M, T, kd): the up outputs, zero-padded to whole groups
    a_in = np.zeros((m, top_k + n_shared, kd), np.float16)
    a_in[:, :, :n_ff] = oracle.u2h(ref_up)
    a_bits = oracle.h2u(a_in)
    got_dn = ops.moe_down(_t(a_in, dev), wd, _t(ids, dev), _t(wts, dev), n_shared, exp_parallel, world, rank)
    ref_dn = oracle.gptq_moe_down(a_bits, down_s, ids, wts, n_shared, e, False, exp_parallel, world, rank)
    gd = got_dn.view(torch.int16).cpu().numpy().view(np.uint16)
    assert np.array_equal(gd, ref_dn), int((gd != ref_dn).sum())
    # ADD_C: the output accumulates into what is there (the reference's shared-expert path, :389-390)
    base = synth.act(rng, m, dim)
    out = _t(base, dev).clone()
    ops.moe_down(_t(a_in, dev), wd, _t(ids, dev), _t(wts, dev), n_shared, exp_parallel, world, rank, out=out, add_c=True)
    ref_add = oracle.gptq_moe_down(a_bits, down_s, ids, wts, n_shared, e, False, exp_parallel, world, rank, add_c=oracle.h2u(base))
    assert np.array_equal(out.view(torch.int16).cpu().numpy().view(np.uint16), ref_add)


def test_moe_argument_checks(dev):
    from zhilight_amd import ops
    from zhilight_amd._lib import ZLError
    w = ops.W4Weight.random(64, 256, 128, dev, row_interleave=True)
    st = ops.W4MoEWeight(2, 64, 256, 128, torch.stack([w.qw, w.qw]), torch.stack([w.scales, w.scales]), torch.stack([w.zeros, w.zeros]), True)
    x = torch.zeros(1, 256, dtype=torch.float16, device=dev)
    ids = torch.zeros(1, 40, dtype=torch.int32, device=dev)
    with pytest.raises(ZLError):
        ops.moe_up(x, st, ids)                           # more than 32 experts per token
    with pytest.raises(ZLError):
        ops.moe_up(x.float(), st, ids[:, :2])
