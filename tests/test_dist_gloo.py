"""N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
 - tensor-parallel sharding of a GPTQ linear (column- and row-parallel) + all-gather / all-reduce gives
   the unsharded oracle result;
 - the bench contract's max-over-ranks timing and replica aggregation."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    import zl_oracle as zo
    from zhilight_amd import parallel
    rng = np.random.default_rng(0)          # same seed on every rank: identical full tensors
    k, n, g, m = 2048, 256, 128, 2
    qw, qz, sc = synth.gptq_hf(rng, k, n, g)
    km = [torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a.view(np.int16) if a.dtype == np.uint16 else a)
          for a in zo.gptq_prepare_k_major(qw, qz, sc, g)]
    x = synth.act(rng, m, k)
    full = zo.gptq_gemm_k_major(zo.h2u(x), *zo.gptq_prepare_k_major(qw, qz, sc, g))

    def np3(t3):
        return t3[0].numpy().view(np.uint32), t3[1].numpy(), t3[2].numpy().view(np.uint16)
    # column parallel: rows are independent -> all-gather reproduces the full result bit for bit
    col = parallel.shard_k_major(*km, g, "column", rank, world)
    y_col = torch.from_numpy(zo.u2h(zo.gptq_gemm_k_major(zo.h2u(x), *np3(col))).copy())   # fp16: gloo has no int16
    y_all = parallel.all_gather_columns(y_col).numpy().view(np.uint16)
    ok_col = bool(np.array_equal(y_all, full))
    # row parallel: fp16 partial outputs summed by the all-reduce (fp16 SUM, as NCCL does for 2 ranks)
    row = parallel.shard_k_major(*km, g, "row", rank, world)
    kw = k // world
    part = zo.gptq_gemm_k_major(zo.h2u(np.ascontiguousarray(x[:, rank * kw:(rank + 1) * kw])), *np3(row))
    t = torch.from_numpy(zo.u2h(part).copy())
    parallel.reduce_sum(t)
    exact = zo.gptq_gemm_k_major_exact(zo.h2u(x), *zo.gptq_prepare_k_major(qw, qz, sc, g))
    err_row = float(np.abs(t.numpy().astype(np.float64) - exact).max() / np.sqrt((exact ** 2).mean()))
    # bench contract
    elapsed = parallel.max_over_ranks(1.0 + rank)       # rank 1 is the slow one
    val = parallel.aggregate_throughput(64, elapsed, world)
    q.put((rank, ok_col, err_row, elapsed, val))
    dist.barrier()
    dist.destroy_process_group()


def test_tp_sharding_and_bench_aggregation_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_col, err_row, elapsed, val in res:
        assert ok_col, "column-parallel all-gather differs from the unsharded result"
        assert err_row < 6e-3, err_row          # fp16 partial sums: same noise level as the R kernel itself
        assert elapsed == 2.0 and val == 2 * 64 / 2.0


def test_shard_validation():
    from zhilight_amd import parallel
    qw = torch.zeros(8, 128 // 8 * 3, dtype=torch.int32)        # K = 384 = 3 groups
    qz = torch.zeros(8, 3, dtype=torch.uint8)
    sc = torch.zeros(8, 3, dtype=torch.float16)
    import pytest
    with pytest.raises(ValueError):
        parallel.shard_k_major(qw, qz, sc, 128, "row", 0, 2)    # 192 is not a multiple of the group size
    a = parallel.shard_k_major(qw, qz, sc, 128, "row", 2, 3)
    assert a[0].shape == (8, 16) and a[1].shape == (8, 1)
    b = parallel.shard_k_major(qw, qz, sc, 128, "column", 1, 4)
    assert b[0].shape == (2, 48)
