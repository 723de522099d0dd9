"""Worker of tests/test_gpu_comm.py: one rank of a two-process one-shot all-reduce over hipIpc-mapped buffers.
usage: python _ar_worker.py <rank> <world> <exchange dir> <device index of this rank>"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, xdir, devi = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
torch.cuda.set_device(devi)
dev = torch.device("cuda", devi)
from zhilight_amd.parallel import OneShotAllReduce  # noqa: E402

MAXB = 1 << 20
addr, _ = OneShotAllReduce.alloc(MAXB)
with open(os.path.join(xdir, f"h{rank}.tmp"), "wb") as fh:
    fh.write(OneShotAllReduce.export(addr))
os.rename(os.path.join(xdir, f"h{rank}.tmp"), os.path.join(xdir, f"h{rank}"))
bufs = []
for r in range(world):
    if r == rank:
        bufs.append(addr)
        continue
    path = os.path.join(xdir, f"h{r}")
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 60:
            raise SystemExit("peer handle never appeared")
        time.sleep(0.01)
    bufs.append(OneShotAllReduce.open(open(path, "rb").read()))
ar = OneShotAllReduce(rank, world, bufs, MAXB, dev)


def inputs(msg, r, n, dtype):
    g = torch.Generator().manual_seed(1000 * msg + r)
    return torch.randn(n, generator=g).to(dtype)


ok = True
for msg, (n, dtype, with_res) in enumerate([(4096, torch.float16, True), (8, torch.float16, False), (32 * 4096, torch.float16, True),
                                            (4096 * 64, torch.bfloat16, True), (4096, torch.float16, True)] * 3):
    xs = [inputs(msg, r, n, dtype) for r in range(world)]
    res = inputs(msg, 99, n, dtype)
    x = xs[rank].to(dev)
    out = torch.empty_like(x)
    ar.all_reduce(x, residual=res.to(dev) if with_res else None, out=out)
    tot = xs[0].float()
    for o in xs[1:]:
        tot = tot + o.float()
    want = tot.to(dtype)
    if with_res:
        want = (res.float() + want.float()).to(dtype)
    ok = ok and torch.equal(out.cpu(), want)
print("RESULT", rank, "ok" if ok and ar.status() == 0 else f"FAIL status={ar.status()}", flush=True)
