"""Worker of tests/test_gpu_comm.py: one rank of a multi-process one-shot all-reduce over hipIpc-mapped buffers (fixed
messages with / without the residual, then a run of messages of changing size with one rank held back before each).
usage: python _ar_worker.py <rank> <world> <exchange dir> <device index of this rank>"""
import os
import sys
import time

import torch

torch.set_num_threads(1)    # several workers x the default intra-op pool oversubscribe the host: parallel regions of the larger
                            # messages then stall a rank for longer than the kernel's bounded waits (0.6 s)

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
import random  # noqa: E402
rnd = random.Random(7)                      # the same sequence on every rank
sizes = [8, 4096, 6144, 8 * 4096, 3 * 4096 + 8, 32 * 4096, 64 * 4096, 2048]
skewed = [(rnd.choice(sizes), torch.float16, False, rnd.randrange(world)) for _ in range(60)]
fixed = [(4096, torch.float16, True, -1), (8, torch.float16, False, -1), (32 * 4096, torch.float16, True, -1),
         (4096 * 64, torch.bfloat16, True, -1), (4096, torch.float16, True, -1)] * 3
# cold-start skew between the processes (first use of an operator on a freshly paged-in box can stall one of them for
# seconds, longer than the bounded waits of the kernel): touch every host-side code path once, then meet
for (n, dtype, _, _) in fixed[:5]:
    a = inputs(0, 0, n, dtype)
    torch.equal((a.float() + a.float()).to(dtype).to(dev).cpu(), a)
torch.cuda.synchronize()
open(os.path.join(xdir, f"ready{rank}"), "w").close()
t0 = time.time()
while not all(os.path.exists(os.path.join(xdir, f"ready{r}")) for r in range(world)):
    if time.time() - t0 > 120:
        raise SystemExit("peers never became ready")
    time.sleep(0.005)
for msg, (n, dtype, with_res, slow) in enumerate(fixed + skewed):
    if slow == rank:
        time.sleep(0.003)                   # this rank enters the message late: its peers are one message ahead at most
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
    good = torch.equal(out.cpu(), want)
    if not good or (msg % 16 == 15 and ar.status() != 0):
        print("FIRST FAILURE rank", rank, "message", msg, "n", n, "slow", slow, "status", ar.status(), flush=True)
        ok = False
        break
print("RESULT", rank, "ok" if ok and ar.status() == 0 else f"FAIL status={ar.status()}", flush=True)
