"""How long the bounded waits of the one-shot all-reduce last: rank 0 of a 2-rank exchange whose peer never launches."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from zhilight_amd.parallel import OneShotAllReduce
dev = torch.device('cuda:0')
maxb = 1 << 20
addrs = [OneShotAllReduce.alloc(maxb)[0] for _ in range(2)]
ar = OneShotAllReduce(0, 2, addrs, maxb, dev)
for n in (4096, 262144):
    x = torch.randn(n, device=dev).half()
    torch.cuda.synchronize(); t0 = time.time()
    ar.all_reduce(x, out=torch.empty_like(x)); torch.cuda.synchronize()
    print(n, 'timeout after %.3f s' % (time.time() - t0), 'status', ar.status())
