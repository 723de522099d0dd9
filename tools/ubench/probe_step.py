"""Timeline of the phase-kernel launches INSIDE a real batch-1 decode step under hipGraph replay (in-kernel 100 MHz
wall-clock stamps, probe build):  tools/ubench/variant.sh pprobe zhilight_amd/csrc/w4_phase.hip -DZL_PHASE_PROBE
usage: ZHILIGHT_AMD_SO=zhilight_amd/build/variants/libpprobe.so python tools/ubench/probe_step.py [layer]
Prints, for the three phase-kernel launches of one layer (qkv, o, gate|up) and the next layer's qkv, each stamp's
min / median / max over the waves relative to the first wave of the first of them."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zhilight_amd import _lib  # noqa: E402
from zhilight_amd.llama import LLaMA, ModelConfig, QuantConfig  # noqa: E402

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
cfg = ModelConfig.llama3_8b()
cfg.num_layers = int(os.environ.get("LAYERS", "32"))
model = LLaMA(cfg, QuantConfig(5, 128), dev).init_random(seed=1)
ctx = model.new_context(1, 1152, 1024, fill_random=True)
ctx.tokens.fill_(17)
model.step_greedy(ctx)
torch.cuda.synchronize()
L = _lib.lib()
seq0 = L.zl_debug_probe_seq()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    model.step_greedy(ctx)
per_layer = (L.zl_debug_probe_seq() - seq0) // cfg.num_layers
print("phase-kernel launches per layer:", per_layer)
probe = torch.zeros(4 * 16384 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
L.zl_debug_set_probe_p(C.c_void_p(probe.data_ptr()), C.c_int(seq0 + layer * per_layer))
g.replay()
torch.cuda.synchronize()
L.zl_debug_set_probe_p(C.c_void_p(0), C.c_int(0))
t = probe.cpu().numpy().reshape(4, 16384, 8).astype(np.float64)
t0 = t[0][t[0][:, 0] > 0][:, 0].min()
names = ["entry", "ring issued", "x staged+barrier", "first item done", "stream done", "parked+barrier", "end"]
for li in range(4):
    a = t[li]
    a = a[a[:, 0] > 0][:, :7]
    if not len(a):
        continue
    a = np.where(a > 0, (a - t0) * 10.0, np.nan)
    print(f"launch +{li}: {len(a)} waves, entry {np.nanmin(a[:, 0]) / 1e3:.2f} .. end {np.nanmax(a[:, 6]) / 1e3:.2f} us")
    for i, nm in enumerate(names):
        c = a[:, i]
        if np.all(np.isnan(c)):
            continue
        print(f"  {nm:18s} min {np.nanmin(c) / 1e3:7.2f}  p10 {np.nanpercentile(c, 10) / 1e3:7.2f}  median {np.nanmedian(c) / 1e3:7.2f}"
              f"  p90 {np.nanpercentile(c, 90) / 1e3:7.2f}  max {np.nanmax(c) / 1e3:7.2f} us")
