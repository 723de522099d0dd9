"""The tensor-parallel exchange step as a table of microseconds against message bytes: the one-shot peer-read all-reduce with fp16
rows (zl_ar_all_reduce) and with the rows as group-32 INT8 codes (zl_ar_all_reduce_int8 = ModelContext::reduce_tp_int8,
src/model/model_context.cpp:244-326, in one launch).  `world` ranks run as threads of THIS process on mutually concurrent streams
of device 0 (what a 1-GPU box can do: the peers' buffers are local HBM, so the table prices the kernels -- launches, flag
phases, quantisation arithmetic -- not the xGMI links; the bytes a rank would pull over the links are listed next to it).

    python tools/bench_allreduce.py [--world 2] [--reps 30]"""
import argparse
import json
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    from zhilight_amd.parallel import OneShotAllReduce
    import pytest
    from test_gpu_comm import _concurrent_streams
    dev = torch.device("cuda:0")
    world = a.world
    maxb = 8 << 20
    addrs = [OneShotAllReduce.alloc(maxb)[0] for _ in range(world)]
    ars = [OneShotAllReduce(r, world, addrs, maxb, dev) for r in range(world)]
    try:
        streams = _concurrent_streams(world, dev)
    except pytest.skip.Exception as e:      # noqa
        raise SystemExit("no concurrent streams: " + str(e))
    rows = []
    for nbytes in (8 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 8 << 20):
        n = nbytes // 2
        xs = [torch.randn(n, device=dev).half() for _ in range(world)]
        hid = [torch.zeros(n, device=dev, dtype=torch.float16) for _ in range(world)]
        rec = {"message_bytes_fp16": nbytes, "rows_of_8192": n // 8192}
        for mode in ("fp16", "int8"):
            # `reps` messages per rank captured in ONE hipGraph per rank (no host pacing inside the timed region); the ranks'
            # graphs are replayed concurrently
            times, errs, graphs = [0.0] * world, [], []
            for r in range(world):
                fn = ars[r].all_reduce_int8 if mode == "int8" else ars[r].all_reduce
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[r]):
                    for _ in range(a.reps):
                        fn(xs[r], residual=hid[r], out=hid[r])
                graphs.append(g)
            torch.cuda.synchronize()

            def run(r):
                try:
                    with torch.cuda.stream(streams[r]):
                        graphs[r].replay()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        graphs[r].replay()
                        e1.record()
                        streams[r].synchronize()
                        times[r] = e0.elapsed_time(e1) * 1e3 / a.reps
                except Exception as e:      # noqa
                    errs.append(e)
            th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=300)
            if errs or any(x.status() for x in ars):
                raise SystemExit("exchange failed: %r" % errs)
            rec["us_" + mode] = round(max(times), 2)
            del graphs
        # bytes a rank pulls over its links per message
        rec["link_bytes_fp16"] = (world - 1) * nbytes
        rec["link_bytes_int8"] = int(2 * (world - 1) / world * (n * (1 + 2 / 32)))
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    print("\nworld %d, device 0 only (peers' buffers are local HBM); us per message incl. the fused residual add" % world)
    print("%14s %10s %10s %18s %18s" % ("message bytes", "fp16 us", "int8 us", "link bytes fp16", "link bytes int8"))
    for r in rows:
        print("%14d %10.2f %10.2f %18d %18d" % (r["message_bytes_fp16"], r["us_fp16"], r["us_int8"], r["link_bytes_fp16"], r["link_bytes_int8"]))


if __name__ == "__main__":
    main()
