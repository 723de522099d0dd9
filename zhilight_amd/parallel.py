"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The reference shards the path tensor-parallel: q/k/v and gate/up are column-parallel
(each rank keeps N/TP output rows of the k-major weight, src/nn/attention/attention.cpp:123-137,
src/nn/feedforward/feedforward.cpp:95-97), o_proj / down_proj are row-parallel (each rank keeps K/TP
input columns, group boundaries must divide: src/nn/linear/linear.cpp:1212-1234) followed by ONE
all-reduce SUM of the (M, dim_model) fp16 partial outputs (src/model/model_context.cpp:203-242).
Independent requests additionally shard as replicas (no collective), which is what bench.py runs.
"""
import os

import torch
import torch.distributed as dist


class TPGroup:
    """Tensor-parallel group of this process: rank, size and the torch.distributed group (None = world).  The two
    collectives of the path are methods so that a test can stand in for RCCL (tests/test_gpu_model.py runs both
    ranks of a TP = 2 model on one GPU with a thread-barrier group)."""

    def __init__(self, rank=None, size=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.size = dist.get_world_size(group) if size is None else size

    def all_reduce_sum(self, t: torch.Tensor):
        return reduce_sum(t, self.group)

    def all_gather_columns(self, t: torch.Tensor):
        return all_gather_columns(t, self.group)


def shard_k_major(qweight, qzeros, scales, group_size, mode, rank, world):
    """Slice k-major GPTQ tensors (N,K/8) int32 / (N,K/G) uint8 / (N,K/G) fp16 for tensor parallelism.
    mode "column": output rows [rank*N/world, ...); mode "row": input columns, K/world must be a
    multiple of the group size (and of 8)."""
    n, k8 = qweight.shape
    k = k8 * 8
    if mode == "column":
        if n % world:
            raise ValueError("N not divisible by the TP degree")
        a, b = rank * n // world, (rank + 1) * n // world
        return qweight[a:b].contiguous(), qzeros[a:b].contiguous(), scales[a:b].contiguous()
    if mode == "row":
        if k % world or (k // world) % group_size or (k // world) % 8:
            raise ValueError("K/TP must be a multiple of the group size")
        kw = k // world
        return (qweight[:, rank * kw // 8:(rank + 1) * kw // 8].contiguous(),
                qzeros[:, rank * kw // group_size:(rank + 1) * kw // group_size].contiguous(),
                scales[:, rank * kw // group_size:(rank + 1) * kw // group_size].contiguous())
    raise ValueError(mode)


def reduce_sum(t: torch.Tensor, group=None):
    """ModelContext::reduce_sum: in-place all-reduce SUM of the fp16/bf16 partial outputs."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_columns(t: torch.Tensor, group=None):
    """Vocab- / column-parallel outputs -> full (M, N) (RawEmbedding::RowParallelImpl::projection)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=-1)


def max_over_ranks(value: float, device="cpu", group=None) -> float:
    """bench.py contract: the timed region's duration is the MAX over ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def aggregate_throughput(units_per_rank: float, elapsed_max: float, world: int) -> float:
    """Whole-job throughput of `world` independent replicas (weak scaling): all units / max time."""
    return world * units_per_rank / elapsed_max


# ----------------------------------------------------------------------------------------------------------------------
# Direct transports (include/zhilight_amd_comm.h): an RCCL communicator per GPU and the one-shot peer-read all-reduce.
# torch.distributed is only the BOOTSTRAP channel here (it ships the 128-byte unique id / the 64-byte IPC handles once).
# ----------------------------------------------------------------------------------------------------------------------
import ctypes as _C

_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2, torch.int32: 3, torch.int8: 4}


def _comm():
    from ._lib import comm_lib
    return comm_lib()


def _check(st, what):
    if st != 0:
        from ._lib import ZLError
        raise ZLError(f"{what}: status {st}" + (f" (ncclResult_t {st - 1000})" if st >= 1000 else ""))


def _stream():
    return _C.c_void_p(torch.cuda.current_stream().cuda_stream)


class RcclComm:
    """ncclCommInitRank on this process' current device (engine.cpp:140-157); collectives go to torch's CURRENT stream
    (c10d::NCCL*, 3rd/bmengine/bmengine/c10d/c10d.cpp:24-146).  `exchange(obj_or_None) -> obj` broadcasts rank 0's bytes."""

    def __init__(self, rank, size, broadcast_bytes):
        L = _comm()
        uid = (_C.c_char * 128)()
        if rank == 0:
            _check(L.zl_comm_unique_id(uid), "ncclGetUniqueId")
        raw = broadcast_bytes(bytes(uid) if rank == 0 else None)
        self._h = _C.c_void_p()
        _check(L.zl_comm_create(_C.byref(self._h), size, rank, _C.c_char_p(raw)), "ncclCommInitRank")
        self.rank, self.size = rank, size

    def all_reduce_sum(self, t, out=None):
        out = t if out is None else out
        _check(_comm().zl_comm_all_reduce_sum(self._h, _C.c_void_p(t.data_ptr()), _C.c_void_p(out.data_ptr()), _C.c_int64(t.numel()),
                                              _DT[t.dtype], _stream()), "ncclAllReduce")
        return out

    def all_gather(self, t):
        out = torch.empty((self.size,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        _check(_comm().zl_comm_all_gather(self._h, _C.c_void_p(t.data_ptr()), _C.c_void_p(out.data_ptr()), _C.c_int64(t.numel()),
                                          _DT[t.dtype], _stream()), "ncclAllGather")
        return out

    def reduce_scatter_sum(self, t):
        out = torch.empty((t.numel() // self.size,), dtype=t.dtype, device=t.device)
        _check(_comm().zl_comm_reduce_scatter_sum(self._h, _C.c_void_p(t.data_ptr()), _C.c_void_p(out.data_ptr()),
                                                  _C.c_int64(out.numel()), _DT[t.dtype], _stream()), "ncclReduceScatter")
        return out

    def send(self, t, peer):
        _check(_comm().zl_comm_send(self._h, _C.c_void_p(t.data_ptr()), _C.c_int64(t.numel()), _DT[t.dtype], peer, _stream()), "ncclSend")

    def recv(self, t, peer):
        _check(_comm().zl_comm_recv(self._h, _C.c_void_p(t.data_ptr()), _C.c_int64(t.numel()), _DT[t.dtype], peer, _stream()), "ncclRecv")

    @staticmethod
    def group_start():
        _check(_comm().zl_comm_group_start(), "ncclGroupStart")

    @staticmethod
    def group_end():
        _check(_comm().zl_comm_group_end(), "ncclGroupEnd")

    def broadcast(self, t, root=0):
        _check(_comm().zl_comm_broadcast(self._h, _C.c_void_p(t.data_ptr()), _C.c_int64(t.numel()), _DT[t.dtype], root, _stream()),
               "ncclBroadcast")
        return t

    def close(self):
        if self._h:
            _comm().zl_comm_destroy(self._h)
            self._h = _C.c_void_p()


class OneShotAllReduce:
    """zl_ar_*: every rank's partial rows published in a buffer the peers map, per-chunk flags pushed to the peers, rows read
    back over the direct links and summed in rank order (+ the layer's residual add), one launch per message.
    `buffers`: the `size` device addresses as THIS rank sees them (own buffer at index rank)."""

    def __init__(self, rank, size, buffers, max_message_bytes, device):
        L = _comm()
        self.rank, self.size, self.max_bytes = rank, size, max_message_bytes
        self.state = torch.zeros(int(L.zl_ar_state_bytes()), dtype=torch.uint8, device=device)
        arr = (_C.c_void_p * size)(*[_C.c_void_p(b) for b in buffers])
        _check(L.zl_ar_init(_C.c_void_p(self.state.data_ptr()), size, rank, arr, _C.c_int64(max_message_bytes), _stream()), "zl_ar_init")

    @staticmethod
    def alloc(max_message_bytes):
        """(address, bytes) of a zeroed fine-grained buffer peers can map"""
        L = _comm()
        nbytes = int(L.zl_ar_buffer_bytes(_C.c_int64(max_message_bytes)))
        p = _C.c_void_p()
        _check(L.zl_ar_alloc(_C.c_int64(nbytes), _C.byref(p)), "zl_ar_alloc")
        return p.value, nbytes

    @staticmethod
    def export(addr):
        h = (_C.c_char * 64)()
        _check(_comm().zl_ar_export(_C.c_void_p(addr), h), "hipIpcGetMemHandle")
        return bytes(h)

    @staticmethod
    def open(handle):
        p = _C.c_void_p()
        _check(_comm().zl_ar_open(_C.c_char_p(handle), _C.byref(p)), "hipIpcOpenMemHandle")
        return p.value

    def all_reduce(self, x, residual=None, out=None):
        """out = T(T(sum over ranks of x) + residual); x (.., n) fp16 / bf16 contiguous, n * rows * 2 <= max_message_bytes"""
        if x.numel() * 2 > self.max_bytes:
            from ._lib import ZLError
            raise ZLError("one-shot all-reduce: message larger than the exchange buffers")
        out = x if out is None else out
        _check(_comm().zl_ar_all_reduce(_C.c_void_p(self.state.data_ptr()), _C.c_void_p(x.data_ptr()),
                                        _C.c_void_p(residual.data_ptr()) if residual is not None else None, _C.c_void_p(out.data_ptr()),
                                        _C.c_int64(x.numel()), _DT[x.dtype], _stream()), "zl_ar_all_reduce")
        return out

    def all_reduce_int8(self, x, residual=None, out=None):
        """the same sum with the rows travelling as group-32 int8 codes (ModelContext::reduce_tp_int8 as one launch):
        out = T(dequantised reduce) (+ residual in T arithmetic); numel % (64 * world) == 0"""
        if x.numel() * 2 > self.max_bytes:
            from ._lib import ZLError
            raise ZLError("one-shot all-reduce: message larger than the exchange buffers")
        out = torch.empty_like(x) if out is None else out
        _check(_comm().zl_ar_all_reduce_int8(_C.c_void_p(self.state.data_ptr()), _C.c_void_p(x.data_ptr()),
                                             _C.c_void_p(residual.data_ptr()) if residual is not None else None, _C.c_void_p(out.data_ptr()),
                                             _C.c_int64(x.numel()), _C.c_int(self.size), _DT[x.dtype], _stream()), "zl_ar_all_reduce_int8")
        return out

    def status(self):
        return int(_comm().zl_ar_status(_C.c_void_p(self.state.data_ptr()), _stream()))


class DirectTPGroup(TPGroup):
    """TPGroup on the direct transports: RCCL communicator for the bandwidth-bound messages, the one-shot all-reduce (with the
    residual add fused) up to `oneshot_bytes`.  Bootstrapped over an existing torch.distributed group (gloo is enough)."""

    def __init__(self, group=None, oneshot_bytes=1 << 20, device=None, rccl=True):
        """rccl=False: the one-shot exchange only (every collective of the decode step fits it; what two ranks sharing ONE GPU
        can run -- RCCL refuses duplicate devices): larger messages then raise."""
        super().__init__(group=group)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        src = dist.get_global_rank(group, 0) if group is not None else 0     # `src` of broadcast_object_list is a GLOBAL rank

        def bcast(b):
            box = [b]
            dist.broadcast_object_list(box, src=src, group=group)
            return box[0]
        self.comm = RcclComm(self.rank, self.size, bcast) if rccl else None
        self.rccl_ranks = self.size if rccl else 0
        addr, _ = OneShotAllReduce.alloc(oneshot_bytes)
        self._own_buffer = addr
        handles = [None] * self.size
        dist.all_gather_object(handles, OneShotAllReduce.export(addr), group=group)
        bufs = [addr if r == self.rank else OneShotAllReduce.open(handles[r]) for r in range(self.size)]
        self.oneshot = OneShotAllReduce(self.rank, self.size, bufs, oneshot_bytes, dev)
        self.oneshot_bytes = oneshot_bytes

    def _need_rccl(self, what):
        if self.comm is None:
            from ._lib import ZLError
            raise ZLError(f"DirectTPGroup(rccl=False): {what} does not fit the one-shot exchange")
        return self.comm

    def all_reduce_sum(self, t):
        if t.numel() * 2 <= self.oneshot_bytes and t.numel() % 8 == 0 and t.dtype in (torch.float16, torch.bfloat16):
            return self.oneshot.all_reduce(t)
        return self._need_rccl("this all-reduce").all_reduce_sum(t)

    def check(self):
        """raise if any one-shot exchange since the last check timed out (its output rows are NaN-poisoned by the kernel, so a
        missed check cannot turn into silently wrong sums); synchronises the stream -- call between steps, not under capture"""
        n = self.oneshot.status()
        if n:
            from ._lib import ZLError
            raise ZLError(f"one-shot all-reduce: {n} bounded waits expired (a peer stalled for more than the kernel's poll budget)")

    def all_reduce_add(self, part, hidden):
        """hidden <- hidden + sum over ranks of part, in place (block.cpp:123-140); fused into the one-shot launch when it fits"""
        if part.numel() * 2 <= self.oneshot_bytes and part.numel() % 8 == 0 and part.dtype in (torch.float16, torch.bfloat16):
            return self.oneshot.all_reduce(part, residual=hidden, out=hidden)
        return None

    def reduce_tp_int8(self, data):
        """ModelContext::reduce_tp_int8 (src/model/model_context.cpp:244-326; the reference switches to it above
        REDUCE_TP_INT8_THRES rows): the sum over the ranks with every transfer as group-32 int8 codes + T scales -- 1.06 bytes
        per value and hop instead of 2.  Step by step as the reference: quantise all WS slices; send slice d to rank d and
        receive the WS - 1 peers' codes of MY slice (rank distance order); add my own unquantised slice, re-quantise; all-gather
        the re-quantised slices; dequantise.  data (rows, n) T with rows * n % (32 * WS) == 0; returns a new tensor."""
        from . import ops
        if (getattr(self, "oneshot", None) is not None and data.numel() * 2 <= self.oneshot_bytes and data.numel() % (64 * self.size) == 0 and data.is_contiguous() and data.data_ptr() % 8 == 0
                and data.dtype in (torch.float16, torch.bfloat16) and os.environ.get("ZL_REDUCE_INT8_ONESHOT", "1") != "0"):
            return self.oneshot.all_reduce_int8(data)           # the five steps as one launch, same bits (csrc_comm/comm.hip: k_ar_q8)
        comm = self._need_rccl("reduce_tp_int8")
        ws, rank = self.size, self.rank
        n = data.numel()
        if n % (32 * ws):
            raise ValueError("reduce_tp_int8: numel must be a multiple of 32 x world size")
        m = n // ws // 32
        flat = data.contiguous().view(ws, m, 32)
        q_send, s_send = ops.quant_group_32(flat)
        s_send = s_send.view(ws, m)
        q_recv = torch.empty(ws - 1, m, 32, dtype=torch.int8, device=data.device)
        s_recv = torch.empty(ws - 1, m, dtype=data.dtype, device=data.device)
        comm.group_start()
        for i in range(ws - 1):
            src, dst = (rank + i + 1) % ws, (rank - i - 1) % ws
            comm.send(q_send[dst], dst)
            comm.recv(q_recv[i], src)
            comm.send(s_send[dst], dst)
            comm.recv(s_recv[i], src)
        comm.group_end()
        q_sum = torch.empty(ws, m, 32, dtype=torch.int8, device=data.device)
        s_sum = torch.empty(ws, m, dtype=data.dtype, device=data.device)
        q_mine, s_mine = ops.dequant_sum_quant_g32(flat[rank], q_recv, s_recv)
        q_sum[rank].copy_(q_mine)
        s_sum[rank].copy_(s_mine)
        comm.group_start()
        for i in range(ws - 1):
            dst, src = (rank + i + 1) % ws, (rank - i - 1) % ws
            comm.send(q_sum[rank], dst)
            comm.recv(q_sum[src], src)
            comm.send(s_sum[rank], dst)
            comm.recv(s_sum[src], src)
        comm.group_end()
        return ops.dequant_group_32(q_sum, s_sum.view(-1)).view(data.shape)

    def all_gather_columns(self, t):
        if self.comm is None:
            # one-shot only: a gather is the sum of the ranks' zero-padded slices (adding zeros is exact)
            rows, n = t.shape[0], t.shape[-1]
            wide = torch.zeros(rows, self.size * n, dtype=t.dtype, device=t.device)
            wide[:, self.rank * n:(self.rank + 1) * n] = t
            if wide.numel() * 2 > self.oneshot_bytes or wide.numel() % 8:
                self._need_rccl("this all-gather")
            return self.oneshot.all_reduce(wide)
        parts = self.comm.all_gather(t.contiguous())
        return torch.cat(list(parts), dim=-1)

    def close(self):
        """release the exchange buffer (peers must have stopped using it) and the communicator"""
        if getattr(self, "_own_buffer", None):
            _comm().zl_ar_free(_C.c_void_p(self._own_buffer))
            self._own_buffer = None
