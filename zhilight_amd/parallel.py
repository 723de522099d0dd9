"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

The reference shards the path tensor-parallel: q/k/v and gate/up are column-parallel
(each rank keeps N/TP output rows of the k-major weight, src/nn/attention/attention.cpp:123-137,
src/nn/feedforward/feedforward.cpp:95-97), o_proj / down_proj are row-parallel (each rank keeps K/TP
input columns, group boundaries must divide: src/nn/linear/linear.cpp:1212-1234) followed by ONE
all-reduce SUM of the (M, dim_model) fp16 partial outputs (src/model/model_context.cpp:203-242).
Independent requests additionally shard as replicas (no collective), which is what bench.py runs.
"""
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
