"""Config 5's MoE feed-forward over FP8 128x128-block experts: the host flow of FeedForward's dispatch route
(reference src/nn/feedforward/feedforward.cpp:417-482 route, :599-629 sort_token, :1045-1072 get_grouped_input_gpu, :1075-1150
forward_gpu_dispatch), strung together from the C-ABI launchers of zhilight_amd/ops.py:

    logits = router(x)                                   functions::Gemm                       -> zl_gemm_nt
    ids, weights, loads = top-k / group-limited top-k    top_k_softmax / group_topk_softmax    -> zl_moe_top_k_softmax / zl_moe_group_topk
    m_indices, padded positions, total                   fill_m_indices_padded_indices         -> zl_moe_fill_m_indices
    order = sort of the (token, slot) pairs by expert    functions::arange, sort_pair_1d (CUB) -> zl_arange_i32, zl_sort_pairs_i32 (stable), zl_divide_i32
    rev = position of a pair inside its expert's run     calc_reverse_idx                      -> zl_moe_calc_reverse_idx
    grouped input: per-token 1x128 cast, rows and scales scattered to the 64-aligned runs     -> zl_fp8_per_token_cast, zl_scatter_update_dim0
    w0, w1 = grouped GEMMs (in, gated); w0 = act(w0) * w1; w2 = grouped GEMM (out)            -> zl_fp8_block_gemm_group, zl_gate_mul
    y[token] = sum_slot weight * w2[run(expert) + rev]   sum_experts (per-expert inputs)       -> zl_moe_sum_experts_arr
    y += shared_expert(x)                                with_share (:483-492)                 -> zl_fp8_block_gemm_group x3, zl_element_add_scale

Expert parallel (MOE_EXP_PARALLEL=1 in the reference, feedforward.cpp:251-305): rank r of `world_size` holds the experts e with
e % world_size == r (local index e // world_size); every rank routes ALL tokens (the hidden rows are replicated), sorts the (token, slot)
pairs by (rank, expert) -- plus_for_sort -- takes ITS slice of the order, runs its grouped GEMMs and combines only its experts' rows;
forward() then returns this rank's PARTIAL sum, which the layer's ordinary all-reduce (ModelContext::reduce_sum, block.cpp:123-140)
completes.  A static shared expert is tensor-parallel in that mode (its dim_ff sharded: the caller passes the rank's shard), so its partial
rides the same all-reduce.  Load-balanced shared experts (MOE_DYN_SHARED=1, feedforward.cpp:268-276, :459-462): every rank keeps a full
copy of the n shared experts BEHIND its routed ones (pseudo expert ids (experts_local + s) * world_size + rank); the router writes
top_k + n slots per token, the extra ones with weight 1, and route_shared_lb hands each of them to the first rank that still has spare
capacity -- from there on they are ordinary experts of the grouped flow.  Since round 4 the sort and the row scatters are launchers of the C ABI as well
(functions::arange / sort_pair_1d / divide / scatter_update_dim0); what is left to the framework is the scale transpose and
slicing views -- there is no CPU or torch fallback for any arithmetic or index step."""
from typing import Optional

import os

import torch

from . import ops


class Fp8BlockMoE:
    """experts: stacked FP8 block weights -- w_in / w_gated (E, dim_ff, dim_model) uint8 with scales (E, ceil(dim_ff / 128), dim_model / 128)
    fp32, w_out (E, dim_model, dim_ff) with scales (E, ceil(dim_model / 128), dim_ff / 128); router (E, dim_model) in the activation dtype."""

    def __init__(self, router, w_in, s_in, w_gated, s_gated, w_out, s_out, top_k, norm_topk_prob=True, routed_scaling_factor=1.0,
                 scoring_func="softmax", n_group=1, topk_group=1, e_score_correction_bias: Optional[torch.Tensor] = None, act="silu",
                 block_m=64, shared=None, world_size=1, rank=0, dyn_shared=0):
        """shared: None or (w_in (ff_s, dim), s_in, w_gated, s_gated, w_out (dim, ff_s), s_out) of the always-on shared expert.
        world_size > 1: expert parallel -- the stacked weights hold THIS rank's experts (global e = local * world_size + rank), the router
        stays global (E = experts_local * world_size rows).
        dyn_shared = n > 0: the LAST n experts of the stack are this rank's copies of the n load-balanced shared experts (the router then has
        (experts_local - n) * world_size rows); excludes a static `shared`"""
        self.world_size, self.rank = int(world_size), int(rank)
        if self.world_size < 1 or not 0 <= self.rank < self.world_size or self.world_size & (self.world_size - 1):
            raise ops.ZLError("Fp8BlockMoE: world_size is a power of two (the combine masks expert ids with world_size - 1, as the "
                              "reference does) and 0 <= rank < world_size")
        self.router, self.top_k = router, top_k
        self.w_in, self.s_in, self.w_gated, self.s_gated, self.w_out, self.s_out = w_in, s_in, w_gated, s_gated, w_out, s_out
        self.dyn_shared = int(dyn_shared)
        self.num_experts = w_in.shape[0] * self.world_size                  # global, pseudo ids of the balanced shared experts included
        self.num_routed = self.num_experts - self.dyn_shared * self.world_size
        if self.dyn_shared < 0 or self.num_routed <= 0 or (self.dyn_shared and shared is not None):
            raise ops.ZLError("Fp8BlockMoE: dyn_shared experts sit behind at least one routed expert and exclude a static shared expert")
        if router.shape[0] != self.num_routed:
            raise ops.ZLError("Fp8BlockMoE: the router has one row per GLOBAL routed expert")
        self.norm_topk_prob, self.routed_scaling_factor, self.scoring_func = norm_topk_prob, routed_scaling_factor, scoring_func
        self.n_group, self.topk_group, self.bias, self.act, self.block_m = n_group, topk_group, e_score_correction_bias, act, block_m
        self.shared = shared
        # the grouped GEMMs and the shared expert read ZLF8M-packed copies of the codes (1 KiB contiguous fragment loads: DeepSeek-V3's
        # expert gate|up 9.7 -> 7.9 us at one row, 12.1 -> 10.0 at 32; the same bits).  A second copy: ZL_FP8_PACKED=0 keeps only the
        # row-major one (which the per-token composition of the tests reads either way)
        self.packed = None
        if os.environ.get("ZL_FP8_PACKED", "1") != "0":
            self.packed = tuple(ops.Fp8BlockMWeight(w) for w in (w_in, w_gated, w_out))
            if shared is not None:
                self.shared_packed = tuple(ops.Fp8BlockMWeight(shared[i]) for i in (0, 2, 4)) + (shared,)
        if w_in.shape != w_gated.shape or w_out.shape[1] != w_in.shape[2] or w_out.shape[2] != w_in.shape[1]:
            raise ops.ZLError("Fp8BlockMoE: expert weight shapes do not form in / gated / out projections")

    def route(self, x):
        """(ids (T, k) int32, weights (T, k) fp32, all_loads (E + 1,) int32: tokens per expert | per rank) -- FeedForward::route"""
        logits = ops.gemm_nt_f32(x, self.router)                               # fp32 logits, as the reference's router Linear (set_output_type(kFloat))
        all_loads = torch.zeros(self.num_experts + self.world_size, dtype=torch.int32, device=x.device)
        expert_load, worker_load = all_loads[:self.num_experts], all_loads[self.num_experts:]
        ext = self.top_k + self.dyn_shared
        if self.topk_group > 1:
            w, ids = ops.moe_group_topk(logits, self.bias, self.n_group, self.topk_group, self.top_k, top_k_ext=ext, norm_topk_prob=self.norm_topk_prob,
                                        weight_scale=self.routed_scaling_factor, scoring_func=self.scoring_func, worker_load=worker_load,
                                        expert_load=expert_load, num_worker=self.world_size)
        else:
            w, ids = ops.moe_top_k_softmax(logits, self.top_k, top_k_ext=ext, norm_topk_prob=self.norm_topk_prob, weight_scale=self.routed_scaling_factor,
                                           scoring_func=self.scoring_func, worker_load=worker_load, expert_load=expert_load, num_worker=self.world_size)
        if self.dyn_shared:                                                    # the extra slots: the first rank with spare capacity (route_shared_lb)
            ops.moe_route_shared_lb(ids, w, worker_load, expert_load, self.top_k, self.num_routed // self.world_size)
        return ids, w, all_loads

    def with_share(self, x, ret):
        """FeedForward::with_share (feedforward.cpp:483-492): ret + shared_expert(x) when there is a static shared expert"""
        if self.shared is None:
            return ret
        w_in, s_in, w_gated, s_gated, w_out, s_out = self.shared
        if self.packed is not None and x.shape[0] <= 32:                       # (the packed entry point: up to 32 rows per launch)
            sp = getattr(self, "shared_packed", None)
            if sp is None or sp[3] is not self.shared:                        # (a shared expert attached or replaced after construction)
                sp = self.shared_packed = tuple(ops.Fp8BlockMWeight(self.shared[i]) for i in (0, 2, 4)) + (self.shared,)
            w_in, w_gated, w_out = sp[:3]
        h0 = ops.fp8_block_linear(x, w_in, s_in)
        h1 = ops.fp8_block_linear(x, w_gated, s_gated)
        ops.gate_mul(h0, h1, self.act)
        return ops.element_add_scale(ret, ops.fp8_block_linear(h0, w_out, s_out), 1.0)

    def forward(self, x):
        """x (T, dim_model) fp16 / bf16 -> (T, dim_model): FeedForward::forward_gpu_dispatch"""
        if x.dim() != 2 or x.dtype not in (torch.float16, torch.bfloat16):
            raise ops.ZLError("Fp8BlockMoE: (tokens, dim_model) half or bfloat16 rows")
        tokens, dim = x.shape
        e, k = self.num_experts, self.top_k + self.dyn_shared              # (num_experts_may_share, top_k_may_share)
        ids, weights, all_loads_t = self.route(x)
        all_loads = all_loads_t.cpu().tolist()                               # (the reference's to_vector: the one host sync of the flow)
        ws, rk = self.world_size, self.rank
        ep = ws > 1
        m_indices, padded_idx, total = ops.moe_fill_m_indices_padded_indices(all_loads, self.block_m, e, x.device, exp_parallel=ep, rank=rk,
                                                                            world_size=ws)
        if total == 0:                                                        # (EP: none of this rank's experts was picked)
            return self.with_share(x, torch.zeros_like(x))
        # (token, slot) pairs sorted by expert, stable: sorted position j holds pair order[j]; its token is order[j] // k
        # FeedForward::sort_token (feedforward.cpp:599-629): arange, stable sort_pair_1d by expert id -- by (rank, expert id) under EP:
        # plus_for_sort -- then THIS rank's slice of the order, divided by top_k; kernels since round 4
        flat = ids.reshape(-1).contiguous()
        keys = ops.moe_plus_for_sort(flat, e, ws) if ep else flat
        _, order = ops.sort_pairs_i32(keys, ops.arange_i32(flat.numel(), x.device), max_key=e * (1 + ws))
        rev = ops.moe_calc_reverse_idx(ids, order, all_loads, e, world_size=ws, sorted_by_rank=ep)
        if ep:
            local_start, rank_load = sum(all_loads[e:e + rk]), all_loads[e + rk]
            order = order[local_start:local_start + rank_load].contiguous()
        sorted_tokens = ops.divide_i32(order, k)
        # grouped input (get_grouped_input_gpu, :1040-1075): codes and 1x128 scales of the sorted tokens scattered to the 64-aligned
        # positions (scatter_update_dim0), padding rows zero
        a8, sa = ops.fp8_per_token_cast(x, scale_col_major=False)
        g8 = torch.zeros((total, dim), dtype=torch.uint8, device=x.device)
        ops.scatter_update_dim0(g8, padded_idx, a8, sorted_tokens)
        gs = torch.zeros((total, dim // 128), dtype=torch.float32, device=x.device)
        ops.scatter_update_dim0(gs, padded_idx, sa[:tokens].contiguous(), sorted_tokens)
        gs_t = gs.t().contiguous()                                            # (dim / 128, total): column-major scales, aligned_m = total
        w_in, w_gated, w_out = self.packed if self.packed is not None else (self.w_in, self.w_gated, self.w_out)
        w0 = ops.fp8_block_gemm(g8, gs_t, w_in, self.s_in, m_indices=m_indices, dtype=x.dtype)
        w1 = ops.fp8_block_gemm(g8, gs_t, w_gated, self.s_gated, m_indices=m_indices, dtype=x.dtype)
        ops.gate_mul(w0, w1, self.act)
        b8, sb = ops.fp8_per_token_cast(w0)                                   # Fp8Block::quant_input of the grouped rows (total % 4 == 0)
        w2 = ops.fp8_block_gemm(b8, sb, w_out, self.s_out, m_indices=m_indices, dtype=x.dtype)
        # each expert's run of w2 (global expert order, 64-aligned starts), then the weighted combine
        # (EP: only this rank's experts have rows; sum_experts skips the others, so the result is the rank's partial)
        parts, off = [], 0
        for exp in range(e):
            n = all_loads[exp]
            if n > 0 and exp % ws == rk:
                parts.append(w2[off:off + n])
                off += (n + self.block_m - 1) // self.block_m * self.block_m
            else:
                parts.append(None)
        return self.with_share(x, ops.moe_sum_experts_arr(parts, ids.reshape(-1), rev, weights, exp_parallel=ep, world_size=ws, local_rank=rk))

    def forward_per_token(self, x):
        """the same sum written token by token and slot by slot (Fp8Block::forward per expert on one row) -- what the grouped flow
        must reproduce bit for bit; used by the tests"""
        tokens, dim = x.shape
        ids, weights, _ = self.route(x)
        rows = []
        for t, row_ids in enumerate(ids.cpu().tolist()):
            xt = x[t:t + 1]
            for exp in row_ids:
                h0 = ops.fp8_block_linear(xt, self.w_in[exp], self.s_in[exp])
                h1 = ops.fp8_block_linear(xt, self.w_gated[exp], self.s_gated[exp])
                ops.gate_mul(h0, h1, self.act)
                rows.append(ops.fp8_block_linear(h0, self.w_out[exp], self.s_out[exp]))
        y = torch.cat(rows, dim=0)
        pos = ops.arange_i32(tokens * self.top_k, x.device)
        return self.with_share(x, ops.moe_sum_experts(y, pos, weights))
