"""Multi-GPU execution of the two paths that shard: one process per
GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

hash_sum group-by — the GroupByNode structure (acero/groupby_aggregate_node.cc:210-337:
thread-local partial state -> Merge -> Finalize) with ranks in place of threads:
  1. each rank aggregates its contiguous row shard into a local table (no communication);
  2. the partial aggregates (one row per local group) are radix-partitioned by
     hash(key) % world_size on the device (arx_groupby_partition);
  3. ONE all-to-all exchanges the partials (counts first, then the five columns);
  4. each rank merges what it received (Merge semantics, hash_aggregate_numeric.cc:85-107)
     and finalizes its disjoint key range.  The global result is the concatenation.
Filter / take / cast / compare do not shard: replicas only.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .array import alloc, current_stream
from .compute import GroupBySum, ScalarAggregateOptions


def partition_partials(partial: dict, num_parts: int, device):
    """arx_groupby_partition: rows grouped by destination + device int64[num_parts] counts."""
    lib = _lib.get_lib()
    stream = current_stream(device)
    g = int(partial["keys"].numel())
    ws = alloc(lib.arx_groupby_partition_workspace_bytes(num_parts), device)
    out = dict(keys=torch.empty(max(g, 1), dtype=torch.int32, device=device),
               key_is_valid=torch.empty(max(g, 1), dtype=torch.uint8, device=device),
               sums=torch.empty(max(g, 1), dtype=torch.int64, device=device),
               counts=torch.empty(max(g, 1), dtype=torch.int64, device=device),
               no_nulls=torch.empty(max(g, 1), dtype=torch.uint8, device=device))
    part_counts = torch.zeros(num_parts, dtype=torch.int64, device=device)
    check(lib.arx_groupby_partition(partial["keys"].data_ptr(), partial["key_is_valid"].data_ptr(),
                                    partial["sums"].data_ptr(), partial["counts"].data_ptr(),
                                    partial["no_nulls"].data_ptr(), g, num_parts, ws.data_ptr(),
                                    ws.numel(), out["keys"].data_ptr(), out["key_is_valid"].data_ptr(),
                                    out["sums"].data_ptr(), out["counts"].data_ptr(),
                                    out["no_nulls"].data_ptr(), part_counts.data_ptr(), stream))
    return {k: v[:g] for k, v in out.items()}, part_counts


def exchange_partials(parts: dict, send_counts: torch.Tensor, group=None):
    """The single all-to-all of the group-by: returns the partials this rank now owns."""
    world = dist.get_world_size(group)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    send = [int(x) for x in send_counts.cpu().tolist()]
    recv = [int(x) for x in recv_counts.cpu().tolist()]
    total = sum(recv)
    out = {}
    for name, col in parts.items():
        buf = torch.empty(max(total, 1), dtype=col.dtype, device=col.device)[:total]
        src = col.contiguous()
        if world == 1:
            buf.copy_(src)
        else:
            dist.all_to_all_single(buf, src, output_split_sizes=recv, input_split_sizes=send,
                                   group=group)
        out[name] = buf
    return out


def sharded_group_by_sum(keys, values, capacity: int, options: ScalarAggregateOptions | None = None,
                         group=None):
    """keys/values: this rank's row shard (device Arrays).  Returns this rank's slice of the
    result: (keys, key_is_valid, sums, valid) device tensors over a disjoint set of keys."""
    device = keys.device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    local = GroupBySum(capacity, device, options)
    local.consume(keys, values)
    if world == 1:
        return local.finalize()
    partial = local.export()
    parts, counts = partition_partials(partial, world, device)
    mine = exchange_partials(parts, counts, group)
    owned = GroupBySum(max(16, 2 * int(mine["keys"].numel()) + 2), device, options)
    owned.merge(mine)
    return owned.finalize()
