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


# --------------------------------------------------------------------------- sort_indices
def _all_to_all_v(col: torch.Tensor, send: list, recv: list, group=None) -> torch.Tensor:
    total = sum(recv)
    buf = torch.empty(max(total, 1), dtype=col.dtype, device=col.device)[:total]
    dist.all_to_all_single(buf, col.contiguous(), output_split_sizes=recv, input_split_sizes=send,
                           group=group)
    return buf


def sharded_sort_indices(values, order: str = "ascending", null_placement: str = "at_end", group=None,
                         splitter_bits: int = 12):
    """array_sort_indices over a row-sharded array, one exchange step.

    `values`: this rank's contiguous shard (uint64 / int64 device Array) of the global array
    formed by concatenating the shards in rank order.  Returns (indices, start): `indices` is a
    device int64 tensor of GLOBAL row numbers — this rank's contiguous slice [start, start+len)
    of the globally sorted order; concatenating the ranks' slices in rank order gives exactly
    ArraySortIndices' result (stable; nulls at the end or the start in row order,
    vector_array_sort.cc:524-540, vector_sort_internal.h:225-293).

      1. histogram of the top bits of the order-transformed keys -> all-reduce -> P-1 splitters;
      2. stable partition of the non-null rows by destination rank on the device;
      3. ONE all-to-all(v) of (transformed key, local row) pairs — 12 B per row; receive buffers
         concatenate in source-rank order, so equal keys stay in global row order and the global row
         numbers are rebuilt from the block's source rank;
      4. local stable radix sort (arx_sort_indices_64) + gather of the global rows;
      5. null rows travel (row numbers only) to the last / first rank.
    """
    from . import compute as cp
    from .array import Array, int64, uint64

    if values.type not in (uint64, int64):
        raise _lib.ArrowNotImplementedError("sharded_sort_indices: uint64 / int64 keys only")
    device = values.device
    lib, stream = _lib.get_lib(), current_stream(device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = values.length
    if world == 1:
        # one shard: no splitters, no exchange — the single-GPU kernel path as is
        perm = cp.call_function("array_sort_indices", [values], cp.ArraySortOptions(order, null_placement))
        return perm.data[: n * 8].view(torch.int64), 0
    order_code = _lib.SORT_DESCENDING if order == "descending" else _lib.SORT_ASCENDING
    is_signed = int(values.type == int64)

    # global row number of local row 0
    lens = torch.zeros(world, dtype=torch.int64, device=device)
    lens[rank] = n
    if world > 1:
        dist.all_reduce(lens, group=group)
    lens_h = [int(x) for x in lens.cpu().tolist()]
    shard_offset = sum(lens_h[:rank])

    # 1. splitters
    nbins = 1 << splitter_bits
    hist = torch.zeros(nbins, dtype=torch.int64, device=device)
    span = values.span()
    check(lib.arx_sort_key_histogram(C.byref(span), is_signed, order_code, splitter_bits, hist.data_ptr(),
                                     stream))
    if world > 1:
        dist.all_reduce(hist, group=group)
    cum = torch.cumsum(hist, 0).cpu()
    total_valid = int(cum[-1]) if nbins else 0
    split = []
    for p in range(1, world):
        target = (total_valid * p + world - 1) // world
        b = int(torch.searchsorted(cum, torch.tensor(target, dtype=cum.dtype)).item()) + 1 if total_valid else nbins
        split.append(min(b, nbins))
    split_arr = (C.c_uint32 * max(1, len(split)))(*split)

    # 2. stable partition by destination
    ws_bytes = lib.arx_sort_indices_workspace_bytes(n) + 256
    ws = alloc(ws_bytes, device)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    keys_part = torch.empty(max(n, 1), dtype=torch.int64, device=device)
    rows_part = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    n_valid = C.c_int64(0)
    check(lib.arx_sort_partition_by_bins(C.byref(span), is_signed, order_code, splitter_bits, split_arr, world,
                                         ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), keys_part.data_ptr(),
                                         rows_part.data_ptr(), counts.data_ptr(), C.byref(n_valid), stream))
    nv = n_valid.value
    keys_part = keys_part[:nv]
    rows_part = rows_part[:nv]      # LOCAL row numbers (uint32 in an int32 tensor)

    # 3. the exchange: 12 B per row (transformed key + local row); the receiver rebuilds global row
    #    numbers from the source rank of every received block (blocks arrive in source-rank order)
    offsets = [sum(lens_h[:r]) for r in range(world)]
    if world > 1:
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=group)
        send = [int(x) for x in counts.cpu().tolist()]
        recv = [int(x) for x in recv_counts.cpu().tolist()]
        keys_recv = _all_to_all_v(keys_part, send, recv, group)
        rows_recv = _all_to_all_v(rows_part, send, recv, group)
        base = torch.repeat_interleave(torch.tensor(offsets, dtype=torch.int64, device=device),
                                       torch.tensor(recv, dtype=torch.int64, device=device))
        gidx_recv = (rows_recv.to(torch.int64) & 0xFFFFFFFF) + base
    else:
        keys_recv = keys_part
        gidx_recv = (rows_part.to(torch.int64) & 0xFFFFFFFF) + shard_offset

    # 4. local stable sort of the transformed keys + gather of the global rows
    m = int(keys_recv.numel())
    if m > 0:
        karr = Array(uint64, m, [None, keys_recv.contiguous().view(torch.uint8)], 0, 0)
        perm = cp.call_function("array_sort_indices", [karr], cp.ArraySortOptions("ascending", "at_end"))
        garr = Array(int64, m, [None, gidx_recv.contiguous().view(torch.uint8)], 0, 0)
        sorted_rows = cp.take(garr, perm, boundscheck=False).data[: m * 8].view(torch.int64)
    else:
        sorted_rows = torch.empty(0, dtype=torch.int64, device=device)

    # 5. nulls: row numbers only, to the last (at_end) or first (at_start) rank, in global row order
    target = world - 1 if null_placement == "at_end" else 0
    n_null = n - nv
    null_rows = torch.empty(0, dtype=torch.int64, device=device)
    if n_null > 0:
        fws = alloc(lib.arx_filter_workspace_bytes(n) + 64, device)
        fws_ptr = (fws.data_ptr() + 63) & ~63
        out = torch.empty(n_null, dtype=torch.int32, device=device)
        got = C.c_int64(0)
        check(lib.arx_bitmap_to_indices(values.validity.data_ptr(), values.offset, n, 1, fws_ptr,
                                        fws.numel() - (fws_ptr - fws.data_ptr()), out.data_ptr(),
                                        C.byref(got), stream))
        null_rows = (out[: got.value].to(torch.int64) & 0xFFFFFFFF) + shard_offset
    if world > 1:
        nsend = [0] * world
        nsend[target] = int(null_rows.numel())
        ncounts = torch.tensor(nsend, dtype=torch.int64, device=device)
        nrecv_t = torch.empty_like(ncounts)
        dist.all_to_all_single(nrecv_t, ncounts, group=group)
        nrecv = [int(x) for x in nrecv_t.cpu().tolist()]
        null_rows = _all_to_all_v(null_rows, nsend, nrecv, group)
    if rank == target and null_rows.numel() > 0:
        sorted_rows = torch.cat([sorted_rows, null_rows] if null_placement == "at_end"
                                else [null_rows, sorted_rows])  # buffer concatenation only

    # start of this rank's slice in the global order
    mine = torch.zeros(world, dtype=torch.int64, device=device)
    mine[rank] = sorted_rows.numel()
    if world > 1:
        dist.all_reduce(mine, group=group)
    start = int(mine[:rank].sum().item())
    return sorted_rows, start
