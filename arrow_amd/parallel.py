"""Multi-GPU execution of the two paths that shard: one process per
GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

hash_sum group-by — the GroupByNode structure (acero/groupby_aggregate_node.cc:210-337:
thread-local partial state -> Merge -> Finalize) with ranks in place of threads:
  1. each rank aggregates its contiguous row shard into a local table (no communication);
  2. the partial aggregates (one row per local group) are radix-partitioned by
     hash(key) % world_size on the device, straight out of the table (arx_groupby_export_partitioned);
  3. ONE all-to-all(v) exchanges the partials as 24-byte records (after one exchange of the block sizes);
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
from .compute import GroupBySum, RangeGroupBySum, ScalarAggregateOptions


RECORD_BYTES = 24   # sizeof(ArxGroupPartial)
SORT_RECORD_BYTES = 12   # sizeof(ArxSortRecord)
SORT_SAMPLE_SHIFT = 4      # the sharded sort's sampled key range / splitter histogram: one tile of 8192 rows in 16 (0: never sampled)
SORT_SAMPLE_MIN_ROWS = 1 << 20   # ... when the longest shard has at least this many rows (tests lower both)
_U64 = (1 << 64) - 1


def _host_counts(*tensors):
    """ONE device->host read-back for the split sizes of an all-to-all(v) (NCCL wants them on the host)."""
    flat = torch.cat([t.reshape(-1) for t in tensors]).cpu().tolist()
    out, pos = [], 0
    for t in tensors:
        n = t.numel()
        out.append([int(x) for x in flat[pos:pos + n]])
        pos += n
    return out


def _all_to_all_bytes(buf: torch.Tensor, send: list, recv: list, unit: int, group=None) -> torch.Tensor:
    """all_to_all(v) of a uint8 buffer whose blocks are `send[p] * unit` bytes."""
    total = sum(recv) * unit
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=buf.device)[:total]
    dist.all_to_all_single(out, buf.contiguous(), output_split_sizes=[r * unit for r in recv],
                           input_split_sizes=[c * unit for c in send], group=group)
    return out


def export_partitioned(local: GroupBySum, num_parts: int):
    """arx_groupby_export_partitioned: the state's groups as 24-byte records grouped by destination
    (hash(key) % num_parts) + device int64[num_parts] counts.  No dense column export in between."""
    lib = _lib.get_lib()
    device = local.device
    stream = current_stream(device)
    g = local.num_groups()
    ws = alloc(lib.arx_groupby_partition_workspace_bytes(num_parts), device)
    records = torch.empty(max(g, 1) * RECORD_BYTES, dtype=torch.uint8, device=device)
    part_counts = torch.zeros(num_parts, dtype=torch.int64, device=device)
    check(lib.arx_groupby_export_partitioned(local.state.data_ptr(), num_parts, ws.data_ptr(), ws.numel(),
                                             records.data_ptr(), part_counts.data_ptr(), stream))
    return records[: g * RECORD_BYTES], part_counts


def merge_records(owned: GroupBySum, records: torch.Tensor) -> None:
    lib = _lib.get_lib()
    n = records.numel() // RECORD_BYTES
    if n:
        check(lib.arx_groupby_sum_i64_merge_records(owned.state.data_ptr(), owned.capacity, records.data_ptr(), n,
                                                    current_stream(owned.device)))


class Stages:
    """Per-stage wall milliseconds of one sharded call (consume / export / exchange / merge / finalize ...): the stream
    is synchronised at every mark, so a `stages=` run is for diagnosis, not the timed run.  Each rank fills its own;
    bench.py reports the maximum over the ranks."""

    def __init__(self, device):
        import time

        self.device, self.ms, self._clock = device, {}, time.perf_counter
        self._sync()
        self._t = self._clock()

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def reset(self) -> None:
        self.ms.clear()
        self._sync()
        self._t = self._clock()

    def mark(self, name: str) -> None:
        self._sync()
        now = self._clock()
        self.ms[name] = self.ms.get(name, 0.0) + (now - self._t) * 1e3
        self._t = now


def _mark(stages, name):
    if stages is not None:
        stages.mark(name)


ROW_RECORD_BYTES = 16   # sizeof(ArxRowRecord)


def partition_rows(keys, values, num_parts: int):
    """arx_groupby_partition_rows: this shard's ROWS as 16-byte records grouped by the rank that owns their key."""
    lib = _lib.get_lib()
    device = keys.device
    stream = current_stream(device)
    n = keys.length
    ws = alloc(lib.arx_groupby_partition_workspace_bytes(num_parts), device)
    records = torch.empty(max(n, 1) * ROW_RECORD_BYTES, dtype=torch.uint8, device=device)
    part_counts = torch.zeros(num_parts, dtype=torch.int64, device=device)
    ks, vs = keys.span(), values.span()
    check(lib.arx_groupby_partition_rows(C.byref(ks), C.byref(vs), num_parts, ws.data_ptr(), ws.numel(), records.data_ptr(),
                                         part_counts.data_ptr(), stream))
    return records[: n * ROW_RECORD_BYTES], part_counts


def unpack_rows(records: torch.Tensor):
    """Received row records -> (keys Array int32, values Array int64) with validity bitmaps."""
    from .array import Array, bitmap_nbytes, int32, int64

    lib = _lib.get_lib()
    device = records.device
    n = records.numel() // ROW_RECORD_BYTES
    keys = torch.empty(max(n, 1) * 4, dtype=torch.uint8, device=device)
    vals = torch.empty(max(n, 1) * 8, dtype=torch.uint8, device=device)
    kbits = torch.zeros(bitmap_nbytes(n), dtype=torch.uint8, device=device)
    vbits = torch.zeros(bitmap_nbytes(n), dtype=torch.uint8, device=device)
    if n:
        check(lib.arx_groupby_unpack_rows(records.data_ptr(), n, keys.data_ptr(), vals.data_ptr(), kbits.data_ptr(),
                                          vbits.data_ptr(), current_stream(device)))
    return Array(int32, n, [kbits, keys], -1, 0), Array(int64, n, [vbits, vals], -1, 0)


def consume_partials_regions(keys, values, capacity: int, num_parts: int):
    """arx_groupby_sum_i64_consume_partials: the partitioned consume whose groups leave as 24-byte ArxGroupPartial records
    in the region of the rank that owns each key (hash(key) % num_parts) — no local table at all (`capacity`, the slots
    one would have, only plans the pass).  Returns (records uint8 tensor of num_parts regions, records_per_part, counts
    device int64[num_parts]) or None when the shard must go through a local table: keys / values of other types than
    (int32, int64), nulls, a small shard, a region that overflowed."""
    from .array import int32, int64
    from .compute import _next_pow2, _workspace

    if keys.type != int32 or values.type != int64:
        return None
    if (keys.null_count != 0 and keys.buffers[0] is not None) or (values.null_count != 0 and values.buffers[0] is not None):
        return None
    lib = _lib.get_lib()
    device = keys.device
    stream = current_stream(device)
    capacity = _next_pow2(max(2, int(capacity)))
    ws_bytes = lib.arx_groupby_consume_workspace_bytes(keys.length, capacity)
    if not ws_bytes:
        return None
    ws = _workspace(device, ws_bytes + 256, "groupby")
    ws_ptr = (ws.data_ptr() + 255) & ~255
    ws_len = ws.numel() - (ws_ptr - ws.data_ptr())
    per_part = int(lib.arx_groupby_partials_capacity(keys.length, capacity, num_parts))
    records = torch.empty(num_parts * per_part * RECORD_BYTES, dtype=torch.uint8, device=device)
    counts = torch.empty(num_parts, dtype=torch.int64, device=device)
    ks, vs = keys.span(), values.span()
    rc = lib.arx_groupby_sum_i64_consume_partials(None, capacity, C.byref(ks), C.byref(vs), ws_ptr, ws_len, num_parts,
                                                  records.data_ptr(), per_part, counts.data_ptr(), stream)
    if rc in (_lib.ARX_NOT_IMPLEMENTED, _lib.ARX_CAPACITY_ERROR):
        return None
    check(rc)
    return records, per_part, counts


def consume_partials(keys, values, capacity: int, num_parts: int):
    """The local pass WITHOUT the local table: this shard's partial aggregates as 24-byte records grouped by owner,
    compacted for the all-to-all, + device int64[num_parts] counts; (None, None) when the shard has to go through a
    local table.  A key may appear in several records of a block (once per work unit that met it): the receiver's merge
    adds them up, as it adds the partials of different ranks."""
    out = consume_partials_regions(keys, values, capacity, num_parts)
    if out is None:
        return None, None
    regions, per_part, counts = out
    host = [int(c) for c in counts.cpu().tolist()]
    words = regions.view(torch.int64)      # (a record = three 8-byte words: the copy moves words, not bytes)
    blocks = [words[p * per_part * 3: (p * per_part + host[p]) * 3] for p in range(num_parts)]
    return torch.cat(blocks).view(torch.uint8), counts


_FORCE_RANGE_STATE = [False]     # tests: take the range-partitioned state at any row count


def range_partitions_of(rank: int, world: int, partitions: int):
    """The contiguous run [first, first + count) of a range-partitioned state's partitions that `rank` owns."""
    first = rank * partitions // world
    return first, (rank + 1) * partitions // world - first


def _sharded_range_group_by_sum(keys, values, options, group, stages, world, rank):
    """The sharded group-by on the RANGE-PARTITIONED state (round 6; compute.RangeGroupBySum, csrc/groupby_lines.h) — for
    int32 keys from a narrow range (ids, codes), no nulls.  The owner of a key is the owner of its PARTITION (a slice of
    the key range), contiguous runs of partitions per rank:
      1. ONE all-reduce(MAX) of {-min, max} of every shard's sampled keys -> the same plan on every rank;
      2. local consume: write-combined range scatter + direct-indexed LDS aggregate into the rank's dense state;
      3. ONE all-to-all of the state itself — the run of blocks every owner gets has the same, known size on every
         rank: no count exchange, no packing — beside one all-reduce(MIN) of "my consume took its rows";
      4. the owner adds the P runs in one pass (Merge = vector add) and compacts its partitions' groups in key order.
    Returns the result tuple, or None when this path does not apply or any rank's consume declined (every rank then
    takes the table path together: the status is agreed on by the all-reduce)."""
    from .array import int32, int64

    if keys.type != int32 or values.type != int64:
        return None
    # (small shards: the table operator's plans; the row count is not agreed on between ranks, so with several ranks the
    #  range agreement below decides alone)
    if world == 1 and keys.length < RangeGroupBySum.MIN_ROWS and not _FORCE_RANGE_STATE[0]:
        return None
    nulls = (keys.null_count != 0 and keys.buffers[0] is not None) or (values.null_count != 0 and values.buffers[0] is not None)
    device = keys.device
    rng = RangeGroupBySum.sampled_key_range(keys) if not nulls else torch.tensor([2**62, 2**62], dtype=torch.int64, device=device)
    # (a shard with nulls poisons the range: every rank then declines together, without another collective)
    if world > 1:
        dist.all_reduce(rng, op=dist.ReduceOp.MAX, group=group)
    neg_lo, hi = [int(x) for x in rng.cpu().tolist()]
    # (the plan is a function of the agreed range alone: the same on every rank)
    plan = RangeGroupBySum.plan_for(1, -neg_lo, hi) if -neg_lo <= hi and hi + neg_lo < 2**31 else None
    if plan is None:
        return None
    st = RangeGroupBySum(plan, device, options)
    ok = st.consume(keys, values)
    _mark(stages, "consume")
    if world == 1:
        if not ok:
            return None
        out = st.finalize()
        _mark(stages, "finalize")
        return out
    status = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(status, op=dist.ReduceOp.MIN, group=group)
    parts, pb = int(plan.partitions), st.partition_bytes()
    send = [range_partitions_of(r, world, parts)[1] * pb // 8 for r in range(world)]
    first, mine = range_partitions_of(rank, world, parts)
    got = torch.empty(max(world * mine * pb // 8, 1), dtype=torch.int64, device=device)[: world * mine * pb // 8]
    dist.all_to_all_single(got, st.state, output_split_sizes=[mine * pb // 8] * world, input_split_sizes=send, group=group)
    _mark(stages, "exchange")
    if int(status.item()) == 0:     # some rank's consume declined (a hot key, a key outside the sampled range): all take the table path
        return None
    if mine:
        # block 0 += blocks 1 .. P-1 (one pass), then the groups of this rank's partitions in key order
        lib = _lib.get_lib()
        check(lib.arx_groupby_range_merge(got.data_ptr(), got.data_ptr() + mine * pb, plan.width, mine, world - 1, mine * pb,
                                          current_stream(device)))
    _mark(stages, "merge")
    out = st.finalize(first, mine, blocks=got)
    _mark(stages, "finalize")
    return out


def sharded_group_by_sum(keys, values, capacity: int, options: ScalarAggregateOptions | None = None,
                         group=None, exchange: str = "partials", stages: Stages | None = None,
                         local_table: bool | None = None, range_state: bool | None = None):
    """keys/values: this rank's row shard (device Arrays).  Returns this rank's slice of the
    result: (keys, key_is_valid, sums, valid) device tensors over a disjoint set of keys.

    exchange = "partials" (default): local aggregate, then one exchange of the per-destination counts
    and ONE all-to-all(v) of the 24-byte partial-aggregate records — G x 24 bytes leave a rank at most.
    exchange = "rows" (the row-level radix exchange BASELINE.json's north star words): the rows themselves are
    radix-partitioned by hash(key) % world_size, exchanged as 16-byte records (ONE all-to-all(v)) and aggregated by
    the rank that owns their key — N x 16 bytes, the better plan only when almost every row is its own group.
    Both give the same groups on the same ranks.
    range_state (round 6; "partials" exchange with local_table = None): None = int32 keys from a narrow range take the
    range-partitioned state first (_sharded_range_group_by_sum: keys are owned by PARTITION of the key range there, so a
    rank's result is a contiguous slice of the keys in ascending order) and fall back to the paths below where it
    declines; False = never; True = raise where it declines.
    local_table (the "partials" exchange): None = the local pass writes its partials straight into per-owner record
    regions (consume_partials: no local table, no export pass) and falls back to local table + export where that form
    declines; True = always the local table (round 4's path); False = never (raises where the form declines)."""
    device = keys.device
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if exchange not in ("partials", "rows"):
        raise ValueError("exchange must be 'partials' or 'rows'")
    # Round 6: keys from a narrow range take the range-partitioned state — no hash table, the owner of a key is the owner
    # of its partition, ONE all-to-all of dense blocks whose sizes every rank knows.  range_state: None = try it and fall
    # back (all ranks together) where it declines; False = never; True = it must apply.
    if range_state is not False and exchange == "partials" and local_table is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        _FORCE_RANGE_STATE[0] = range_state is True
        out = _sharded_range_group_by_sum(keys, values, options, group, stages, world, rank)
        if out is not None:
            return out
        if range_state is True:
            raise _lib.ArrowNotImplementedError("sharded_group_by_sum(range_state=True): the key range (or a shard's rows) does "
                                                "not suit the range-partitioned state")
        if stages is not None:
            stages.reset()      # (the declined attempt's marks would be added to the table path's)
    if world == 1:
        local = GroupBySum(capacity, device, options)
        local.consume(keys, values)
        _mark(stages, "consume")
        out = local.finalize()
        _mark(stages, "finalize")
        return out
    if exchange == "rows":
        records, counts = partition_rows(keys, values, world)
        _mark(stages, "partition_rows")
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=group)
        send, recv = _host_counts(counts, recv_counts)
        mine = _all_to_all_bytes(records, send, recv, ROW_RECORD_BYTES, group)
        _mark(stages, "exchange")
        rk, rv = unpack_rows(mine)
        # the owner's table is sized from what ARRIVED (as the partials path and the C++ ShardedGroupBySum do): a rank can
        # own more distinct keys than its own shard held, which the caller's per-shard capacity says nothing about
        owned = GroupBySum(max(capacity, 16, 2 * sum(recv) + 2), device, options)
        owned.consume(rk, rv)
        _mark(stages, "consume")
        out = owned.finalize()
        _mark(stages, "finalize")
        return out
    records = None
    if local_table is not True:
        records, counts = consume_partials(keys, values, capacity, world)
        if records is not None:
            _mark(stages, "consume")
        elif local_table is False:
            raise _lib.ArrowNotImplementedError("sharded_group_by_sum(local_table=False): this shard needs the local table "
                                                "(nulls, a small batch, or more partials than their regions hold)")
    if records is None:
        local = GroupBySum(capacity, device, options)
        local.consume(keys, values)
        _mark(stages, "consume")
        records, counts = export_partitioned(local, world)
        _mark(stages, "export")
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts, group=group)
    send, recv = _host_counts(counts, recv_counts)
    mine = _all_to_all_bytes(records, send, recv, RECORD_BYTES, group)
    _mark(stages, "exchange")
    owned = GroupBySum(max(16, 2 * sum(recv) + 2), device, options)
    merge_records(owned, mine)
    _mark(stages, "merge")
    out = owned.finalize()
    _mark(stages, "finalize")
    return out


# --------------------------------------------------------------------------- sort_indices
def sharded_sort_indices(values, order: str = "ascending", null_placement: str = "at_end", group=None,
                         splitter_bits: int = 12, stages: Stages | None = None, records_form: bool | None = None):
    """array_sort_indices over a row-sharded array, one exchange step.

    `values`: this rank's contiguous shard (uint64 / int64 device Array) of the global array
    formed by concatenating the shards in rank order.  Returns (indices, start): `indices` is a
    device int64 tensor of GLOBAL row numbers — this rank's contiguous slice [start, start+len)
    of the globally sorted order; concatenating the ranks' slices in rank order gives exactly
    ArraySortIndices' result (stable; nulls at the end or the start in row order,
    vector_array_sort.cc:524-540, vector_sort_internal.h:225-293).

      0. one 16-byte all-reduce (MAX): the range [min, max] of the order-transformed keys of all shards.  Row ids,
         timestamps and small integers share their top bits; splitter bins of the raw key would send every row to
         one rank, so the bins are taken inside that window: bin = top bits of (key - min) << clz(max - min);
      1. ONE all-reduce: histogram of those bins (-> P-1 splitters, and every
         rank's slice of the result) + the shard lengths (-> global row numbers) + the null counts;
      2. stable partition of the non-null rows by destination rank on the device, packed as 12-byte
         {transformed key, local row} records; the shard's null rows (row numbers only) ride in the same
         buffer, in the block of the last (at_end) / first (at_start) rank;
      3. one exchange of the block sizes and ONE all-to-all(v) of the records; blocks arrive in source-rank
         order, so equal keys stay in global row order;
      4. arx_sort_unpack_records rebuilds keys + global rows, then the local stable sort (arx_sort_indices_64)
         and one gather of the global rows.
    records_form (round 6): None = when no shard has a null and the global rows fit 32 bits, steps 2 - 4 are the records
    form — the partition writes {key, GLOBAL row} straight from the column with no stable pass
    (arx_sort_partition_records_global), the receiver sorts its records by (key, row) (arx_sort_records: the order a
    stable sort of the keys gives) — no unpack, no column sort, no gather; False = always the steps above; True = raise
    where the form does not apply.
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
        _mark(stages, "local_sort")
        return perm.data[: n * 8].view(torch.int64), 0
    order_code = _lib.SORT_DESCENDING if order == "descending" else _lib.SORT_ASCENDING
    placement_code = _lib.NULLS_AT_START if null_placement == "at_start" else _lib.NULLS_AT_END
    is_signed = int(values.type == int64)
    nulls_first = null_placement == "at_start"
    target = 0 if nulls_first else world - 1

    # 0. the key window: {max of ~key, max of key}, both by MAX; unsigned order as int64 = sign bit flipped.  The same
    #    all-reduce carries what decides the round-6 SAMPLED form for every rank alike: "this shard may have nulls" and the
    #    shard's length (MAX: no shard has nulls, and world x the longest shard < 2^32 rows) — then the records form is certain,
    #    nothing below needs exact per-bin counts, and the range and the histogram are taken from 1 tile of 8192 rows in 16
    #    (two passes over every key were 1.8 of a rank's 18 ms at P = 4, profiles/r06_u_*).
    span = values.span()
    may_have_nulls = int(values.buffers[0] is not None and values.null_count != 0)
    key_range = torch.zeros(4, dtype=torch.int64, device=device)
    sample_shift = SORT_SAMPLE_SHIFT if (records_form is not False and not may_have_nulls and n >= SORT_SAMPLE_MIN_ROWS) else 0
    check(lib.arx_sort_key_range_sampled(C.byref(span), is_signed, order_code, sample_shift, key_range.data_ptr(), stream))
    sign = torch.iinfo(torch.int64).min
    key_range[:2] ^= sign
    key_range[2] = 1 if (may_have_nulls or records_form is False) else 0
    key_range[3] = n
    dist.all_reduce(key_range, op=dist.ReduceOp.MAX, group=group)
    reduced = key_range.tolist()
    inv_min, key_max = [int(x) ^ sign for x in reduced[:2]]
    key_min, key_max = ~inv_min & _U64, key_max & _U64
    # (a rank that sampled its range while the ranks together take the exact form: its range may miss keys — they fall into
    #  the window's end bins, in the histogram and in the partition alike, so every count stays exact)
    sampled = SORT_SAMPLE_SHIFT > 0 and reduced[2] == 0 and int(reduced[3]) * world < 2**32 and int(reduced[3]) >= SORT_SAMPLE_MIN_ROWS
    if sampled and key_max > key_min:      # a sample misses a few keys at both ends: widen (keys outside fall into the end bins)
        margin = ((key_max - key_min) >> 6) + 1
        key_min, key_max = max(0, key_min - margin), min(_U64, key_max + margin)
    window = _lib.ArxSortKeyWindow(0, 0, 0)
    if key_max > key_min:        # (no valid row anywhere: min = 2^64 - 1 > max = 0; one distinct key: nothing to split)
        window = _lib.ArxSortKeyWindow(key_min, 64 - (key_max - key_min).bit_length(), 0)

    # 1. one all-reduce: [histogram (2^bits) | shard lengths (world) | valid rows per shard (world)]
    nbins = 1 << splitter_bits
    stats = torch.zeros(nbins + 2 * world, dtype=torch.int64, device=device)
    check(lib.arx_sort_key_histogram_window_sampled(C.byref(span), is_signed, order_code, splitter_bits, C.byref(window),
                                                    SORT_SAMPLE_SHIFT if sampled else 0, stats.data_ptr(), stream))
    stats[nbins + rank] = n
    stats[nbins + world + rank] = n if sampled else stats[:nbins].sum()
    dist.all_reduce(stats, group=group)
    stats_h = stats.cpu()
    _mark(stages, "histogram")
    cum = torch.cumsum(stats_h[:nbins], 0)
    lens_h = [int(x) for x in stats_h[nbins:nbins + world].tolist()]
    valid_h = [int(x) for x in stats_h[nbins + world:].tolist()]
    total_valid = int(cum[-1])           # (sampled: the rows of the SAMPLE — the splitters are its quantiles)
    total_nulls = 0 if sampled else sum(lens_h) - total_valid
    split = []
    for p in range(1, world):
        want = (total_valid * p + world - 1) // world
        b = int(torch.searchsorted(cum, torch.tensor(want, dtype=cum.dtype)).item()) + 1 if total_valid else nbins
        split.append(min(b, nbins))
    split_arr = (C.c_uint32 * max(1, len(split)))(*split)
    # what every rank will own (known without another collective): the bins below its splitter
    edges = [0] + split + [nbins]
    owned_valid = [int(cum[edges[r + 1] - 1]) - (int(cum[edges[r] - 1]) if edges[r] > 0 else 0)
                   if edges[r + 1] > edges[r] else 0 for r in range(world)]
    owned = [owned_valid[r] + (total_nulls if r == target else 0) for r in range(world)]
    start = sum(owned[:rank])
    offsets = [sum(lens_h[:r]) for r in range(world)]

    # Round 6: no nulls anywhere and global rows that fit 32 bits — the records carry GLOBAL rows, the partition needs no
    # stable pass (arx_sort_partition_records_global: one tile-level pass from the column) and the receiver sorts its
    # records by (key, row) (arx_sort_records): no unpack, no column sort + gather.
    if records_form is True and not (total_nulls == 0 and sum(lens_h) < 2**32):
        raise _lib.ArrowNotImplementedError("sharded_sort_indices(records_form=True): a shard has nulls, or the global rows pass 2^32")
    if records_form is not False and total_nulls == 0 and sum(lens_h) < 2**32:
        records = torch.empty(max(n, 1) * SORT_RECORD_BYTES, dtype=torch.uint8, device=device)
        counts = torch.zeros(world, dtype=torch.int64, device=device)
        pws = alloc(1024, device)
        check(lib.arx_sort_partition_records_global(C.byref(span), is_signed, order_code, splitter_bits, C.byref(window), split_arr,
                                                    world, offsets[rank], pws.data_ptr(), pws.numel(), records.data_ptr(),
                                                    counts.data_ptr(), stream))
        _mark(stages, "partition")
        # the block sizes: every rank's counts to every rank (one all-gather of world numbers — what the count exchange moved
        # before —: a rank's row is what it receives, the column sums are what every rank will own, exact whatever placed
        # the splitters)
        rows_of = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(rows_of, counts, group=group)
        matrix_h = torch.stack(rows_of).cpu()
        send = [int(x) for x in matrix_h[rank].tolist()]
        recv = [int(x) for x in matrix_h[:, rank].tolist()]
        owned = [int(x) for x in matrix_h.sum(dim=0).tolist()]
        start = sum(owned[:rank])
        got = _all_to_all_bytes(records[: n * SORT_RECORD_BYTES], send, recv, SORT_RECORD_BYTES, group)
        _mark(stages, "exchange")
        m = sum(recv)
        assert m == owned[rank], (m, owned[rank])
        sorted_rows = torch.empty(max(m, 1), dtype=torch.int64, device=device)[:m]
        if m:
            ws_bytes = lib.arx_sort_indices_workspace_bytes(m) + 256
            ws = alloc(ws_bytes, device)
            ws_ptr = (ws.data_ptr() + 255) & ~255
            check(lib.arx_sort_records(got.data_ptr(), m, ws_ptr, ws.numel() - (ws_ptr - ws.data_ptr()), sorted_rows.data_ptr(), stream))
        _mark(stages, "local_sort")
        return sorted_rows, start

    # 2. stable partition by destination, packed records (+ this shard's null rows)
    ws_bytes = lib.arx_sort_indices_workspace_bytes(n) + 256
    ws = alloc(ws_bytes, device)
    ws_ptr = (ws.data_ptr() + 255) & ~255
    records = torch.empty(max(n, 1) * SORT_RECORD_BYTES, dtype=torch.uint8, device=device)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    n_valid = C.c_int64(0)
    check(lib.arx_sort_partition_records_window(C.byref(span), is_signed, order_code, placement_code, splitter_bits,
                                                C.byref(window), split_arr, world, ws_ptr,
                                                ws.numel() - (ws_ptr - ws.data_ptr()), records.data_ptr(),
                                                counts.data_ptr(), C.byref(n_valid), stream))
    n_null = n - n_valid.value
    _mark(stages, "partition")

    # 3. block sizes, then the ONE data exchange
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts, group=group)
    send_valid, recv_valid = _host_counts(counts, recv_counts)
    send = [send_valid[p] + (n_null if p == target else 0) for p in range(world)]
    recv_nulls = [(lens_h[s] - valid_h[s]) if rank == target else 0 for s in range(world)]
    recv = [recv_valid[s] + recv_nulls[s] for s in range(world)]
    got = _all_to_all_bytes(records[: n * SORT_RECORD_BYTES], send, recv, SORT_RECORD_BYTES, group)
    _mark(stages, "exchange")

    # 4. unpack (keys + GLOBAL rows, compacted in source order), local stable sort, one gather
    m_valid, m_null = sum(recv_valid), sum(recv_nulls)
    meta = torch.tensor([[recv_valid[s], recv_nulls[s], offsets[s]] for s in range(world)], dtype=torch.int64).to(device)
    keys_recv = torch.empty(max(m_valid, 1), dtype=torch.int64, device=device)
    gidx_recv = torch.empty(max(m_valid, 1), dtype=torch.int64, device=device)
    null_rows = torch.empty(max(m_null, 1), dtype=torch.int64, device=device)
    check(lib.arx_sort_unpack_records(got.data_ptr(), m_valid + m_null, meta.data_ptr(), world, int(nulls_first),
                                      keys_recv.data_ptr(), gidx_recv.data_ptr(), null_rows.data_ptr(), stream))
    if m_valid > 0:
        karr = Array(uint64, m_valid, [None, keys_recv.view(torch.uint8)], 0, 0)
        perm = cp.call_function("array_sort_indices", [karr], cp.ArraySortOptions("ascending", "at_end"))
        garr = Array(int64, m_valid, [None, gidx_recv.view(torch.uint8)], 0, 0)
        sorted_rows = cp.take(garr, perm, boundscheck=False).data[: m_valid * 8].view(torch.int64)
    else:
        sorted_rows = torch.empty(0, dtype=torch.int64, device=device)
    if m_null > 0:
        parts = [null_rows[:m_null], sorted_rows] if nulls_first else [sorted_rows, null_rows[:m_null]]
        sorted_rows = torch.cat(parts)   # buffer concatenation only (the target rank, shards with nulls)
    assert int(sorted_rows.numel()) == owned[rank], (int(sorted_rows.numel()), owned[rank])
    _mark(stages, "local_sort")
    return sorted_rows, start
