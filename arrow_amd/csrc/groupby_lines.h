// The LINES plan of the partitioned group-by (round 6) — included by groupby.hip inside namespace arx.
//
// For int32 keys whose values span a range of at most kGblMaxBins x 12288 (ids, codes, dictionary indices: the usual
// int32 key column; BASELINE configs[3] draws 1e7 keys from [0, 1e7)) the partition of a key is a SLICE OF ITS RANGE —
// bin = (key - kmin) / width — and a partition's groups are aggregated in a DIRECT-INDEXED LDS table (no tags, no
// probing, load factor 1).  What the flat scatter of the wide plan paid for was partial 128-byte lines: unaligned
// 144-byte (tile, bin) runs, 1.21x write amplification, 3.6 TB/s (profiles/groupby_traffic.json).
// Here every global store of the scatter is a WHOLE line (scripts/micro/wc_lines_bench.hip, profiles/r06_a..c: the
// memory side alone moves 12 B read + whole lines written to 814 frontiers at 5.1 TB/s; this kernel 4.6 TB/s):
//
//   line     128 bytes = two halves of {6 x u64 value, 6 x u16 key remainder, u32 count}: 12 records, 10.67 B/record,
//            self-contained (a line belongs to ONE bin; a count < 6 marks a workgroup's last, partial line).
//   scatter  persistent workgroups (one per CU), ONE line per bin in LDS.  A row takes its slot with a returning LDS
//            atomic on the bin's fill counter; the row that takes slot 11 queues the bin; after a barrier 8 lanes copy
//            a full line out, 16 bytes each.  A line's place comes from the workgroup's chunk of kGblK lines of that
//            bin's room — one returning global atomic per chunk, issued when the chunk's last line leaves and picked
//            up before the next flush, on a cursor that has its own 128-byte line (packed cursors queue single-lane
//            atomics at ~110 M/s per line: 782 GB/s, profiles/r06_a_*).  A row that finds its line full is carried
//            into the next batch (two per thread); a thread with more falls back to append / flush rounds.
//   rooms    from a SAMPLED histogram (one 64-row unit per stratum of S units, 2^26 rows in all): estimate + 6 sigma
//            + the workgroups' chunk tails.  A room that overflows, keys outside the sampled range beyond a few, or
//            a hot key (rounds without end) set a flag: nothing has touched the table yet, the caller's other plans
//            take the rows.
//   aggregate quads of lanes read a half line with 16-byte loads (three lanes two values each, the fourth the six
//            remainders + count), ds_add_u64 / ds_add_u32 into sums[width] / counts[width], then the flush into the
//            HBM table as the other plans do.
// 4e9 rows / 1e7 keys: scatter 19.7 ms + aggregate 6.7 ms in the micro-benchmark against 26 - 29.5 + 9.9 ms for the
// wide plan's two kernels.
constexpr int kGblThreads = 1024;
constexpr int kGblMaxBins = 1216;            // 1216 x 128 B of line buffers = 152 KB of the CU's 160 KB
constexpr int kGblCap = 12;                  // records per line
constexpr int kGblK = 8;                     // lines per chunk (one global atomic each: WRITE_SIZE counts 32 bytes for it)
constexpr int kGblR = 4;                     // rows per thread and batch (8: the registers spill)
constexpr int kGblCursorStride = 32;         // u32 between two bins' cursors: a 128-byte line each
constexpr int kGblMaxWidth = 12288;          // groups per partition: sums u64 + counts u32 = 144 KB of LDS
constexpr int kGblMinBins = 384;             // fewer bins: a batch of 4096 rows brings a bin many times the 12 slots of its ONE line (round after round)
constexpr uint32_t kGblSkip = 0xFFFFFFFFu;
constexpr uint32_t kGblNever = 0xFFFFFFFEu;  // state of a bin this workgroup never wrote a line of
constexpr int kGblUnitRows = 64;             // rows of a sampling unit (one wave load)

struct GblArgs {
  const int32_t* keys;
  const int64_t* values;
  Bits kvalid, vvalid;
  int64_t n;
  int64_t rows_per_wg;     // a multiple of kGblR * kGblThreads
  int32_t kmin;
  int wshift;              // width = 1 << wshift, or (wshift == 0) 12288
  int width;
  int bins;
  int64_t sample_stride;   // S: one 64-row unit of every S is sampled
  int wgs;                 // workgroups of the scatter
  uint32_t total_lines;    // lines the buffer holds
  uint32_t unit_lines;     // lines per aggregate work unit
  uint32_t* cursor;        // [bins * kGblCursorStride] next free line (absolute) of every bin's room
  uint32_t* room_start;    // [bins + 1]
  uint32_t* hist;          // [bins] sampled rows per bin
  uint32_t* flags;         // [0] a room / the buffer overflowed, [1] rounds without end, [2] rows outside [kmin, kmin + bins * width)
  uint32_t* unit_start;    // [bins + 1] aggregate work units before bin b
  uint8_t* lines;
  uint8_t* dense;          // where the aggregate adds a partition's sums and counts: bin b at b * width * 16 = {u64 sums[width], u64 counts[width]} (64-bit counts: a state may take more than 2^32 rows of one key over its consumes and merges)
};

__device__ __forceinline__ void gbl_split(const GblArgs& a, uint32_t d, uint32_t& bin, uint32_t& rem) {
  if (a.wshift != 0) {
    bin = d >> a.wshift;
    rem = d & ((1u << a.wshift) - 1u);
  } else {   // 12288 = 3 * 4096: (d >> 12) / 3, exact for d < 2^28
    bin = ((d >> 12) * 0xAAABu) >> 17;
    rem = d - bin * 12288u;
  }
}

template <bool HAS_NULLS>
__device__ __forceinline__ bool gbl_row_streamed(const GblArgs& a, int64_t r) {
  if constexpr (!HAS_NULLS) {
    return true;
  } else {   // rows with a null key or a null value are gbp_null_rows_kernel's
    const uint64_t kv = load_word(a.kvalid, r >> 6);
    const uint64_t vv = load_word(a.vvalid, r >> 6);
    return ((kv & vv) >> (r & 63)) & 1ull;
  }
}

// the unit of stratum s that the sample reads (a pseudo-random position, so that a periodic input cannot line up with it)
__device__ __forceinline__ int64_t gbl_sampled_unit(int64_t s, int64_t stride) {
  uint64_t z = static_cast<uint64_t>(s) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z ^= z >> 29;
  return s * stride + static_cast<int64_t>(z % static_cast<uint64_t>(stride));
}

// ---- the sampled key range (slots of null keys included: they only widen it)
__global__ __launch_bounds__(kBlock) void gbl_range_kernel(const int32_t* __restrict__ keys, int64_t n, int64_t stride,
                                                           int32_t* __restrict__ out_min_max) {
  const int lane = lane_id();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  const int64_t units = (n + kGblUnitRows - 1) / kGblUnitRows;
  const int64_t strata = (units + stride - 1) / stride;
  int32_t lo = INT32_MAX, hi = INT32_MIN;
  // eight units in flight per wave (one per iteration leaves the wave waiting for its own load)
  for (int64_t s0 = (static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * 8; s0 < strata; s0 += nwaves * 8) {
    int32_t k[8];
    bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t u = gbl_sampled_unit(s0 + j < strata ? s0 + j : strata - 1, stride);
      u = u < units ? u : units - 1;
      const int64_t r = u * kGblUnitRows + lane;
      ok[j] = s0 + j < strata && r < n;
      k[j] = keys[r < n ? r : n - 1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ok[j]) {
        lo = k[j] < lo ? k[j] : lo;
        hi = k[j] > hi ? k[j] : hi;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int32_t ol = __shfl_xor(lo, d, 64), oh = __shfl_xor(hi, d, 64);
    lo = ol < lo ? ol : lo;
    hi = oh > hi ? oh : hi;
  }
  // ONE atomic pair per workgroup (same-address atomics queue up behind the L2 at ~10 ns apiece: a pair per wave of a
  // 2048-workgroup grid was 160 of this kernel's 190 us, profiles/r06_g_*)
  __shared__ int32_t wlo[kWavesPerBlock], whi[kWavesPerBlock];
  if (lane == 0) {
    wlo[threadIdx.x >> 6] = lo;
    whi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < kWavesPerBlock; ++wv) {
      lo = wlo[wv] < lo ? wlo[wv] : lo;
      hi = whi[wv] > hi ? whi[wv] : hi;
    }
    if (lo <= hi) {
      atomicMin(&out_min_max[0], lo);
      atomicMax(&out_min_max[1], hi);
    }
  }
}

// ---- the sampled histogram over the plan's bins
template <bool HAS_NULLS>
__global__ __launch_bounds__(kGblThreads) void gbl_hist_kernel(GblArgs a) {
  __shared__ uint32_t h[kGblMaxBins];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int b = tid; b < a.bins; b += kGblThreads) h[b] = 0;
  __syncthreads();
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * (kGblThreads / 64);
  const int64_t units = (a.n + kGblUnitRows - 1) / kGblUnitRows;
  const int64_t strata = (units + a.sample_stride - 1) / a.sample_stride;
  const uint32_t span = static_cast<uint32_t>(a.bins) * static_cast<uint32_t>(a.width);
  for (int64_t s0 = (static_cast<int64_t>(blockIdx.x) * (kGblThreads / 64) + (tid >> 6)) * 8; s0 < strata; s0 += nwaves * 8) {
    int32_t k[8];
    bool ok[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int64_t u = gbl_sampled_unit(s0 + j < strata ? s0 + j : strata - 1, a.sample_stride);
      u = u < units ? u : units - 1;
      const int64_t r = u * kGblUnitRows + lane;
      ok[j] = s0 + j < strata && r < a.n && gbl_row_streamed<HAS_NULLS>(a, r < a.n ? r : a.n - 1);
      k[j] = a.keys[r < a.n ? r : a.n - 1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t d = static_cast<uint32_t>(k[j] - a.kmin);
      if (ok[j] && d < span) {
        uint32_t bin, rem;
        gbl_split(a, d, bin, rem);
        atomicAdd(&h[bin], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = tid; b < a.bins; b += kGblThreads) {
    if (h[b] != 0) atomicAdd(&a.hist[b], h[b]);
  }
}

// ---- rooms from the sampled histogram: estimate + 6 sigma of the sample + 1/64 + the workgroups' chunk tails
__global__ __launch_bounds__(kGblThreads) void gbl_rooms_kernel(GblArgs a) {
  __shared__ uint32_t wave_tot[kGblThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (a.bins + kGblThreads - 1) / kGblThreads;   // 1 or 2
  uint32_t need[2] = {0, 0};
  uint32_t mine = 0;
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    if (b < a.bins) {
      const double est = static_cast<double>(a.hist[b]) * static_cast<double>(a.sample_stride);
      const double sigma = a.sample_stride > 1 ? sqrt(est * static_cast<double>(a.sample_stride) + 1.0) : 0.0;
      const double rows = est + 6.0 * sigma + est / 64.0 + 64.0 * static_cast<double>(a.sample_stride > 1 ? a.sample_stride : 0);
      // (a workgroup leaves at most two chunks' worth of pads in a bin: its last chunk's tail and the chunk it reserved ahead)
      double lines = rows / kGblCap + static_cast<double>(a.wgs) * (2 * kGblK) + 2.0;
      lines = lines < 4.0e9 ? lines : 4.0e9;
      need[k] = (static_cast<uint32_t>(lines) + kGblK - 1) / kGblK * kGblK;
      mine += need[k];
    }
  }
  // (sums past 2^32 wrap: the 64-bit total is checked below)
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  unsigned long long total64 = wave_reduce_sum_u64(static_cast<unsigned long long>(mine));
  __shared__ unsigned long long tot64[kGblThreads / 64];
  if (lane == 63) wave_tot[wave] = incl;
  if (lane == 0) tot64[wave] = total64;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += wave_tot[k];
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    if (b < a.bins) {
      a.room_start[b] = pre;
      a.cursor[b * kGblCursorStride] = pre;
      pre += need[k];
    }
  }
  if (tid == 0) {
    unsigned long long all = 0;
    for (int k = 0; k < kGblThreads / 64; ++k) all += tot64[k];
    a.room_start[a.bins] = all > 0xFFFFFFFFull ? 0xFFFFFFFFu : static_cast<uint32_t>(all);
    if (all > static_cast<unsigned long long>(a.total_lines)) a.flags[0] = 2u;   // the rooms do not fit the buffer
  }
}

// ---- the write-combined scatter
template <bool HAS_NULLS>
__global__ __launch_bounds__(kGblThreads) void gbl_scatter_kernel(GblArgs a) {
  __shared__ uint64_t vals[kGblMaxBins * kGblCap];
  __shared__ uint16_t rems[kGblMaxBins * kGblCap];
  // state = next line (absolute) << 1 | the workgroup holds that line: rooms and chunks start at multiples of kGblK, so
  // the line that ends a chunk is the one before a multiple of kGblK (31 bits of line numbers: 2.5e10 rows)
  __shared__ uint32_t fill[kGblMaxBins], state[kGblMaxBins];
  __shared__ uint16_t wlist[kGblThreads];   // per wave: which of its bins have a full line (flush_phase)
  __shared__ uint32_t again[2], stop;
  constexpr int R = kGblR, K = kGblK;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int b = tid; b < a.bins; b += kGblThreads) {
    fill[b] = 0;
    state[b] = kGblNever;
  }
  if (tid < 2) again[tid] = 0;
  if (tid == 0) stop = a.flags[0];   // (the rooms did not fit: nothing to do; one thread reads the flag for the workgroup)
  __syncthreads();
  if (stop != 0) return;
  const int64_t lo = static_cast<int64_t>(blockIdx.x) * a.rows_per_wg;
  const int64_t hi = lo + a.rows_per_wg < a.n ? lo + a.rows_per_wg : a.n;
  if (lo >= hi) return;
  const uint32_t span = static_cast<uint32_t>(a.bins) * static_cast<uint32_t>(a.width);
  // (round 6, after the same change to the sort's level 1 — profiles/r06_p_*: the waves of these kernels issue for a third
  //  of their time and four of them share a SIMD's VALU, so a batch's instructions are part of its duration.  Rows are
  //  addressed by 32-bit offsets from the workgroup's first; only a workgroup's last batch checks for the end of its rows;
  //  and the flush finds the full lines itself instead of every row that takes a line's last slot queueing its bin with
  //  a returning LDS atomic and a wait — some lane of 64 nearly always does)
  const uint32_t wg_rows = static_cast<uint32_t>(hi - lo);
  const int32_t* __restrict__ keys0 = a.keys + lo;
  const int64_t* __restrict__ values0 = a.values + lo;
  constexpr uint32_t kBatch = static_cast<uint32_t>(R) * kGblThreads;
  int32_t kc[R], kn[R];
  int64_t vc[R], vn[R];
  auto issue = [&](uint32_t rel, auto full) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      uint32_t r = rel + static_cast<uint32_t>(i * kGblThreads + tid);
      if constexpr (!decltype(full)::value) r = r < wg_rows ? r : wg_rows - 1;
      kn[i] = __builtin_nontemporal_load(keys0 + r);
      vn[i] = __builtin_nontemporal_load(values0 + r);
    }
  };
  auto issue_at = [&](uint32_t rel) {
    if (rel + kBatch <= wg_rows) issue(rel, std::true_type{});
    else issue(rel, std::false_type{});
  };
  int cur = 0;
  uint32_t outliers = 0;
  auto place = [&](uint32_t bin, uint32_t slot, uint32_t rem, int64_t val) {
    vals[bin * kGblCap + slot] = static_cast<uint64_t>(val);
    rems[bin * kGblCap + slot] = static_cast<uint16_t>(rem);
  };
  auto next_line = [&](uint32_t bin) -> uint32_t {
    uint32_t s = state[bin];
    if ((s & 1u) == 0) s = (atomicAdd(&a.cursor[bin * kGblCursorStride], static_cast<uint32_t>(K)) << 1) | 1u;
    const uint32_t line = s >> 1;
    state[bin] = ((line + 1) << 1) | ((line + 1) % K != 0 ? 1u : 0u);
    return line;
  };
  // (a line past its room lands in the next bin's room — gbl_scan_kernel sees the cursor past the room's end and the
  //  whole pass is dropped —; only the end of the buffer must never be passed)
  auto store_piece = [&](uint32_t line, int sub, arx_u32x4 d) {
    if (line >= a.total_lines) {
      atomicOr(&a.flags[0], 1u);
      return;
    }
    *reinterpret_cast<arx_u32x4*>(a.lines + static_cast<size_t>(line) * 128 + sub * 16) = d;
  };
  auto piece = [&](uint32_t bin, int sub, uint32_t c0, uint32_t c1) -> arx_u32x4 {
    const int q = sub & 3, h = sub >> 2;
    if (q < 3) return *reinterpret_cast<const arx_u32x4*>(&vals[bin * kGblCap + 6 * h + 2 * q]);
    const uint32_t* r = reinterpret_cast<const uint32_t*>(&rems[bin * kGblCap + 6 * h]);
    arx_u32x4 d = {r[0], r[1], r[2], h ? c1 : c0};
    return d;
  };
  // the chunks this thread's bins (tid, tid + 1024) wait for
  bool rf0 = false, rf1 = false;
  uint32_t rv0 = 0, rv1 = 0;
  auto flush_phase = [&]() -> uint32_t {
    if (rf0) {
      state[tid] = (rv0 << 1) | 1u;
      rf0 = false;
    }
    if (rf1) {
      state[tid + kGblThreads] = (rv1 << 1) | 1u;
      rf1 = false;
    }
    __syncthreads();
    const uint32_t go = again[cur];
    if (tid == 0) again[cur ^ 1] = 0;
    // a wave flushes the full lines of ITS bins (bin = thread, then thread + 1024): which ones, compacted through the wave's
    // piece of wlist (written and read by this wave only — its LDS operations execute in order), 8 lanes a line
    const int sub = tid & 7;
    for (int first = 0; first < a.bins; first += kGblThreads) {   // (workgroup-uniform)
      const int mine = first + tid;
      const bool full = mine < a.bins && fill[mine] >= static_cast<uint32_t>(kGblCap);
      const uint64_t fmask = __ballot(full);
      const uint32_t nf = static_cast<uint32_t>(__popcll(fmask));   // (wave-uniform)
      if (full) wlist[(tid & ~63) + __popcll(fmask & ((uint64_t(1) << lane) - 1))] = static_cast<uint16_t>(mine);
      __builtin_amdgcn_wave_barrier();
      for (uint32_t g0 = 0; g0 < nf; g0 += 8) {
        const uint32_t g = g0 + static_cast<uint32_t>(lane >> 3);
        const bool on = g < nf;
        const uint32_t bin = on ? wlist[(tid & ~63) + g] : 0u;
        uint32_t line = 0;
        if (on && sub == 0) line = next_line(bin);
        line = __shfl(line, lane & ~7, 64);
        if (on) {
          store_piece(line, sub, piece(bin, sub, 6, 6));
          if (sub == 0) fill[bin] = 0;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    cur ^= 1;
    if (tid < a.bins) {
      const uint32_t s = state[tid];
      if ((s & 1u) == 0 && s != kGblNever) {
        rv0 = atomicAdd(&a.cursor[tid * kGblCursorStride], static_cast<uint32_t>(K));
        rf0 = true;
      }
    }
    if (tid + kGblThreads < a.bins) {
      const uint32_t s = state[tid + kGblThreads];
      if ((s & 1u) == 0 && s != kGblNever) {
        rv1 = atomicAdd(&a.cursor[(tid + kGblThreads) * kGblCursorStride], static_cast<uint32_t>(K));
        rf1 = true;
      }
    }
    return go;
  };
  int32_t pk0 = 0, pk1 = 0;   // carried rows
  int64_t pv0 = 0, pv1 = 0;
  uint32_t np = 0;
  bool gave_up = false;
  // One batch of R rows per thread; FULL: every row of it lies inside the workgroup's rows.  false: the workgroup gives up.
  auto batch = [&](uint32_t rel, auto full_tag) -> bool {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      kc[i] = kn[i];
      vc[i] = vn[i];
    }
    if (rel + kBatch < wg_rows) issue_at(rel + kBatch);
    uint32_t bn[R], rm[R], sl[R];
    bool act[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const uint32_t rr = rel + static_cast<uint32_t>(i * kGblThreads + tid);
      act[i] = FULL || rr < wg_rows;
      if constexpr (HAS_NULLS) {   // null rows: K0
        const int64_t r = lo + rr;
        act[i] = act[i] && gbl_row_streamed<true>(a, r < hi ? r : hi - 1);
      }
      const uint32_t d = static_cast<uint32_t>(kc[i] - a.kmin);
      if (act[i] && d >= span) {   // outside the sampled range: counted here, consumed by gbl_outliers_kernel
        act[i] = false;
        ++outliers;
      }
      gbl_split(a, d < span ? d : 0u, bn[i], rm[i]);
    }
    uint32_t cb0 = 0, cr0 = 0, cb1 = 0, cr1 = 0, cs0 = kGblSkip, cs1 = kGblSkip;
    if (np > 0) gbl_split(a, static_cast<uint32_t>(pk0 - a.kmin), cb0, cr0);
    if (np > 1) gbl_split(a, static_cast<uint32_t>(pk1 - a.kmin), cb1, cr1);
#pragma unroll
    for (int i = 0; i < R; ++i) sl[i] = act[i] ? atomicAdd(&fill[bn[i]], 1u) : kGblSkip;
    if (np > 0) cs0 = atomicAdd(&fill[cb0], 1u);
    if (np > 1) cs1 = atomicAdd(&fill[cb1], 1u);
    uint32_t pend = 0, cpend = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (sl[i] < static_cast<uint32_t>(kGblCap)) place(bn[i], sl[i], rm[i], vc[i]);
      else if (sl[i] != kGblSkip) pend |= 1u << i;
    }
    if (cs0 < static_cast<uint32_t>(kGblCap)) place(cb0, cs0, cr0, pv0);
    else if (cs0 != kGblSkip) cpend |= 1u;
    if (cs1 < static_cast<uint32_t>(kGblCap)) place(cb1, cs1, cr1, pv1);
    else if (cs1 != kGblSkip) cpend |= 2u;
    if (__builtin_popcount(pend) + __builtin_popcount(cpend) > 2) again[cur] = 1;
    uint32_t go = flush_phase();
    int rounds = 0;
    while (go) {   // (workgroup-uniform) some thread holds more than two rows: rounds until nobody holds any
      uint32_t still = 0, cstill = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if ((pend >> i) & 1u) {
          const uint32_t s = atomicAdd(&fill[bn[i]], 1u);
          if (s < static_cast<uint32_t>(kGblCap)) place(bn[i], s, rm[i], vc[i]);
          else still |= 1u << i;
        }
      }
      if (cpend & 1u) {
        const uint32_t s = atomicAdd(&fill[cb0], 1u);
        if (s < static_cast<uint32_t>(kGblCap)) place(cb0, s, cr0, pv0);
        else cstill |= 1u;
      }
      if (cpend & 2u) {
        const uint32_t s = atomicAdd(&fill[cb1], 1u);
        if (s < static_cast<uint32_t>(kGblCap)) place(cb1, s, cr1, pv1);
        else cstill |= 2u;
      }
      pend = still;
      cpend = cstill;
      if (pend | cpend) again[cur] = 1;
      go = flush_phase();
      if (++rounds > 64) return false;   // a hot key: 12 rows per round would take forever — the other plans take the rows
    }
    // what is still pending rides along with the next batch
    int32_t nk0 = 0, nk1 = 0;
    int64_t nv0 = 0, nv1 = 0;
    uint32_t c = 0;
    auto push = [&](int32_t k, int64_t v) {
      if (c == 0) {
        nk0 = k;
        nv0 = v;
      } else {
        nk1 = k;
        nv1 = v;
      }
      ++c;
    };
    if (cpend & 1u) push(pk0, pv0);
    if (cpend & 2u) push(pk1, pv1);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if ((pend >> i) & 1u) push(kc[i], vc[i]);
    }
    pk0 = nk0;
    pv0 = nv0;
    pk1 = nk1;
    pv1 = nv1;
    np = c < 2 ? c : 2;
    return true;
  };
  issue_at(0);
  for (uint32_t rel = 0; rel < wg_rows && !gave_up; rel += kBatch) {
    if (rel + kBatch <= wg_rows) gave_up = !batch(rel, std::true_type{});
    else gave_up = !batch(rel, std::false_type{});
  }
  if (gave_up) {   // (workgroup-uniform)
    if (tid == 0) atomicOr(&a.flags[1], 1u);
    return;
  }
  // drain the carried rows
  for (int rounds = 0;; ++rounds) {
    uint32_t c = 0;
    int32_t nk0 = 0, nk1 = 0;
    int64_t nv0 = 0, nv1 = 0;
    auto retry = [&](int32_t k, int64_t v) {
      uint32_t b, r;
      gbl_split(a, static_cast<uint32_t>(k - a.kmin), b, r);
      const uint32_t s = atomicAdd(&fill[b], 1u);
      if (s < static_cast<uint32_t>(kGblCap)) {
        place(b, s, r, v);
        return;
      }
      if (c == 0) {
        nk0 = k;
        nv0 = v;
      } else {
        nk1 = k;
        nv1 = v;
      }
      ++c;
    };
    if (np > 0) retry(pk0, pv0);
    if (np > 1) retry(pk1, pv1);
    pk0 = nk0;
    pv0 = nv0;
    pk1 = nk1;
    pv1 = nv1;
    np = c;
    if (np != 0) again[cur] = 1;
    if (!flush_phase()) break;
    if (rounds > 4096) {
      if (tid == 0) atomicOr(&a.flags[1], 1u);
      return;
    }
  }
  if (rf0) state[tid] = (rv0 << 1) | 1u;
  if (rf1) state[tid + kGblThreads] = (rv1 << 1) | 1u;
  __syncthreads();
  // the workgroup's partial lines (with their counts), then empty lines up to the end of its last chunk of every bin
  const int sub = tid & 7;
  for (int b0 = 0; b0 < a.bins; b0 += kGblThreads / 8) {   // (workgroup-uniform trip count: the shuffles below)
    const int b = b0 + (tid >> 3);
    const bool on = b < a.bins;
    const uint32_t f = on ? fill[b] : 0u;
    uint32_t line = 0;
    if (on && f != 0 && sub == 0) line = next_line(static_cast<uint32_t>(b));
    line = __shfl(line, lane & ~7, 64);
    if (on && f != 0) store_piece(line, sub, piece(static_cast<uint32_t>(b), sub, f < 6 ? f : 6, f < 6 ? 0 : f - 6));
    uint32_t s = kGblNever;
    if (on && sub == 0) s = state[b];
    s = __shfl(s, lane & ~7, 64);
    if ((s & 1u) != 0) {   // the rest of the workgroup's last chunk
      const uint32_t first = s >> 1, left = K - first % K;
      for (uint32_t l = 0; l < left; ++l) {
        arx_u32x4 z = {0, 0, 0, 0};
        store_piece(first + l, sub, z);
      }
    }
  }
  outliers = wave_reduce_sum_u32(outliers);
  if (lane == 0 && outliers != 0) atomicAdd(&a.flags[2], outliers);
}

// ---- after the scatter: did every bin stay inside its room?  The aggregate's work units (<= unit_lines lines, a bin's
// units of equal length).
__global__ __launch_bounds__(kGblThreads) void gbl_scan_kernel(GblArgs a) {
  __shared__ uint32_t wave_tot[kGblThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (a.bins + kGblThreads - 1) / kGblThreads;
  uint32_t units[2] = {0, 0};
  uint32_t mine = 0;
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    if (b < a.bins) {
      const uint32_t end = a.cursor[b * kGblCursorStride];
      if (end > a.room_start[b + 1]) atomicOr(&a.flags[0], 1u);
      const uint32_t used = end - a.room_start[b];
      units[k] = (used + a.unit_lines - 1) / a.unit_lines;
      mine += units[k];
    }
  }
  const uint32_t incl = wave_inclusive_scan_u32(mine);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t pre = incl - mine;
  for (int k = 0; k < wave; ++k) pre += wave_tot[k];
  for (int k = 0; k < per; ++k) {
    const int b = tid * per + k;
    if (b < a.bins) {
      a.unit_start[b] = pre;
      pre += units[k];
    }
  }
  if (tid == kGblThreads - 1) a.unit_start[a.bins] = pre;
}

// ---- rows whose key lies outside the sampled range (counted by the scatter): straight into the HBM table
template <bool HAS_NULLS>
__global__ __launch_bounds__(kBlock) void gbl_outliers_kernel(GroupbyView v, GblArgs a) {
  const uint32_t span = static_cast<uint32_t>(a.bins) * static_cast<uint32_t>(a.width);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  uint32_t fresh = 0;
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; r < a.n; r += stride) {
    const int32_t key = a.keys[r];
    if (static_cast<uint32_t>(key - a.kmin) < span) continue;
    if (!gbl_row_streamed<HAS_NULLS>(a, r)) continue;
    const int64_t slot = gb_find_or_insert(v, key, &fresh);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
    } else {
      atomicAdd(&v.sums[slot], static_cast<unsigned long long>(a.values[r]));
      atomicAdd(&v.counts[slot], 1ull);
    }
  }
  gb_publish_new_groups(v, fresh);
}

// ---- the direct-indexed LDS aggregate over a bin's lines
constexpr int kGblAggX = 4;   // 16-byte pieces a thread keeps in flight
__global__ __launch_bounds__(kGblThreads) void gbl_aggregate_kernel(GblArgs a) {
  __shared__ unsigned long long sums[kGblMaxWidth];
  __shared__ uint32_t cnts[kGblMaxWidth];
  __shared__ uint32_t unit_bin, unit_first, unit_end;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t u = blockIdx.x;
  if (u >= a.unit_start[a.bins]) return;   // over-provisioned grid (workgroup-uniform)
  if (tid == 0) {   // the bin of unit u: the last b with unit_start[b] <= u
    int lo = 0, hi = a.bins - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.unit_start[mid] <= u) lo = mid;
      else hi = mid - 1;
    }
    const uint32_t first = a.room_start[lo], used = a.cursor[lo * kGblCursorStride] - first;
    const uint32_t nu = a.unit_start[lo + 1] - a.unit_start[lo], k = u - a.unit_start[lo];
    const uint32_t per = ((used + nu - 1) / nu + 7u) & ~7u;   // equal units, whole kilobytes
    unit_bin = static_cast<uint32_t>(lo);
    unit_first = first + (k * per < used ? k * per : used);
    unit_end = first + ((k + 1) * per < used ? (k + 1) * per : used);
  }
  for (int i = tid; i < a.width; i += kGblThreads) {
    sums[i] = 0;
    cnts[i] = 0;
  }
  __syncthreads();
  const uint32_t bin = unit_bin, l0 = unit_first, l1 = unit_end;
  const arx_u32x4* src = reinterpret_cast<const arx_u32x4*>(a.lines + static_cast<size_t>(l0) * 128);
  const int64_t npieces = static_cast<int64_t>(l1 - l0) * 8;
  const int q = tid & 3;
  auto consume = [&](arx_u32x4 d, bool ok) {
    const int from = lane | 3;   // the quad's fourth lane holds the half line's six remainders and its count
    const uint32_t r0 = __shfl(d[0], from, 64), r1 = __shfl(d[1], from, 64), r2 = __shfl(d[2], from, 64);
    const uint32_t cnt = __shfl(d[3], from, 64);
    if (!ok || q == 3) return;
    const uint32_t rr = q == 0 ? r0 : q == 1 ? r1 : r2;
    if (static_cast<uint32_t>(2 * q) < cnt) {
      const uint32_t rem = rr & 0xFFFFu;
      atomicAdd(&sums[rem], (static_cast<unsigned long long>(d[1]) << 32) | d[0]);
      atomicAdd(&cnts[rem], 1u);
    }
    if (static_cast<uint32_t>(2 * q + 1) < cnt) {
      const uint32_t rem = rr >> 16;
      atomicAdd(&sums[rem], (static_cast<unsigned long long>(d[3]) << 32) | d[2]);
      atomicAdd(&cnts[rem], 1u);
    }
  };
  if (npieces > 0) {
    constexpr int X = kGblAggX;
    arx_u32x4 nxt[X], curd[X];
    auto issue = [&](int64_t p0) {
#pragma unroll
      for (int x = 0; x < X; ++x) {
        const int64_t p = p0 + static_cast<int64_t>(x) * kGblThreads + tid;
        nxt[x] = __builtin_nontemporal_load(src + (p < npieces ? p : npieces - 8 + (tid & 7)));
      }
    };
    issue(0);
    for (int64_t p0 = 0; p0 < npieces; p0 += static_cast<int64_t>(X) * kGblThreads) {
#pragma unroll
      for (int x = 0; x < X; ++x) curd[x] = nxt[x];
      if (p0 + static_cast<int64_t>(X) * kGblThreads < npieces) issue(p0 + static_cast<int64_t>(X) * kGblThreads);
#pragma unroll
      for (int x = 0; x < X; ++x) consume(curd[x], p0 + static_cast<int64_t>(x) * kGblThreads + tid < npieces);
    }
  }
  __syncthreads();
  // the unit's partial aggregates into the partition's slice of the DENSE state: consecutive lanes, consecutive addresses
  // (a wave's 64 adds are a few line-sized requests, not 64 — what the per-group table flush of the other plans costs)
  unsigned long long* dsum = reinterpret_cast<unsigned long long*>(a.dense + static_cast<size_t>(bin) * a.width * 16);
  unsigned long long* dcnt = dsum + a.width;
  for (int i = tid; i < a.width; i += kGblThreads) {
    const uint32_t c = cnts[i];
    if (c == 0) continue;
    atomicAdd(&dsum[i], sums[i]);
    atomicAdd(&dcnt[i], static_cast<unsigned long long>(c));
  }
}

// ---- the dense state's groups into the HBM table (the table API: arx_groupby_sum_i64_consume)
__global__ __launch_bounds__(kBlock) void gbl_table_insert_kernel(GroupbyView v, GblArgs a) {
  const int64_t slots = static_cast<int64_t>(a.bins) * a.width;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  uint32_t fresh = 0;
  for (int64_t s = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; s < slots; s += stride) {
    const int64_t bin = s / a.width, i = s - bin * a.width;
    const unsigned long long* base = reinterpret_cast<const unsigned long long*>(a.dense + static_cast<size_t>(bin) * a.width * 16);
    const unsigned long long c = base[a.width + i];
    if (c == 0) continue;
    const int64_t slot = gb_find_or_insert(v, a.kmin + static_cast<int32_t>(s), &fresh);
    if (slot < 0) {
      atomicExch(&v.hdr->overflow, 1u);
      continue;
    }
    atomicAdd(&v.sums[slot], base[i]);
    atomicAdd(&v.counts[slot], c);
  }
  gb_publish_new_groups(v, fresh);
}

// ---- finalize of a dense state: the groups of partitions [first, first + count) in KEY ORDER.  Pass 1 counts the
// non-empty slots of every 4096-slot tile, one workgroup scans the tiles, pass 2 writes.
constexpr int kGblTile = 4096;
__global__ __launch_bounds__(kBlock) void gbl_finalize_count_kernel(const uint8_t* __restrict__ dense, int width, int64_t slots,
                                                                    unsigned long long* __restrict__ tile_counts) {
  __shared__ uint32_t part[kWavesPerBlock];
  const int64_t s0 = static_cast<int64_t>(blockIdx.x) * kGblTile;
  uint32_t mine = 0;
  for (int j = threadIdx.x; j < kGblTile; j += kBlock) {
    const int64_t s = s0 + j;
    if (s < slots) {
      const int64_t bin = s / width, i = s - bin * width;
      mine += reinterpret_cast<const unsigned long long*>(dense + static_cast<size_t>(bin) * width * 16)[width + i] != 0 ? 1u : 0u;
    }
  }
  mine = wave_reduce_sum_u32(mine);
  if (lane_id() == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += part[w];
    tile_counts[blockIdx.x] = t;
  }
}

// exclusive scan of the tile counts in place (one workgroup); the total lands in *out_total
__global__ __launch_bounds__(1024) void gbl_finalize_scan_kernel(unsigned long long* __restrict__ tile_counts, int64_t ntiles,
                                                                 int64_t* __restrict__ out_total) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t t0 = 0; t0 < ntiles; t0 += 1024) {
    const int64_t t = t0 + tid;
    const unsigned long long c = t < ntiles ? tile_counts[t] : 0ull;
    const unsigned long long incl = wave_inclusive_scan_u64(c);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long pre = carry + incl - c;
    for (int k = 0; k < wave; ++k) pre += wave_tot[k];
    if (t < ntiles) tile_counts[t] = pre;
    __syncthreads();
    if (tid == 1023) carry = pre + c;
    __syncthreads();
  }
  if (tid == 0) *out_total = static_cast<int64_t>(carry);
}

__global__ __launch_bounds__(kBlock) void gbl_finalize_emit_kernel(const uint8_t* __restrict__ dense, int width, int64_t slots, int32_t key0,
                                                                   unsigned long long min_count, const unsigned long long* __restrict__ tile_starts,
                                                                   int32_t* __restrict__ out_keys, int64_t* __restrict__ out_sums,
                                                                   int64_t* __restrict__ out_counts, uint8_t* __restrict__ out_valid) {
  __shared__ uint32_t wave_base[kWavesPerBlock];
  __shared__ uint32_t run;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t s0 = static_cast<int64_t>(blockIdx.x) * kGblTile;
  if (threadIdx.x == 0) run = 0;
  __syncthreads();
  const unsigned long long base = tile_starts[blockIdx.x];
  for (int j0 = 0; j0 < kGblTile; j0 += kBlock) {   // (workgroup-uniform trip count: ballots + barriers below)
    const int64_t s = s0 + j0 + threadIdx.x;
    unsigned long long c = 0, sum = 0;
    if (s < slots) {
      const int64_t bin = s / width, i = s - bin * width;
      const unsigned long long* b = reinterpret_cast<const unsigned long long*>(dense + static_cast<size_t>(bin) * width * 16);
      c = b[width + i];
      if (c != 0) sum = b[i];
    }
    const uint64_t live = __ballot(c != 0);
    if (lane == 0) wave_base[wave] = static_cast<uint32_t>(__popcll(live));
    __syncthreads();
    uint32_t before = run;
    for (int w = 0; w < wave; ++w) before += wave_base[w];
    if (c != 0) {
      const unsigned long long at = base + before + static_cast<uint32_t>(__popcll(live & ((uint64_t(1) << lane) - 1)));
      out_keys[at] = key0 + static_cast<int32_t>(s);
      out_sums[at] = static_cast<int64_t>(sum);
      if (out_counts != nullptr) out_counts[at] = static_cast<int64_t>(c);
      out_valid[at] = c >= min_count ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int w = 0; w < kWavesPerBlock; ++w) t += wave_base[w];
      run += t;
    }
    __syncthreads();
  }
}

// dst += the same words of `count` other states that lie `stride_words` apart (Merge of the dense state: sums wrap, counts
// add — both 64-bit words; the P blocks a rank received for its partitions are summed in ONE pass)
__global__ __launch_bounds__(kBlock) void gbl_merge_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src,
                                                           int64_t words, int count, int64_t stride_words) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  for (int64_t w = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; w < words; w += stride) {
    unsigned long long acc = dst[w];
    for (int r = 0; r < count; ++r) acc += src[r * stride_words + w];
    dst[w] = acc;
  }
}

// ---- host side
static Knob<int> g_gbl{1};                         // 1: the lines plan where the sampled key range allows it (A/B knob groupby_lines; 0 = off)
static Knob<int64_t> g_gbl_min_rows{int64_t(1) << 22};   // below this the other plans (knob groupby_lines_min_rows)
static Knob<int64_t> g_gbl_sample_rows{int64_t(1) << 24};   // rows the histogram sample reads (knob groupby_lines_sample_rows): 6 sigma of the rooms = 4 % at any n
static Knob<int64_t> g_gbl_range_sample_rows{int64_t(1) << 20};   // rows the key-range sample reads (knob groupby_lines_range_sample_rows)
static Knob<int> g_gbl_unit_rows{1 << 21};         // rows per aggregate work unit (knob groupby_lines_unit_rows)
static Knob<int> g_gbl_wgs{0};                     // workgroups of the scatter (0: one per CU; knob groupby_lines_wgs — tests)
static std::atomic<int64_t> g_gbl_slices{0}, g_gbl_fallbacks{0}, g_gbl_outlier_rows{0}, g_gbl_declined{0};

struct GblPlan {
  int64_t kmin;
  int width, wshift, bins, wgs;
  int64_t sample_stride, rows_per_wg;
  uint32_t total_lines, unit_lines;
  size_t off_lines, off_cursor, off_room_start, off_hist, off_flags, off_unit_start, off_dense, total;
};

// lines the buffer must hold for n rows in `bins` bins scattered by `wgs` workgroups (what gbl_rooms_kernel asks for
// with an even spread — the worst case of its 6-sigma terms; an uneven spread needs no more)
static int64_t gbl_lines_for(int64_t n, int bins, int wgs, int64_t stride) {
  const double per_bin = static_cast<double>(n) / bins;
  const double sigma = stride > 1 ? std::sqrt(per_bin * static_cast<double>(stride) + 1.0) : 0.0;
  const double rows = per_bin + 6.0 * sigma + per_bin / 64.0 + 64.0 * static_cast<double>(stride > 1 ? stride : 0);
  const double lines = rows / kGblCap + static_cast<double>(wgs) * (2 * kGblK) + 2.0 + kGblK;
  return static_cast<int64_t>(lines * bins * 1.002) + 1024;
}

static int gbl_cus() {
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_relaxed);
  if (c == 0) {
    int dev = 0, n = 0;
    c = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    cus.store(c, std::memory_order_relaxed);
  }
  return c;
}

static int64_t gbl_stride_for(int64_t n, int64_t sample_rows) {
  return std::max<int64_t>(1, n / std::max<int64_t>(sample_rows, kGblUnitRows));
}

// Partitions for keys in [kmin, kmax]: the narrowest width (8 ... 8192, 12288) that needs at most kGblMaxBins of them.
// false: the range is too wide, or so narrow that the other plans' single LDS table is the better tool.
static bool gbl_partitions(int64_t kmin, int64_t kmax, int* width, int* wshift, int* bins) {
  const int64_t range = kmax - kmin + 1;
  if (range < 1) return false;
  int w = 8, sh = 3;
  while (sh < 13 && (range + w - 1) / w > kGblMaxBins) {
    ++sh;
    w <<= 1;
  }
  if ((range + w - 1) / w > kGblMaxBins) {
    w = kGblMaxWidth;
    sh = 0;
  }
  const int64_t b = (range + w - 1) / w;
  if (b > kGblMaxBins || b < kGblMinBins) return false;
  *width = w;
  *wshift = sh;
  *bins = static_cast<int>(b);
  return true;
}

// the scratch layout of one pass over n rows (with_dense: the table path keeps the dense state in the scratch too)
static bool gbl_plan(int64_t n, int64_t kmin, int width, int wshift, int bins, bool with_dense, GblPlan* p) {
  p->kmin = kmin;
  p->width = width;
  p->wshift = wshift;
  p->bins = bins;
  const int64_t batch = int64_t(kGblR) * kGblThreads;
  int wgs = g_gbl_wgs > 0 ? int(g_gbl_wgs) : gbl_cus();
  wgs = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(wgs, n / (int64_t(1) << 18))));   // >= 2^18 rows per workgroup: chunk tails stay small
  p->wgs = wgs;
  p->rows_per_wg = ceil_div(ceil_div(n, wgs), batch) * batch;
  if (p->rows_per_wg >= (int64_t(1) << 31)) return false;   // (the scatter addresses a workgroup's rows by 32-bit offsets)
  p->sample_stride = gbl_stride_for(n, g_gbl_sample_rows);
  const int64_t lines = gbl_lines_for(n, bins, wgs, p->sample_stride);
  if (lines >= (int64_t(1) << 31) - 4096) return false;   // (the scatter keeps line << 1 | flag in 32 bits)
  p->total_lines = static_cast<uint32_t>(lines);
  p->unit_lines = static_cast<uint32_t>(std::max<int64_t>(64, int64_t(g_gbl_unit_rows) / kGblCap));
  auto align = [](size_t x) { return (x + 255) & ~size_t(255); };
  size_t o = 0;
  p->off_lines = o; o = align(o + static_cast<size_t>(lines) * 128);
  p->off_cursor = o; o = align(o + static_cast<size_t>(kGblMaxBins) * kGblCursorStride * 4);
  p->off_room_start = o; o = align(o + (kGblMaxBins + 1) * 4);
  p->off_hist = o; o = align(o + kGblMaxBins * 4);
  p->off_flags = o; o = align(o + 16);
  p->off_unit_start = o; o = align(o + (kGblMaxBins + 1) * 4);
  p->off_dense = o; o = align(o + (with_dense ? static_cast<size_t>(bins) * width * 16 : 0));
  p->total = o;
  return true;
}

// scratch the lines plan may ask for at most for `n` rows (any key range it accepts; with_dense: + the dense state)
static size_t gbl_workspace_bytes(int64_t n, bool with_dense) {
  if (!g_gbl || n < 1) return 0;
  const int wgs = g_gbl_wgs > 0 ? int(g_gbl_wgs) : gbl_cus();
  const int64_t lines = gbl_lines_for(n, kGblMaxBins, wgs, gbl_stride_for(n, g_gbl_sample_rows));
  return static_cast<size_t>(lines) * 128 + (size_t(kGblMaxBins) * kGblCursorStride * 4 + 4 * (kGblMaxBins + 1) * 4 + 4096) +
         (with_dense ? static_cast<size_t>(kGblMaxBins) * kGblMaxWidth * 16 + 256 : 0);
}

constexpr int kGblDeclined = -2000;   // the plan does not apply / gave up — nothing was consumed

static void gbl_bind(GblArgs& a, const GblPlan& plan, uint8_t* w) {
  a.rows_per_wg = plan.rows_per_wg;
  a.kmin = static_cast<int32_t>(plan.kmin);
  a.wshift = plan.wshift;
  a.width = plan.width;
  a.bins = plan.bins;
  a.sample_stride = plan.sample_stride;
  a.wgs = plan.wgs;
  a.total_lines = plan.total_lines;
  a.unit_lines = plan.unit_lines;
  a.lines = w + plan.off_lines;
  a.cursor = reinterpret_cast<uint32_t*>(w + plan.off_cursor);
  a.room_start = reinterpret_cast<uint32_t*>(w + plan.off_room_start);
  a.hist = reinterpret_cast<uint32_t*>(w + plan.off_hist);
  a.flags = reinterpret_cast<uint32_t*>(w + plan.off_flags);
  a.unit_start = reinterpret_cast<uint32_t*>(w + plan.off_unit_start);
}

// sampled histogram -> rooms -> scatter -> room check + work units; the flags come back to the host (ONE read-back).
// Nothing but the scratch has been written when this returns.
template <bool HAS_NULLS>
static int gbl_partition(const GblArgs& a, uint32_t flags[4], hipStream_t st) {
  ARX_HIP(hipMemsetAsync(a.hist, 0, static_cast<size_t>(kGblMaxBins) * 4, st));
  ARX_HIP(hipMemsetAsync(a.flags, 0, 16, st));
  const int64_t strata = ceil_div(ceil_div(a.n, kGblUnitRows), a.sample_stride);
  // (every workgroup folds its LDS histogram into the global one: bins atomics per workgroup on the same `bins` addresses)
  const unsigned hgrid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(strata, (kGblThreads / 64) * 8), 256)));
  hipLaunchKernelGGL((gbl_hist_kernel<HAS_NULLS>), dim3(hgrid), dim3(kGblThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbl_hist_kernel");
  hipLaunchKernelGGL(gbl_rooms_kernel, dim3(1), dim3(kGblThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbl_rooms_kernel");
  hipLaunchKernelGGL((gbl_scatter_kernel<HAS_NULLS>), dim3(static_cast<unsigned>(a.wgs)), dim3(kGblThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbl_scatter_kernel");
  hipLaunchKernelGGL(gbl_scan_kernel, dim3(1), dim3(kGblThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbl_scan_kernel");
  ARX_HIP(hipMemcpyAsync(flags, a.flags, 16, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  return ARX_OK;
}

static int gbl_aggregate(const GblArgs& a, hipStream_t st) {
  const int64_t max_units = ceil_div(static_cast<int64_t>(a.total_lines), a.unit_lines) + a.bins;
  hipLaunchKernelGGL(gbl_aggregate_kernel, dim3(static_cast<unsigned>(max_units)), dim3(kGblThreads), 0, st, a);
  ARX_CHECK_LAUNCH("gbl_aggregate_kernel");
  return ARX_OK;
}

// the sampled {min, max} of a key column into the caller's device pair (folded with atomic min / max)
static int gbl_sample_range(const int32_t* k, int64_t n, int64_t sample_rows, int32_t* out_pair, hipStream_t st) {
  const int64_t stride = gbl_stride_for(n, sample_rows);
  const int64_t strata = ceil_div(ceil_div(n, kGblUnitRows), stride);
  const unsigned rgrid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(ceil_div(strata, kWavesPerBlock * 8), 512)));
  hipLaunchKernelGGL(gbl_range_kernel, dim3(rgrid), dim3(kBlock), 0, st, k, n, stride, out_pair);
  ARX_CHECK_LAUNCH("gbl_range_kernel");
  return ARX_OK;
}

// All n rows through the lines plan into the HBM TABLE, or kGblDeclined (nothing consumed: the other plans take the rows).
template <bool HAS_NULLS>
static int gbl_try(const GroupbyView& v, const int32_t* k, const int64_t* val, Bits kb, Bits vb, int64_t n, uint8_t* w, size_t ws_bytes,
                   hipStream_t st) {
  // 1. the sampled key range (the head of the scratch holds the two words until the plan is bound)
  int32_t* range = reinterpret_cast<int32_t*>(w);
  const int32_t init[2] = {INT32_MAX, INT32_MIN};
  ARX_HIP(hipMemcpyAsync(range, init, 8, hipMemcpyHostToDevice, st));
  const int rrc = gbl_sample_range(k, n, g_gbl_range_sample_rows, range, st);
  if (rrc != ARX_OK) return rrc;
  int32_t mm[2];
  ARX_HIP(hipMemcpyAsync(mm, range, 8, hipMemcpyDeviceToHost, st));
  ARX_HIP(hipStreamSynchronize(st));
  if (mm[0] > mm[1]) return kGblDeclined;
  // (a sample misses a few keys at both ends of a range: 1/64 of the range on either side)
  const bool sampled = gbl_stride_for(n, g_gbl_range_sample_rows) > 1;
  const int64_t pad = sampled ? (static_cast<int64_t>(mm[1]) - mm[0] + 1) / 64 + 1 : 0;
  const int64_t kmin = std::max<int64_t>(INT32_MIN, static_cast<int64_t>(mm[0]) - pad);
  const int64_t kmax = std::min<int64_t>(INT32_MAX, static_cast<int64_t>(mm[1]) + pad);
  GblPlan plan{};
  int width = 0, wshift = 0, bins = 0;
  if (!gbl_partitions(kmin, kmax, &width, &wshift, &bins) || !gbl_plan(n, kmin, width, wshift, bins, true, &plan) || plan.total > ws_bytes) {
    g_gbl_declined.fetch_add(1, std::memory_order_relaxed);
    return kGblDeclined;
  }
  GblArgs a{};
  a.keys = k;
  a.values = val;
  a.kvalid = kb;
  a.vvalid = vb;
  a.n = n;
  gbl_bind(a, plan, w);
  a.dense = w + plan.off_dense;
  uint32_t flags[4] = {0, 0, 0, 0};
  const int prc = gbl_partition<HAS_NULLS>(a, flags, st);
  if (prc != ARX_OK) return prc;
  if (flags[0] != 0 || flags[1] != 0 || static_cast<int64_t>(flags[2]) > n / 64) {
    g_gbl_fallbacks.fetch_add(1, std::memory_order_relaxed);
    return kGblDeclined;   // nothing has touched the table
  }
  // 2. the point of no return: null rows, outliers, the aggregate into the dense state, its groups into the table
  g_gbl_slices.fetch_add(1, std::memory_order_relaxed);
  if (HAS_NULLS) {
    GbpArgs na{};
    na.keys = k;
    na.values = val;
    na.kvalid = kb;
    na.vvalid = vb;
    na.n = n;
    hipLaunchKernelGGL(gbp_null_rows_kernel, dim3(gb_grid(n / 8 + 1)), dim3(kBlock), 0, st, v, na);
    ARX_CHECK_LAUNCH("gbp_null_rows_kernel");
  }
  if (flags[2] != 0) {
    g_gbl_outlier_rows.fetch_add(flags[2], std::memory_order_relaxed);
    hipLaunchKernelGGL((gbl_outliers_kernel<HAS_NULLS>), dim3(gb_grid(n / 8 + 1)), dim3(kBlock), 0, st, v, a);
    ARX_CHECK_LAUNCH("gbl_outliers_kernel");
  }
  const size_t dense_bytes = static_cast<size_t>(plan.bins) * plan.width * 16;
  ARX_HIP(hipMemsetAsync(a.dense, 0, dense_bytes, st));
  const int arc = gbl_aggregate(a, st);
  if (arc != ARX_OK) return arc;
  hipLaunchKernelGGL(gbl_table_insert_kernel, dim3(gb_grid(static_cast<int64_t>(plan.bins) * plan.width)), dim3(kBlock), 0, st, v, a);
  ARX_CHECK_LAUNCH("gbl_table_insert_kernel");
  return ARX_OK;
}

static int get_groupby_lines_counter(const char* name, int64_t* out) {
  if (strcmp(name, "groupby_slices_lines") == 0) *out = g_gbl_slices.load();
  else if (strcmp(name, "groupby_lines_fallbacks") == 0) *out = g_gbl_fallbacks.load();
  else if (strcmp(name, "groupby_lines_declined") == 0) *out = g_gbl_declined.load();
  else if (strcmp(name, "groupby_lines_outlier_rows") == 0) *out = g_gbl_outlier_rows.load();
  else return 0;
  return 1;
}

static int set_groupby_lines_option(const char* name, int64_t value) {
  if (strcmp(name, "groupby_lines") == 0) g_gbl = value != 0 ? 1 : 0;
  else if (strcmp(name, "groupby_lines_min_rows") == 0) g_gbl_min_rows = std::max<int64_t>(1, value);
  else if (strcmp(name, "groupby_lines_sample_rows") == 0) g_gbl_sample_rows = std::max<int64_t>(kGblUnitRows, value);
  else if (strcmp(name, "groupby_lines_range_sample_rows") == 0) g_gbl_range_sample_rows = std::max<int64_t>(kGblUnitRows, value);
  else if (strcmp(name, "groupby_lines_unit_rows") == 0) g_gbl_unit_rows = static_cast<int>(std::max<int64_t>(768, std::min<int64_t>(value, 1 << 30)));
  else if (strcmp(name, "groupby_lines_wgs") == 0) g_gbl_wgs = static_cast<int>(std::max<int64_t>(0, std::min<int64_t>(value, 4096)));
  else return 0;
  return 1;
}
