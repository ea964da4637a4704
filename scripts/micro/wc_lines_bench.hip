// Round 6: a flat radix level whose EVERY global store is a whole 128-byte line (software write-combining), for the
// group-by's dense-range plan (DESIGN 4.6 "Round 6").  Round 2's chunk_write_bench said what decides a scatter's
// rate: aligned whole lines write at the linear rate (5 TB/s), partial lines at 2.1 - 3.6.  The register-tile flat
// level (gbp_scatter_wide) writes unaligned 144-byte runs: 1.21x write amplification, 3.6 TB/s.
//   lines   : a partition's records live in 128-byte LINES = two 64-byte halves of {6 x u64 value, 6 x u16 key
//             remainder, u32 count}: 12 records / line = 10.67 B / record, self-contained (a line never mixes bins).
//   scatter : persistent workgroups (one per CU), ONE line per bin in LDS.  A row takes its slot with a returning LDS
//             atomic on the line's fill counter; the row that takes slot 11 queues the bin; after a barrier 8 lanes copy
//             a full line out (16 B each), its place in the bin's room from a per-(workgroup, bin) chunk of K lines (one
//             global atomic per K lines).  Rows that found their line full retry after the flush.
//   agg     : quads of lanes read a half line (3 x 2 values + remainders/count), direct-indexed LDS table
//             (ds_add_u64 + ds_add_u32), no tags, no probing.
//   ceiling : linewrite = the memory side alone (read 12 B/row, write whole lines to B pseudo-random frontiers).
//   usage: wc_lines_bench [log2 rows=30] [key range=10000000] [width=12288]
//   build: hipcc --offload-arch=gfx950 -O3 -o build/wc_lines_bench scripts/micro/wc_lines_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024;
constexpr int kMaxBins = 1216;          // 1216 x 128 B = 152 KB of line buffers
constexpr int kCap = 12;                // records per line
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kCStrideMax = 32;   // u32 between two bins' cursors: 32 = one 128-byte line each (1 = packed: single-lane atomics then queue up behind ~26 lines)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void fill(int32_t* k, int64_t* v, int64_t n, uint32_t range) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t z = mix(0x1234 + (uint64_t)i * 0x9E3779B97F4A7C15ull);
    k[i] = (int32_t)((z >> 32) % range);
    v[i] = (int64_t)mix(z);
  }
}

// bin / remainder of a key offset `d` (>= 0): width 8192 (wsel 0) or 12288 (wsel 1: (d >> 12) / 3, exact below 2^28)
__device__ __forceinline__ void split(uint32_t d, int wsel, uint32_t& bin, uint32_t& rem) {
  if (wsel == 0) { bin = d >> 13; rem = d & 8191u; }
  else { bin = ((d >> 12) * 0xAAABu) >> 17; rem = d - bin * 12288u; }
}

struct ScatterArgs {
  const int32_t* keys; const int64_t* vals; int64_t n; int64_t rows_per_wg;
  int32_t kmin; int wsel; int bins; uint32_t room_lines;
  uint32_t* cursor;      // [bins] lines reserved in every room (multiples of K)
  uint8_t* lines;        // bins x room_lines x 128 B
  uint32_t* flags;       // [0] room overflow, [1] skew give-up
  int nt;                // bit 0: nt loads, bit 1: nt stores
  int cstride;           // u32 between two bins' cursors
};

template <int R, int K>
__global__ __launch_bounds__(kThreads) void wc_scatter(ScatterArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t L[kMaxBins * 128];
  __shared__ uint16_t list[2][kMaxBins];
  __shared__ uint32_t nlist[2];
  const int tid = threadIdx.x, lane = tid & 63;
  auto fillp = [&](uint32_t b) { return reinterpret_cast<uint32_t*>(L + b * 128 + 60); };
  auto statep = [&](uint32_t b) { return reinterpret_cast<uint32_t*>(L + b * 128 + 124); };
  for (int b = tid; b < a.bins; b += kThreads) { *fillp(b) = 0; *statep(b) = kNone; }
  if (tid < 2) nlist[tid] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * a.rows_per_wg;
  const int64_t hi = lo + a.rows_per_wg < a.n ? lo + a.rows_per_wg : a.n;
  if (lo >= hi) return;
  int32_t kc[R], kn[R];
  int64_t vc[R], vn[R];
  auto issue = [&](int64_t r0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t r = r0 + i * kThreads + tid;
      r = r < hi ? r : hi - 1;
      if (a.nt & 1) { kn[i] = __builtin_nontemporal_load(a.keys + r); vn[i] = __builtin_nontemporal_load(a.vals + r); }
      else { kn[i] = a.keys[r]; vn[i] = a.vals[r]; }
    }
  };
  int cur = 0;
  // one row into its bin's line; false = the line is full (retry after the flush)
  auto append = [&](int32_t key, int64_t val) -> bool {
    uint32_t bin, rem;
    split((uint32_t)(key - a.kmin), a.wsel, bin, rem);
    const uint32_t slot = atomicAdd(fillp(bin), 1u);
    if (slot >= (uint32_t)kCap) return false;
    const uint32_t h = slot >= 6 ? 1u : 0u, j = slot - 6 * h;
    uint8_t* base = L + bin * 128 + h * 64;
    *reinterpret_cast<uint64_t*>(base + j * 8) = (uint64_t)val;
    *reinterpret_cast<uint16_t*>(base + 48 + j * 2) = (uint16_t)rem;
    if (slot == kCap - 1) list[cur][atomicAdd(&nlist[cur], 1u)] = (uint16_t)bin;
    return true;
  };
  // the place of a bin's next line in its room (lane `leader` of the 8 that copy the line decides)
  auto next_line = [&](uint32_t bin) -> uint32_t {
    uint32_t s = *statep(bin);
    if (s == kNone || (s % K) == 0) s = atomicAdd(&a.cursor[bin * a.cstride], (uint32_t)K);
    *statep(bin) = s + 1;
    return s;
  };
  auto store_piece = [&](uint32_t bin, uint32_t line, int sub, u32x4 d) {
    if (line >= a.room_lines) { atomicOr(&a.flags[0], 1u); return; }
    u32x4* dst = reinterpret_cast<u32x4*>(a.lines + ((size_t)bin * a.room_lines + line) * 128 + sub * 16);
    if (a.nt & 2) __builtin_nontemporal_store(d, dst); else *dst = d;
  };
  issue(lo);
  for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)R * kThreads) {
#pragma unroll
    for (int i = 0; i < R; ++i) { kc[i] = kn[i]; vc[i] = vn[i]; }
    if (r0 + (int64_t)R * kThreads < hi) issue(r0 + (int64_t)R * kThreads);
    uint32_t pend = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (r0 + i * kThreads + tid < hi && !append(kc[i], vc[i])) pend |= 1u << i;
    }
    for (int round = 0;; ++round) {
      const int any = __syncthreads_or(pend != 0);
      const uint32_t nf = nlist[cur];
      if (tid == 0) nlist[cur ^ 1] = 0;
      const int sub = tid & 7;
      for (uint32_t g = tid >> 3; g < nf; g += kThreads / 8) {
        const uint32_t bin = list[cur][g];
        uint32_t line = 0;
        if (sub == 0) line = next_line(bin);
        line = __shfl(line, lane & ~7, 64);
        u32x4 d = *reinterpret_cast<const u32x4*>(L + bin * 128 + sub * 16);
        if ((sub & 3) == 3) d[3] = 6;
        store_piece(bin, line, sub, d);
        if (sub == 0) *fillp(bin) = 0;
      }
      __syncthreads();
      cur ^= 1;
      if (!any) break;
      if (round > 256) { if (tid == 0) atomicOr(&a.flags[1], 1u); return; }
      uint32_t still = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (((pend >> i) & 1u) && !append(kc[i], vc[i])) still |= 1u << i;
      }
      pend = still;
    }
  }
  // the workgroup's partial lines (with their counts), then empty lines up to the end of its last chunk of every bin
  const int sub = tid & 7;
  for (int b = tid >> 3; b < a.bins; b += kThreads / 8) {
    const uint32_t f = *fillp(b);
    uint32_t s = *statep(b);
    if (f != 0) {
      uint32_t line = 0;
      if (sub == 0) line = next_line(b);
      line = __shfl(line, lane & ~7, 64);
      u32x4 d = *reinterpret_cast<const u32x4*>(L + b * 128 + sub * 16);
      if (sub == 3) d[3] = f < 6 ? f : 6;
      if (sub == 7) d[3] = f < 6 ? 0 : f - 6;
      store_piece(b, line, sub, d);
      s = line + 1;
    }
    if (s != kNone) {
      for (; (s % K) != 0; ++s) {
        u32x4 z = {0, 0, 0, 0};
        store_piece(b, s, sub, z);
      }
    }
  }
}

// ---- v2 (after call B: v1 ran 27.5 - 30 ms at 4e9 rows against a 17 ms memory ceiling — it waited on itself):
//  * LDS as separate arrays: v1 kept a bin's line as 128 contiguous bytes, so every fill counter sat on bank 15 (a
//    32-way conflict on every returning atomic) and a value's bank depended on its slot only;
//  * all of a batch's returning LDS atomics are issued before the first dependent store;
//  * a bin's next chunk of K lines is reserved (one returning global atomic) right after the flush that used the
//    chunk's last line and picked up before the next flush: no wave of the flush waits ~2 us for its one lane's atomic;
//  * a row that found its line full is CARRIED into the next batch (two per thread in registers) instead of a second
//    append / barrier / flush / barrier round per batch; a thread with more than two falls back to rounds.
constexpr uint32_t kSkip = 0xFFFFFFFFu;
constexpr uint32_t kNever = 0xFFFFFFE0u;   // state of a bin this workgroup never wrote a line of

template <int R, int K>
__global__ __launch_bounds__(kThreads) void wc_scatter2(ScatterArgs a) {
  __shared__ uint64_t vals[kMaxBins * kCap];
  __shared__ uint16_t rems[kMaxBins * kCap];
  __shared__ uint32_t fill[kMaxBins], state[kMaxBins];   // state = next line << 5 | lines left in the chunk
  __shared__ uint16_t list[2][kMaxBins];
  __shared__ uint32_t nlist[2], again[2];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int b = tid; b < a.bins; b += kThreads) { fill[b] = 0; state[b] = kNever; }
  if (tid < 2) { nlist[tid] = 0; again[tid] = 0; }
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * a.rows_per_wg;
  const int64_t hi = lo + a.rows_per_wg < a.n ? lo + a.rows_per_wg : a.n;
  if (lo >= hi) return;
  int32_t kc[R], kn[R];
  int64_t vc[R], vn[R];
  auto issue = [&](int64_t r0) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      int64_t r = r0 + i * kThreads + tid;
      r = r < hi ? r : hi - 1;
      if (a.nt & 1) { kn[i] = __builtin_nontemporal_load(a.keys + r); vn[i] = __builtin_nontemporal_load(a.vals + r); }
      else { kn[i] = a.keys[r]; vn[i] = a.vals[r]; }
    }
  };
  int cur = 0;
  auto place = [&](uint32_t bin, uint32_t slot, uint32_t rem, int64_t val) {
    vals[bin * kCap + slot] = (uint64_t)val;
    rems[bin * kCap + slot] = (uint16_t)rem;
    if (slot == kCap - 1) list[cur][atomicAdd(&nlist[cur], 1u)] = (uint16_t)bin;
  };
  auto next_line = [&](uint32_t bin) -> uint32_t {
    uint32_t s = state[bin];
    if ((s & 31u) == 0) s = (atomicAdd(&a.cursor[bin * a.cstride], (uint32_t)K) << 5) | (uint32_t)K;
    const uint32_t line = s >> 5;
    state[bin] = ((line + 1) << 5) | ((s & 31u) - 1);
    return line;
  };
  auto store_piece = [&](uint32_t bin, uint32_t line, int sub, u32x4 d) {
    if (line >= a.room_lines) { atomicOr(&a.flags[0], 1u); return; }
    u32x4* dst = reinterpret_cast<u32x4*>(a.lines + ((size_t)bin * a.room_lines + line) * 128 + sub * 16);
    if (a.nt & 2) __builtin_nontemporal_store(d, dst); else *dst = d;
  };
  auto piece = [&](uint32_t bin, int sub, uint32_t c0, uint32_t c1) -> u32x4 {
    const int q = sub & 3, h = sub >> 2;
    if (q < 3) return *reinterpret_cast<const u32x4*>(&vals[bin * kCap + 6 * h + 2 * q]);
    const uint32_t* r = reinterpret_cast<const uint32_t*>(&rems[bin * kCap + 6 * h]);
    u32x4 d = {r[0], r[1], r[2], h ? c1 : c0};
    return d;
  };
  // the chunks this thread's bins (tid, tid + 1024) wait for
  bool rf0 = false, rf1 = false;
  uint32_t rv0 = 0, rv1 = 0;
  auto flush_phase = [&]() -> uint32_t {
    if (rf0) { state[tid] = (rv0 << 5) | (uint32_t)K; rf0 = false; }
    if (rf1) { state[tid + kThreads] = (rv1 << 5) | (uint32_t)K; rf1 = false; }
    __syncthreads();
    const uint32_t go = again[cur];
    const uint32_t nf = nlist[cur];
    if (tid == 0) { nlist[cur ^ 1] = 0; again[cur ^ 1] = 0; }
    const int sub = tid & 7;
    for (uint32_t g = tid >> 3; g < nf; g += kThreads / 8) {
      const uint32_t bin = list[cur][g];
      uint32_t line = 0;
      if (sub == 0) line = next_line(bin);
      line = __shfl(line, lane & ~7, 64);
      store_piece(bin, line, sub, piece(bin, sub, 6, 6));
      if (sub == 0) fill[bin] = 0;
    }
    __syncthreads();
    cur ^= 1;
    if (tid < a.bins) {
      const uint32_t s = state[tid];
      if ((s & 31u) == 0 && s != kNever) { rv0 = atomicAdd(&a.cursor[tid * a.cstride], (uint32_t)K); rf0 = true; }
    }
    if (tid + kThreads < a.bins) {
      const uint32_t s = state[tid + kThreads];
      if ((s & 31u) == 0 && s != kNever) { rv1 = atomicAdd(&a.cursor[(tid + kThreads) * a.cstride], (uint32_t)K); rf1 = true; }
    }
    return go;
  };
  int32_t pk0 = 0, pk1 = 0;   // carried rows
  int64_t pv0 = 0, pv1 = 0;
  uint32_t np = 0;
  issue(lo);
  for (int64_t r0 = lo; r0 < hi; r0 += (int64_t)R * kThreads) {
#pragma unroll
    for (int i = 0; i < R; ++i) { kc[i] = kn[i]; vc[i] = vn[i]; }
    if (r0 + (int64_t)R * kThreads < hi) issue(r0 + (int64_t)R * kThreads);
    uint32_t bn[R], rm[R], sl[R];
#pragma unroll
    for (int i = 0; i < R; ++i) split((uint32_t)(kc[i] - a.kmin), a.wsel, bn[i], rm[i]);
    uint32_t cb0 = 0, cr0 = 0, cb1 = 0, cr1 = 0, cs0 = kSkip, cs1 = kSkip;
    if (np > 0) split((uint32_t)(pk0 - a.kmin), a.wsel, cb0, cr0);
    if (np > 1) split((uint32_t)(pk1 - a.kmin), a.wsel, cb1, cr1);
#pragma unroll
    for (int i = 0; i < R; ++i) sl[i] = r0 + i * kThreads + tid < hi ? atomicAdd(&fill[bn[i]], 1u) : kSkip;
    if (np > 0) cs0 = atomicAdd(&fill[cb0], 1u);
    if (np > 1) cs1 = atomicAdd(&fill[cb1], 1u);
    uint32_t pend = 0, cpend = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (sl[i] < (uint32_t)kCap) place(bn[i], sl[i], rm[i], vc[i]);
      else if (sl[i] != kSkip) pend |= 1u << i;
    }
    if (cs0 < (uint32_t)kCap) place(cb0, cs0, cr0, pv0); else if (cs0 != kSkip) cpend |= 1u;
    if (cs1 < (uint32_t)kCap) place(cb1, cs1, cr1, pv1); else if (cs1 != kSkip) cpend |= 2u;
    if (__builtin_popcount(pend) + __builtin_popcount(cpend) > 2) again[cur] = 1;
    uint32_t go = flush_phase();
    int rounds = 0;
    while (go) {   // (workgroup-uniform) some thread holds more than two rows: rounds until nobody holds any
      uint32_t still = 0, cstill = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if ((pend >> i) & 1u) {
          const uint32_t s = atomicAdd(&fill[bn[i]], 1u);
          if (s < (uint32_t)kCap) place(bn[i], s, rm[i], vc[i]); else still |= 1u << i;
        }
      }
      if (cpend & 1u) { const uint32_t s = atomicAdd(&fill[cb0], 1u); if (s < (uint32_t)kCap) place(cb0, s, cr0, pv0); else cstill |= 1u; }
      if (cpend & 2u) { const uint32_t s = atomicAdd(&fill[cb1], 1u); if (s < (uint32_t)kCap) place(cb1, s, cr1, pv1); else cstill |= 2u; }
      pend = still;
      cpend = cstill;
      if (pend | cpend) again[cur] = 1;
      go = flush_phase();
      if (++rounds > 4096) { if (tid == 0) atomicOr(&a.flags[1], 1u); return; }
    }
    // what is still pending rides along with the next batch
    int32_t nk0 = 0, nk1 = 0;
    int64_t nv0 = 0, nv1 = 0;
    uint32_t c = 0;
    auto push = [&](int32_t k, int64_t v) {
      if (c == 0) { nk0 = k; nv0 = v; } else { nk1 = k; nv1 = v; }
      ++c;
    };
    if (cpend & 1u) push(pk0, pv0);
    if (cpend & 2u) push(pk1, pv1);
#pragma unroll
    for (int i = 0; i < R; ++i) if ((pend >> i) & 1u) push(kc[i], vc[i]);
    pk0 = nk0; pv0 = nv0; pk1 = nk1; pv1 = nv1; np = c;
  }
  // drain the carried rows
  for (int rounds = 0;; ++rounds) {
    uint32_t c = 0;
    int32_t nk0 = 0, nk1 = 0;
    int64_t nv0 = 0, nv1 = 0;
    auto retry = [&](int32_t k, int64_t v) {
      uint32_t b, r;
      split((uint32_t)(k - a.kmin), a.wsel, b, r);
      const uint32_t s = atomicAdd(&fill[b], 1u);
      if (s < (uint32_t)kCap) { place(b, s, r, v); return; }
      if (c == 0) { nk0 = k; nv0 = v; } else { nk1 = k; nv1 = v; }
      ++c;
    };
    if (np > 0) retry(pk0, pv0);
    if (np > 1) retry(pk1, pv1);
    pk0 = nk0; pv0 = nv0; pk1 = nk1; pv1 = nv1; np = c;
    if (np != 0) again[cur] = 1;
    if (!flush_phase()) break;
    if (rounds > 4096) { if (tid == 0) atomicOr(&a.flags[1], 1u); return; }
  }
  if (rf0) state[tid] = (rv0 << 5) | (uint32_t)K;
  if (rf1) state[tid + kThreads] = (rv1 << 5) | (uint32_t)K;
  __syncthreads();
  // the workgroup's partial lines (with their counts), then empty lines up to the end of its last chunk of every bin
  const int sub = tid & 7;
  for (int b = tid >> 3; b < a.bins; b += kThreads / 8) {
    const uint32_t f = fill[b];
    if (f != 0) {
      uint32_t line = 0;
      if (sub == 0) line = next_line(b);
      line = __shfl(line, lane & ~7, 64);
      store_piece(b, line, sub, piece(b, sub, f < 6 ? f : 6, f < 6 ? 0 : f - 6));
    }
    uint32_t s = 0;
    if (sub == 0) s = state[b];
    s = __shfl(s, lane & ~7, 64);
    if (s != kNever) {
      const uint32_t first = s >> 5, left = s & 31u;
      for (uint32_t l = 0; l < left; ++l) {
        u32x4 z = {0, 0, 0, 0};
        store_piece(b, first + l, sub, z);
      }
    }
  }
}

// ---- the memory side alone: read 12 B/row linearly, write whole lines to pseudo-random bins (chunks of K lines)
template <int K>
__global__ __launch_bounds__(256) void linewrite(const int32_t* __restrict__ keys, const int64_t* __restrict__ vals, int64_t n,
                                                 int bins, uint32_t room_lines, uint32_t* cursor, uint8_t* lines, int nt, int cstride) {
  const int lane = threadIdx.x & 63, sub = lane & 7;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // a wave step = 8 lines x K = 96 K rows read (12 rows per line): lane reads rows, writes 16 B pieces
  const int64_t steps = n / (96 * K);
  for (int64_t s = wave; s < steps; s += nwaves) {
    const int64_t row0 = s * 96 * K;
    uint64_t acc = 0;
    for (int i = lane; i < 96 * K; i += 64) acc += (uint64_t)vals[row0 + i] + (uint32_t)keys[row0 + i];
    const uint32_t bin = (uint32_t)((mix((uint64_t)s * 8 + (lane >> 3)) >> 33) % (uint32_t)bins);
    uint32_t base = 0;
    if (sub == 0) base = atomicAdd(&cursor[bin * cstride], (uint32_t)K);
    base = __shfl(base, lane & ~7, 64);
    for (int k = 0; k < K; ++k) {
      const uint32_t line = (base + k) % room_lines;
      u32x4 d = {(uint32_t)acc, (uint32_t)(acc >> 32), (uint32_t)k, bin};
      u32x4* dst = reinterpret_cast<u32x4*>(lines + ((size_t)bin * room_lines + line) * 128 + sub * 16);
      if (nt) __builtin_nontemporal_store(d, dst); else *dst = d;
    }
  }
}

// ---- aggregate: unit = lines [u * unit_lines, ...) of ONE bin; direct-indexed LDS table of `width` groups
struct AggArgs {
  const uint8_t* lines; const uint32_t* cursor; int cstride; uint32_t room_lines; uint32_t unit_lines; int units_per_bin;
  int width; int32_t kmin; unsigned long long* gsum; unsigned long long* gcnt; int64_t range; int dpp;
};

template <int X>
__global__ __launch_bounds__(kThreads) void lines_aggregate(AggArgs a) {
  __shared__ unsigned long long sums[12288];
  __shared__ uint32_t cnts[12288];
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t bin = blockIdx.x / a.units_per_bin, u = blockIdx.x % a.units_per_bin;
  const uint32_t nl = a.cursor[bin * a.cstride];
  const uint32_t l0 = u * a.unit_lines;
  if (l0 >= nl) return;
  const uint32_t l1 = l0 + a.unit_lines < nl ? l0 + a.unit_lines : nl;
  for (int i = tid; i < a.width; i += kThreads) { sums[i] = 0; cnts[i] = 0; }
  __syncthreads();
  const u32x4* src = reinterpret_cast<const u32x4*>(a.lines + ((size_t)bin * a.room_lines + l0) * 128);
  const int64_t npieces = (int64_t)(l1 - l0) * 8;
  const int q = tid & 3;
  auto consume = [&](u32x4 d, bool ok) {
    const int src_lane = lane | 3;
    const uint32_t r0 = __shfl(d[0], src_lane, 64), r1 = __shfl(d[1], src_lane, 64), r2 = __shfl(d[2], src_lane, 64);
    const uint32_t cnt = __shfl(d[3], src_lane, 64);
    if (!ok || q == 3) return;
    const uint32_t rr = q == 0 ? r0 : q == 1 ? r1 : r2;
    if ((uint32_t)(2 * q) < cnt) {
      const uint32_t rem = rr & 0xFFFFu;
      atomicAdd(&sums[rem], ((unsigned long long)d[1] << 32) | d[0]);
      atomicAdd(&cnts[rem], 1u);
    }
    if ((uint32_t)(2 * q + 1) < cnt) {
      const uint32_t rem = rr >> 16;
      atomicAdd(&sums[rem], ((unsigned long long)d[3] << 32) | d[2]);
      atomicAdd(&cnts[rem], 1u);
    }
  };
  u32x4 nxt[X], curd[X];
  auto issue = [&](int64_t p0) {
#pragma unroll
    for (int x = 0; x < X; ++x) {
      const int64_t p = p0 + (int64_t)x * kThreads + tid;
      nxt[x] = __builtin_nontemporal_load(src + (p < npieces ? p : npieces - 8 + (tid & 7)));
    }
  };
  issue(0);
  for (int64_t p0 = 0; p0 < npieces; p0 += (int64_t)X * kThreads) {
#pragma unroll
    for (int x = 0; x < X; ++x) curd[x] = nxt[x];
    if (p0 + (int64_t)X * kThreads < npieces) issue(p0 + (int64_t)X * kThreads);
#pragma unroll
    for (int x = 0; x < X; ++x) consume(curd[x], p0 + (int64_t)x * kThreads + tid < npieces);
  }
  __syncthreads();
  const int64_t key0 = (int64_t)bin * a.width;
  for (int i = tid; i < a.width; i += kThreads) {
    const uint32_t c = cnts[i];
    if (c != 0 && key0 + i < a.range) {
      if (a.units_per_bin == 1) { a.gsum[key0 + i] = sums[i]; a.gcnt[key0 + i] = c; }
      else { atomicAdd(&a.gsum[key0 + i], sums[i]); atomicAdd(&a.gcnt[key0 + i], (unsigned long long)c); }
    }
  }
}

__global__ void ref_sum(const int64_t* v, int64_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (unsigned long long)v[i];
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
__global__ void tab_sum(const unsigned long long* s, const unsigned long long* c, int64_t g, unsigned long long* out) {
  unsigned long long a = 0, b = 0, d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g; i += (int64_t)gridDim.x * blockDim.x) { a += s[i]; b += c[i]; d += c[i] != 0; }
  for (int k = 32; k >= 1; k >>= 1) { a += __shfl_xor(a, k, 64); b += __shfl_xor(b, k, 64); d += __shfl_xor(d, k, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(out + 1, a); atomicAdd(out + 2, b); atomicAdd(out + 3, d); }
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int64_t n = argc > 4 ? atoll(argv[4]) : (int64_t)1 << lg;
  const uint32_t range = argc > 2 ? (uint32_t)atoll(argv[2]) : 10000000u;
  const int width = argc > 3 ? atoi(argv[3]) : 12288;
  const int wsel = width == 8192 ? 0 : 1;
  const int bins = (int)((range + width - 1) / width);
  if (bins > kMaxBins) { fprintf(stderr, "%d bins > %d\n", bins, kMaxBins); return 1; }
  int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  int32_t* keys; int64_t* vals; uint8_t* lines; uint32_t *cursor, *flags; unsigned long long *gsum, *gcnt, *chk;
  const int64_t mean_lines = n / bins / kCap;
  const uint32_t room_lines = (uint32_t)(mean_lines + mean_lines / 16 + 4096 + (int64_t)ncu * 9);
  CK(hipMalloc(&keys, n * 4)); CK(hipMalloc(&vals, n * 8)); CK(hipMalloc(&lines, (size_t)bins * room_lines * 128));
  CK(hipMalloc(&cursor, kMaxBins * 4 * kCStrideMax)); CK(hipMalloc(&flags, 8)); CK(hipMalloc(&gsum, (size_t)range * 8)); CK(hipMalloc(&gcnt, (size_t)range * 8));
  CK(hipMalloc(&chk, 32));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, keys, vals, n, range);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("rows=%lld keys in [0,%u) width=%d bins=%d room=%u lines (%.2f GB) CUs=%d\n", (long long)n, range, width, bins, room_lines,
         (double)bins * room_lines * 128 / 1e9, ncu);
  auto timeit = [&](const char* name, double bytes, auto&& launch, int reps = 4) {
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(cursor, 0, kMaxBins * 4 * kCStrideMax)); CK(hipMemsetAsync(flags, 0, 8));
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) best = ms < best ? ms : best;
    }
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
    return best;
  };
  const double wbytes = (double)n / kCap * 128;
  for (int cs : {1, 32}) {
    char nm[96];
    snprintf(nm, sizeof nm, "linewrite K=1 cursor stride %d (ceiling)", cs);
    timeit(nm, n * 12.0 + wbytes, [&] { hipLaunchKernelGGL(linewrite<1>, dim3(8192), dim3(256), 0, 0, keys, vals, n, bins, room_lines, cursor, lines, 0, cs); }, 3);
    snprintf(nm, sizeof nm, "linewrite K=4 cursor stride %d (ceiling)", cs);
    timeit(nm, n * 12.0 + wbytes, [&] { hipLaunchKernelGGL(linewrite<4>, dim3(8192), dim3(256), 0, 0, keys, vals, n, bins, room_lines, cursor, lines, 0, cs); }, 3);
    snprintf(nm, sizeof nm, "linewrite K=16 cursor stride %d (ceiling)", cs);
    timeit(nm, n * 12.0 + wbytes, [&] { hipLaunchKernelGGL(linewrite<16>, dim3(8192), dim3(256), 0, 0, keys, vals, n, bins, room_lines, cursor, lines, 0, cs); }, 3);
  }
  timeit("linewrite K=16 stride 32 nt stores", n * 12.0 + wbytes, [&] { hipLaunchKernelGGL(linewrite<16>, dim3(8192), dim3(256), 0, 0, keys, vals, n, bins, room_lines, cursor, lines, 1, 32); }, 3);
  ScatterArgs sa{keys, vals, n, 0, 0, wsel, bins, room_lines, cursor, lines, flags, 0, 32};
  auto run_scatter = [&](auto kern, int R, int nt, int wgs, int cs) {
    sa.nt = nt;
    sa.cstride = cs;
    const int64_t per = (n + wgs - 1) / wgs;
    sa.rows_per_wg = (per + (int64_t)R * kThreads - 1) / ((int64_t)R * kThreads) * ((int64_t)R * kThreads);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), 0, 0, sa);
  };
  timeit("wc_scatter R=4 K=4 packed cursors", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter<4, 4>, 4, 1, ncu, 1); }, 3);
  timeit("wc_scatter R=4 K=4 cursor per line", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter<4, 4>, 4, 1, ncu, 32); }, 3);
  timeit("wc_scatter R=4 K=16 cursor per line", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter<4, 16>, 4, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=4 K=4", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<4, 4>, 4, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=4", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 4>, 8, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=2 K=4", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<2, 4>, 2, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=3 K=4", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<3, 4>, 3, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=6 K=4", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<6, 4>, 6, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=4 K=8", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<4, 8>, 4, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=8", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 8>, 8, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=16", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 16>, 8, 1, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=4 plain ld/st", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 4>, 8, 0, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=4 nt ld+st", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 4>, 8, 3, ncu, 32); }, 3);
  timeit("wc_scatter2 R=8 K=4 packed cursors", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<8, 4>, 8, 1, ncu, 1); }, 3);
  // the run the aggregate reads
  const float ts = timeit("wc_scatter2 R=4 K=4 (kept)", n * 12.0 + wbytes, [&] { run_scatter(wc_scatter2<4, 4>, 4, 1, ncu, 32); }, 2);
  uint32_t hflags[2]; CK(hipMemcpy(hflags, flags, 8, hipMemcpyDeviceToHost));
  std::vector<uint32_t> hcur((size_t)bins * 32); CK(hipMemcpy(hcur.data(), cursor, (size_t)bins * 32 * 4, hipMemcpyDeviceToHost));
  uint64_t tot_lines = 0; uint32_t mx = 0; for (int b = 0; b < bins; ++b) { const uint32_t c = hcur[(size_t)b * 32]; tot_lines += c; mx = c > mx ? c : mx; }
  printf("flags: overflow=%u skew=%u; lines written %llu (%.3f GB, %.2f B/row), fullest room %u / %u\n", hflags[0], hflags[1],
         (unsigned long long)tot_lines, tot_lines * 128.0 / 1e9, tot_lines * 128.0 / n, mx, room_lines);
  const uint32_t unit_lines = (uint32_t)(((int64_t)1 << 21) / kCap);
  const int upb = (int)((mx + unit_lines - 1) / unit_lines);
  AggArgs aa{lines, cursor, 32, room_lines, unit_lines, upb, width, 0, gsum, gcnt, (int64_t)range, 0};
  float ta = 1e9f;
  for (int x : {4, 8}) {
    for (int r = 0; r < 2; ++r) {
      CK(hipMemsetAsync(gsum, 0, (size_t)range * 8)); CK(hipMemsetAsync(gcnt, 0, (size_t)range * 8));
      CK(hipEventRecord(e0));
      if (x == 4) hipLaunchKernelGGL(lines_aggregate<4>, dim3(bins * upb), dim3(kThreads), 0, 0, aa);
      else if (x == 8) hipLaunchKernelGGL(lines_aggregate<8>, dim3(bins * upb), dim3(kThreads), 0, 0, aa);
      else hipLaunchKernelGGL(lines_aggregate<12>, dim3(bins * upb), dim3(kThreads), 0, 0, aa);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { printf("lines_aggregate X=%-2d units/bin=%d               %8.3f ms  %7.1f GB/s\n", x, upb, ms, tot_lines * 128.0 / ms / 1e6); ta = ms < ta ? ms : ta; }
    }
  }
  CK(hipMemset(chk, 0, 32));
  hipLaunchKernelGGL(ref_sum, dim3(2048), dim3(256), 0, 0, vals, n, chk);
  hipLaunchKernelGGL(tab_sum, dim3(2048), dim3(256), 0, 0, gsum, gcnt, (int64_t)range, chk);
  unsigned long long h[4]; CK(hipMemcpy(h, chk, 32, hipMemcpyDeviceToHost));
  printf("check: sum of values %016llx, sum of group sums %016llx, rows counted %llu / %lld, groups %llu  => %s\n", h[0], h[1], h[2],
         (long long)n, h[3], (h[0] == h[1] && h[2] == (unsigned long long)n) ? "OK" : "MISMATCH");
  printf("scatter + aggregate = %.3f ms for %lld rows (%.1f Grows/s)\n", ts + ta, (long long)n, n / (ts + ta) / 1e6);
  return 0;
}
