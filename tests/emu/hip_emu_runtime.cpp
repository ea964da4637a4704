// Scheduler of the CPU SIMT emulator (see hip_emu.h).  TEST INFRASTRUCTURE ONLY.
#include <sys/mman.h>

#include "hip_emu.h"

namespace hipemu {

static State g_state;
State& st() { return g_state; }

static constexpr size_t kStack = 256 * 1024;
static std::vector<void*> g_stacks;

static void* stack_for(size_t i) {
  while (g_stacks.size() <= i) {
    void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); abort(); }
    g_stacks.push_back(p);
  }
  return g_stacks[i];
}

void yield() {
  State& s = st();
  swapcontext(&s.cur->ctx, &s.sched);
}

static void release_if_complete(Wave& w) {
  if (w.nactive > 0 && w.arrived >= w.nactive) { w.arrived = 0; ++w.gen; }
}

static void trampoline() {
  State& s = st();
  Fiber* f = s.cur;
  (*s.body)();
  f->done = true;
  Wave& w = s.waves[f->tid >> 6];
  --w.nactive;
  release_if_complete(w);
  --s.block_nactive;
  if (s.block_nactive > 0 && s.block_arrived >= s.block_nactive) { s.block_arrived = 0; ++s.block_gen; }
  swapcontext(&f->ctx, &s.sched);
}

static void run_block(const std::function<void()>& body) {
  State& s = st();
  const int nthreads = static_cast<int>(s.block.x * s.block.y * s.block.z);
  s.fibers.assign(nthreads, Fiber{});
  s.waves.assign((nthreads + 63) / 64, Wave{});
  s.block_nactive = nthreads;
  s.block_arrived = 0;
  s.body = &body;
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = s.fibers[t];
    f.tid = t;
    f.tidx = dim3(t % s.block.x, (t / s.block.x) % s.block.y, t / (s.block.x * s.block.y));
    f.done = false;
    s.waves[t >> 6].nactive++;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = stack_for(t);
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &s.sched;
    makecontext(&f.ctx, trampoline, 0);
  }
  int remaining = nthreads;
  long spins = 0;
  while (remaining > 0) {
    remaining = 0;
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = s.fibers[t];
      if (f.done) continue;
      s.cur = &f;
      swapcontext(&s.sched, &f.ctx);
      if (!f.done) ++remaining;
    }
    if (++spins > 50000000L) { fprintf(stderr, "hipemu: deadlock in workgroup (%u,%u)\n", s.bidx.x, s.bidx.y); abort(); }
  }
  s.cur = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = st();
  s.grid = grid;
  s.block = block;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.bidx = dim3(x, y, z);
        run_block(body);
      }
}

}  // namespace hipemu
