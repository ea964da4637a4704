// Scheduler of the CPU SIMT emulator (see hip_emu.h).  TEST INFRASTRUCTURE ONLY.
#include <sys/mman.h>

#include "hip_emu.h"

#if !defined(__x86_64__)
#error "the emulator's context switch is written for x86-64 (System V ABI)"
#endif

// hipemu_switch(&save_sp, load_sp): push the callee-saved registers, store this stack pointer, adopt the other
// one, pop its registers and return into it.  (MXCSR / x87 control words are never changed by the kernels.)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl hipemu_switch
    .type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

static State g_state;
State& st() { return g_state; }

static constexpr size_t kStack = 256 * 1024;
static std::vector<void*> g_stacks;

// Every fiber stack sits above one PROT_NONE guard page: a kernel whose per-thread locals (or -O1 recursion)
// outgrow kStack faults there instead of silently scribbling over the neighbouring fiber's stack.
static constexpr size_t kGuard = 4096;
static void* stack_for(size_t i) {
  while (g_stacks.size() <= i) {
    void* p = mmap(nullptr, kStack + kGuard, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); abort(); }
    if (mprotect(p, kGuard, PROT_NONE) != 0) { perror("mprotect"); abort(); }
    g_stacks.push_back(static_cast<char*>(p) + kGuard);
  }
  return g_stacks[i];
}

void yield() {
  State& s = st();
  hipemu_switch(&s.cur->sp, s.sched_sp);
}

// HIPEMU_SCHEDULE=reverse: workgroups run from the last to the first and the fibers of a workgroup are resumed
// in descending thread order.  The hardware promises no order between workgroups (nor between the waves of one),
// so every result must be identical under both schedules; a kernel that silently relies on "block 0 ran first"
// or on ascending lane order between barriers shows up as a difference.
static bool g_reverse = false;

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
  hipemu_switch(&f->sp, s.sched_sp);   // never resumed
  abort();
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
    // first switch into the fiber: six zeroed register slots, then `ret` into trampoline with the stack
    // pointer where a call would have left it (8 below a 16-byte boundary)
    void** top = reinterpret_cast<void**>(static_cast<char*>(stack_for(t)) + kStack);
    top[-1] = nullptr;                                   // trampoline's (unused) return address
    top[-2] = reinterpret_cast<void*>(&trampoline);
    for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
    f.sp = top - 8;
  }
  int remaining = nthreads;
  long spins = 0;
  while (remaining > 0) {
    remaining = 0;
    for (int k = 0; k < nthreads; ++k) {
      Fiber& f = s.fibers[g_reverse ? nthreads - 1 - k : k];
      if (f.done) continue;
      s.cur = &f;
      hipemu_switch(&s.sched_sp, f.sp);
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
  const char* sched = getenv("HIPEMU_SCHEDULE");
  g_reverse = sched != nullptr && strcmp(sched, "reverse") == 0;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.bidx = g_reverse ? dim3(grid.x - 1 - x, grid.y - 1 - y, grid.z - 1 - z) : dim3(x, y, z);
        run_block(body);
      }
}

}  // namespace hipemu

// Self-test hook for tests/test_emu_schedule.py: the order in which the (block, thread) pairs of a launch first run.
extern "C" int hipemu_selftest_order(int* order, int nblocks, int nthreads) {
  int n = 0;
  hipemu::launch(dim3(nblocks), dim3(nthreads), [&]() {
    order[n++] = static_cast<int>(blockIdx.x) * nthreads + static_cast<int>(threadIdx.x);
    __syncthreads();
  });
  return n;
}
