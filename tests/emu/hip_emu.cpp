// hip_emu.cpp - fiber runtime of the CPU kernel emulator (TEST INFRASTRUCTURE ONLY, see hip_emu.h).
#include "hip_emu.h"

#include <sys/mman.h>

namespace emu {
thread_local BlockState* g_blk = nullptr;
thread_local emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
thread_local unsigned char* g_dyn_smem = nullptr;
std::mutex g_atomic_mu;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

static void set_tid(int t) {
  const emu_dim3& bd = g_blockDim;
  g_threadIdx.x = t % bd.x;
  g_threadIdx.y = (t / bd.x) % bd.y;
  g_threadIdx.z = t / (bd.x * bd.y);
}

static void switch_to(int next) {
  BlockState* b = g_blk;
  int prev = b->cur;
  b->cur = next;
  set_tid(next);
  emu_switch(&b->fibers[prev].sp, b->fibers[next].sp);
}

void yield() {
  BlockState* b = g_blk;
  int n = b->nthreads, c = b->cur;
  for (int i = 1; i <= n; ++i) {
    int t = (c + i) % n;
    if (!b->fibers[t].done) {
      if (t != c) switch_to(t);
      return;
    }
  }
}

int lane_id() { return g_blk->cur & 63; }
WaveState& wave() { return g_blk->waves[g_blk->cur >> 6]; }

void block_barrier() {
  BlockState* b = g_blk;
  if (++b->arrived == b->nthreads) { b->arrived = 0; b->gen++; return; }
  unsigned my = b->gen;
  while (b->gen == my) yield();
}

void wave_barrier() {
  BlockState* b = g_blk;
  WaveState& w = b->waves[b->cur >> 6];
  int wsize = std::min(64, b->nthreads - (b->cur >> 6) * 64);
  if (++w.arrived == wsize) { w.arrived = 0; w.gen++; return; }
  unsigned my = w.gen;
  while (w.gen == my) yield();
}

extern "C" void emu_fiber_main() {
  BlockState* b = g_blk;
  (*b->body)();
  int me = b->cur;
  b->fibers[me].done = true;
  b->live--;
  if (b->live == 0) {
    void* dummy;
    emu_switch(&dummy, b->main_sp);
  }
  yield();
  abort();  // unreachable: a finished fiber is never resumed
}

static const size_t kStack = 256 * 1024;

static void run_block(BlockState& b, int nthreads, const std::function<void()>& body) {
  b.nthreads = nthreads; b.cur = 0; b.arrived = 0; b.gen = 0; b.live = nthreads; b.body = &body;
  for (auto& w : b.waves) { w.arrived = 0; w.gen = 0; }
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = b.fibers[t];
    f.done = false;
    // initial frame: 6 callee-saved regs (zero) + return address -> emu_fiber_main.
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    uint64_t* sp = (uint64_t*)(top - 8);      // after `ret`, rsp = top  -> entry rsp%16==0 ... fix below
    // System V: at function entry rsp % 16 == 8.  `ret` pops the address, so place it at top-16.
    sp = (uint64_t*)(top - 16);
    *sp = (uint64_t)(uintptr_t)&emu_fiber_main;
    sp -= 6;
    for (int i = 0; i < 6; ++i) sp[i] = 0;
    f.sp = sp;
  }
  g_blk = &b;
  set_tid(0);
  emu_switch(&b.main_sp, b.fibers[0].sp);
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  int nthreads = (int)(block.x * block.y * block.z);
  long nblocks = (long)grid.x * grid.y * grid.z;
  int nworkers = (int)std::min<long>(nblocks, std::max(1u, std::thread::hardware_concurrency()));
  std::atomic<long> next{0};
  auto worker = [&]() {
    BlockState b;
    b.fibers.resize(nthreads);
    b.waves.resize((nthreads + 63) / 64);
    char* stacks = (char*)mmap(nullptr, kStack * nthreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char*)MAP_FAILED) { perror("emu mmap"); abort(); }
    for (int t = 0; t < nthreads; ++t) b.fibers[t].stack = stacks + kStack * t;
    std::vector<unsigned char> dyn(smem + 64);
    g_dyn_smem = (unsigned char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
    g_blockDim = block; g_gridDim = grid;
    for (;;) {
      long i = next.fetch_add(1);
      if (i >= nblocks) break;
      g_blockIdx.x = (unsigned)(i % grid.x);
      g_blockIdx.y = (unsigned)((i / grid.x) % grid.y);
      g_blockIdx.z = (unsigned)(i / ((long)grid.x * grid.y));
      run_block(b, nthreads, body);
    }
    munmap(stacks, kStack * nthreads);
    g_blk = nullptr;
  };
  if (nworkers <= 1) { std::thread t(worker); t.join(); return; }
  std::vector<std::thread> ts;
  for (int i = 0; i < nworkers; ++i) ts.emplace_back(worker);
  for (auto& t : ts) t.join();
}
}  // namespace emu

// test hooks (tests/test_emu_ops.py): the emulator's e4m3 conversion, to be pinned against an independent implementation (torch)
// and against the hardware probe's outputs (profiles/r02_f8_semantics_probe.txt)
extern "C" int sdm_emu_f32_to_e4m3(float x) { return (int)emu_f32_to_e4m3(x); }
extern "C" float sdm_emu_e4m3_to_f32(int v) { return emu_e4m3_to_f32((unsigned char)v); }
