// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <stdint.h>
#include <stdio.h>
#include <chrono>
#include <map>
#include <string>
#include <algorithm>
#include <vector>

namespace hipemu {
Idx threadIdx_, blockIdx_;
dim3 blockDim_, gridDim_;
void* dyn_shared = nullptr;
int shfl_buf[1024];

namespace {
#if !defined(__x86_64__)
#error "tests/hipemu switches fibers with a dozen lines of x86-64 assembly (below); port fiber_switch for this host"
#endif
// Fiber switch: push the System V callee-saved registers, swap stack pointers, pop, return.  (glibc's swapcontext also
// saves the signal mask -- one rt_sigprocmask system call per switch, which was a fifth of the interpreter's run time.)
extern "C" void hipemu_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipemu_fiber_switch
.type hipemu_fiber_switch,@function
hipemu_fiber_switch:
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
.size hipemu_fiber_switch,.-hipemu_fiber_switch
)");

constexpr size_t STACK = 1u << 20;
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  Idx tid;
};
std::vector<Fiber> pool;
void* sched_sp = nullptr;
Fiber* cur = nullptr;
const std::function<void()>* cur_body = nullptr;

void tramp() {
  (*cur_body)();
  cur->done = true;
  hipemu_fiber_switch(&cur->sp, sched_sp);
  abort();  // a finished fiber is never resumed
}

// a fresh fiber: six zeroed callee-saved registers, then tramp as the address fiber_switch returns to; rsp is 8 mod 16
// on entry to tramp, as after a call
void fiber_reset(Fiber& f) {
  uintptr_t* top = (uintptr_t*)(f.stack + STACK);
  top[-1] = 0;
  top[-2] = (uintptr_t)&tramp;
  for (int i = 3; i <= 8; i++) top[-i] = 0;
  f.sp = top - 8;
}
}  // namespace

void sync_threads() { hipemu_fiber_switch(&cur->sp, sched_sp); }

namespace {
// OG_EMU_PROF=1: seconds per kernel name, printed at exit (where does the interpreter spend a test's time?)
struct Prof {
  std::map<std::string, std::pair<double, size_t>> t;
  bool on = getenv("OG_EMU_PROF") && atoi(getenv("OG_EMU_PROF"));
  ~Prof() {
    if (!on) return;
    std::vector<std::pair<double, std::string>> v;
    for (auto& kv : t) v.push_back({kv.second.first, kv.first + " x" + std::to_string(kv.second.second)});
    std::sort(v.rbegin(), v.rend());
    for (size_t i = 0; i < v.size() && i < 25; i++) fprintf(stderr, "[hipemu] %8.2f s  %s\n", v[i].first, v[i].second.c_str());
  }
} prof;
}  // namespace

static void launch_impl(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body, const char* name) {
  if (!prof.on) return launch_impl(grid, block, shmem, body);
  const auto t0 = std::chrono::steady_clock::now();
  launch_impl(grid, block, shmem, body);
  auto& e = prof.t[name ? name : "?"];
  e.first += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  e.second++;
}

static void launch_impl(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  const size_t nt = (size_t)block.x * block.y * block.z;
  if (nt == 0 || (size_t)grid.x * grid.y * grid.z == 0) return;
  while (pool.size() < nt) {
    Fiber f;
    f.stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f.stack == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    pool.push_back(f);
  }
  void* sh = nullptr;
  if (shmem) sh = aligned_alloc(256, (shmem + 255) / 256 * 256);
  dyn_shared = sh;
  blockDim_ = block;
  gridDim_ = grid;
  cur_body = &body;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx_ = {bx, by, bz};
        size_t t = 0;
        for (unsigned tz = 0; tz < block.z; tz++)
          for (unsigned ty = 0; ty < block.y; ty++)
            for (unsigned tx = 0; tx < block.x; tx++, t++) {
              Fiber& f = pool[t];
              f.done = false;
              f.tid = {tx, ty, tz};
              fiber_reset(f);
            }
        size_t remaining = nt;
        while (remaining) {
          for (size_t i = 0; i < nt; i++) {
            Fiber& f = pool[i];
            if (f.done) continue;
            cur = &f;
            threadIdx_ = f.tid;
            hipemu_fiber_switch(&sched_sp, f.sp);
            if (f.done) remaining--;
          }
        }
      }
  cur_body = nullptr;
  dyn_shared = nullptr;
  free(sh);
}
}  // namespace hipemu
