// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/hipemu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>
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
constexpr size_t STACK = 1u << 20;
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  Idx tid;
};
std::vector<Fiber> pool;
ucontext_t sched_ctx;
Fiber* cur = nullptr;
const std::function<void()>* cur_body = nullptr;

void tramp() {
  (*cur_body)();
  cur->done = true;
  swapcontext(&cur->ctx, &sched_ctx);
}
}  // namespace

void sync_threads() { swapcontext(&cur->ctx, &sched_ctx); }

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
              getcontext(&f.ctx);
              f.ctx.uc_stack.ss_sp = f.stack;
              f.ctx.uc_stack.ss_size = STACK;
              f.ctx.uc_link = nullptr;
              makecontext(&f.ctx, tramp, 0);
            }
        size_t remaining = nt;
        while (remaining) {
          for (size_t i = 0; i < nt; i++) {
            Fiber& f = pool[i];
            if (f.done) continue;
            cur = &f;
            threadIdx_ = f.tid;
            swapcontext(&sched_ctx, &f.ctx);
            if (f.done) remaining--;
          }
        }
      }
  cur_body = nullptr;
  dyn_shared = nullptr;
  free(sh);
}
}  // namespace hipemu
