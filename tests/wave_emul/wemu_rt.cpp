// Test infrastructure -- NOT part of the product.  Scheduler of the wavefront emulation (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>

#include <vector>

namespace wemu {

Idx g_tid, g_bid;
dim3 g_bdim, g_gdim;
int g_lane;

namespace {

enum State { READY, PARKED, DONE };
struct Fibre {
  void* sp;      // saved stack pointer while the fibre is not running
  char* stack;
  int state;
  Op* op;
  Idx tid;
  int lane;
};

constexpr size_t kStack = 256 << 10;  // per work-item (virtual; pages are touched on demand)
constexpr int kMaxThreads = 1024;
constexpr size_t kDynLds = 160 << 10;

char* g_stacks = nullptr;
char* g_dyn_lds = nullptr;
Fibre g_fibres[kMaxThreads];
Fibre* g_cur = nullptr;
void* g_sched_sp = nullptr;
Body g_body;
void* g_closure;
long g_counters[8];
size_t g_dyn_bytes = 0;

// callee-saved registers of the SysV x86-64 ABI + the stack pointer; nothing else survives a call anyway
extern "C" void wemu_switch(void** save_sp, void* new_sp);
asm(R"(
.text
.globl wemu_switch
.type wemu_switch,@function
wemu_switch:
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
.size wemu_switch,.-wemu_switch
)");

void fibre_main() {
  g_body(g_closure);
  g_cur->state = DONE;
  wemu_switch(&g_cur->sp, g_sched_sp);
  abort();  // a finished fibre is never resumed
}

void fibre_init(Fibre& f, int linear, const dim3& b) {
  f.state = READY;
  f.op = nullptr;
  f.tid = {linear % b.x, (linear / b.x) % b.y, linear / (b.x * b.y)};
  f.lane = linear & 63;
  // top of stack: [fake return address of fibre_main][fibre_main as wemu_switch's return target][six registers]
  uint64_t* top = reinterpret_cast<uint64_t*>(f.stack + kStack);
  top[-1] = 0;
  top[-2] = reinterpret_cast<uint64_t>(&fibre_main);
  for (int i = 3; i <= 8; i++) top[-i] = 0;
  f.sp = top - 8;
}

void run(Fibre& f) {
  g_cur = &f;
  g_tid = f.tid;
  g_lane = f.lane;
  f.state = READY;
  wemu_switch(&g_sched_sp, f.sp);
}

inline float h2f(_Float16 h) { return (float) h; }

// carries out the operation of the lanes `mask` of one wave (all parked at one site)
void execute(Fibre* wave, uint64_t mask, int n_lanes) {
  int first = __builtin_ctzll(mask);
  const int kind = wave[first].op->kind;
  switch (kind) {
    case PERMUTE: {
      uint32_t vals[64];
      for (int l = 0; l < n_lanes; l++)
        if ((mask >> l) & 1) vals[l] = wave[l].op->val;
      for (int l = 0; l < n_lanes; l++)
        if ((mask >> l) & 1) {
          Op& o = *wave[l].op;
          o.result = (o.src >= 0 && o.src < n_lanes && ((mask >> o.src) & 1)) ? vals[o.src] : o.fallback;
        }
      break;
    }
    case BALLOT: {
      uint64_t r = 0;
      for (int l = 0; l < n_lanes; l++)
        if (((mask >> l) & 1) && wave[l].op->val) r |= 1ull << l;
      for (int l = 0; l < n_lanes; l++)
        if ((mask >> l) & 1) wave[l].op->result = r;
      break;
    }
    case FIRST: {
      const uint32_t v = wave[first].op->val;
      for (int l = 0; l < n_lanes; l++)
        if ((mask >> l) & 1) wave[l].op->result = v;
      break;
    }
    case MFMA_16x16x16_F16:
    case MFMA_16x16x32_F16: {
      // matrix instructions ignore EXEC for their sources on the hardware; the kernels issue them with full waves
      if (mask != ~0ull) {
        fprintf(stderr, "wave_emul: MFMA issued with EXEC mask %016llx\n", (unsigned long long) mask);
        abort();
      }
      const int kq = kind == MFMA_16x16x16_F16 ? 4 : 8;  // k values per lane
      float A[16][32], B[32][16];
      for (int l = 0; l < 64; l++) {
        const _Float16* a = static_cast<const _Float16*>(wave[l].op->a);
        const _Float16* b = static_cast<const _Float16*>(wave[l].op->b);
        for (int k = 0; k < kq; k++) {
          A[l & 15][kq * (l >> 4) + k] = h2f(a[k]);
          B[kq * (l >> 4) + k][l & 15] = h2f(b[k]);
        }
      }
      for (int l = 0; l < 64; l++) {
        const float* c = static_cast<const float*>(wave[l].op->c);
        float* d = static_cast<float*>(wave[l].op->d);
        const int j = l & 15;
        for (int r = 0; r < 4; r++) {
          const int i = 4 * (l >> 4) + r;
          float acc = c[r];
          for (int k = 0; k < 4 * kq; k++) acc += A[i][k] * B[k][j];
          d[r] = acc;
        }
      }
      break;
    }
    default:
      abort();
  }
}

void run_block(int n_threads) {
  const int n_waves = (n_threads + 63) / 64;
  for (int t = 0; t < n_threads; t++) fibre_init(g_fibres[t], t, g_bdim);
  memset(g_dyn_lds, 0xCD, g_dyn_bytes);
  for (;;) {
    for (int w = 0; w < n_waves; w++) {
      Fibre* wave = g_fibres + 64 * w;
      const int n_lanes = std::min(64, n_threads - 64 * w);
      for (;;) {
        for (int l = 0; l < n_lanes; l++)
          if (wave[l].state == READY) run(wave[l]);
        // every lane of the wave is parked or done: the lanes parked at the lowest cross-lane site form the next EXEC mask
        const void* site = nullptr;
        bool divergent = false;
        for (int l = 0; l < n_lanes; l++) {
          if (wave[l].state != PARKED || wave[l].op->kind >= BARRIER) continue;
          const void* s = wave[l].op->site;
          if (site == nullptr) site = s;
          else if (s != site) {
            divergent = true;
            if (s < site) site = s;
          }
        }
        if (site == nullptr) break;  // nothing but barriers and finished lanes
        uint64_t mask = 0;
        for (int l = 0; l < n_lanes; l++)
          if (wave[l].state == PARKED && wave[l].op->kind < BARRIER && wave[l].op->site == site) mask |= 1ull << l;
        if (divergent) g_counters[2]++;
        g_counters[1]++;
        execute(wave, mask, n_lanes);
        for (int l = 0; l < n_lanes; l++)
          if ((mask >> l) & 1) wave[l].state = READY;
      }
    }
    // no wave can move: every live work-item sits at the barrier (or the group has finished)
    int at_barrier = 0, any = 0;
    for (int t = 0; t < n_threads; t++)
      if (g_fibres[t].state == PARKED) {
        at_barrier++;
        if (g_fibres[t].op->kind == BARRIER_OR && g_fibres[t].op->val) any = 1;
      }
    if (at_barrier == 0) break;  // (a wave's loop above ends with its lanes finished or at a barrier)
    g_counters[3]++;
    for (int t = 0; t < n_threads; t++)
      if (g_fibres[t].state == PARKED) {
        g_fibres[t].op->result = (uint64_t) any;
        g_fibres[t].state = READY;
      }
  }
}

}  // namespace

uint64_t park(Op& op) {
  Fibre* f = g_cur;
  f->op = &op;
  f->state = PARKED;
  wemu_switch(&f->sp, g_sched_sp);
  return op.result;
}

void* dyn_lds() { return g_dyn_lds; }

void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, Body body, void* closure) {
  const size_t n_threads = (size_t) block.x * block.y * block.z;
  if (n_threads == 0 || n_threads > (size_t) kMaxThreads || dyn_lds_bytes > kDynLds) {
    fprintf(stderr, "wave_emul: launch of %zu work-items per group / %zu bytes of dynamic LDS\n", n_threads, dyn_lds_bytes);
    abort();
  }
  if (g_stacks == nullptr) {
    g_stacks = static_cast<char*>(mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (g_stacks == MAP_FAILED || posix_memalign(reinterpret_cast<void**>(&g_dyn_lds), 256, kDynLds) != 0) abort();
    for (int t = 0; t < kMaxThreads; t++) g_fibres[t].stack = g_stacks + kStack * t;
  }
  g_counters[0]++;
  g_dyn_bytes = std::min(kDynLds, (dyn_lds_bytes + 4095) & ~(size_t) 4095);  // (a page of poison behind what was asked for)
  g_body = body;
  g_closure = closure;
  g_bdim = block;
  g_gdim = grid;
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) {
        g_bid = {x, y, z};
        run_block((int) n_threads);
        g_counters[4] += (long) n_threads;
      }
  g_cur = nullptr;
}

extern "C" long wemu_counter(int which) { return which >= 0 && which < 8 ? g_counters[which] : -1; }

}  // namespace wemu
