// Test infrastructure -- NOT part of the product.  Scheduler of the wavefront emulation (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>

#include <link.h>
#include <pthread.h>

#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace wemu {

Idx g_tid, g_bid;
dim3 g_bdim, g_gdim;
int g_lane;

namespace {

enum State { READY, PARKED, DONE };
// Where a fibre is in the program, beyond an address: the live function activations (frame pointer + the basic block it is in,
// from the compiler's block instrumentation: __sanitizer_cov_trace_pc) and, per activation, the loops it is inside with the number
// of back-edges it has taken.  A loop is learnt the first time any fibre jumps backwards inside an activation: [head, latch] are the
// target and source blocks (at -O0 a loop's blocks are contiguous and its latch is the last of them).
constexpr int kMaxFrames = 32, kMaxLoops = 32, kMaxLevels = 40;
struct FrameRec { uintptr_t fp, block; };
struct LoopRec { uintptr_t fp, head, latch; uint32_t iter; };
struct Fibre {
  void* sp;      // saved stack pointer while the fibre is not running
  char* stack;
  int state;
  Op* op;
  Idx tid;
  int lane;
  int nf, nl;
  FrameRec* top;      // &fr[nf - 1], or &no_frame
  uintptr_t top_hi;   // the latch of the innermost loop of that activation (blocks beyond it leave the loop)
  FrameRec no_frame;
  FrameRec fr[kMaxFrames];
  LoopRec lp[kMaxLoops];
  // taken when it parks: the return addresses of its call chain, kernel side first, with the frame each lies in
  int nlev;  // -1: not read yet
  const uintptr_t* park_fp;
  uintptr_t ra[kMaxLevels], rfp[kMaxLevels];
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
int g_block_threads = 0;
int g_schedule = 0;  // 0: workgroups and waves in ascending order; 1: both descending; 2: waves rotate, workgroups interleave from both ends
// ---- the memory traffic of a launch, as the source text performs it (load / store instrumentation of the "traffic" build) ----
struct LaunchTraffic {
  std::string kernel;
  unsigned long long v[8];
};
std::vector<LaunchTraffic> g_traffic;
std::unordered_set<uintptr_t> g_ld_lines, g_st_lines;
unsigned long long g_tr[6];
uintptr_t g_host_stack_lo = 0, g_host_stack_hi = 0;  // the launching thread's stack: kernel arguments (by-reference captures of the launch)
uintptr_t g_image_lo = 0, g_image_hi = 0;  // this shared object: `__shared__` statics (and constant tables) live in it
bool g_tracing = false;                     // a launch is running and the build is instrumented
// ---- a race detector between streams (traffic build + WEMU happens-before hooks of the host stand-ins; tests/wave_emul/stream_races.py) ----
// The host orders its streams with events.  The stand-ins report every record / wait / host-side synchronisation here (vector clocks,
// one component per stream); every launch carries its stream; every global access of a launch is checked against the last write and the
// last reads of its 4-byte word: an access that is not ordered behind a conflicting access of ANOTHER stream is a race the GPU is
// free to lose.  (Torch operations of the host are not seen: they can hide a race, not invent one.  Freed memory must not be reused
// while this runs -- the caching allocator of the GPU keeps a stream's blocks to that stream, malloc does not: LD_PRELOAD a free() that
// does nothing.)
constexpr int kStreams = 8;
struct VC { int c[kStreams]; };
bool g_hb_on = false;
VC g_vc[kStreams], g_host_vc;
std::vector<VC> g_tokens;
int g_cur_stream = 0, g_cur_kernel = 0;
std::vector<std::string> g_kernel_names;
struct Shadow {
  int w_stream = -1, w_epoch = 0, w_kernel = 0;
  int r_epoch[kStreams] = {0}, r_kernel[kStreams] = {0};
  unsigned streams = 0;  // every stream that touched the word
};
std::unordered_map<uintptr_t, Shadow> g_shadow;
struct Race { std::string first, second, kind; uintptr_t addr; long count; };
std::vector<Race> g_races;
void report_race(int k_first, int k_second, const char* kind, uintptr_t granule) {
  for (auto& r : g_races)
    if (r.first == g_kernel_names[k_first] && r.second == g_kernel_names[k_second] && r.kind == kind) {
      r.count++;
      return;
    }
  g_races.push_back({g_kernel_names[k_first], g_kernel_names[k_second], kind, granule << 2, 1});
}
inline void hb_access(uintptr_t a, bool store) {
  const int s = g_cur_stream;
  const VC& me = g_vc[s];
  Shadow& sh = g_shadow[a >> 2];
  sh.streams |= 1u << s;
  if (sh.w_stream >= 0 && sh.w_stream != s && sh.w_epoch > me.c[sh.w_stream]) report_race(sh.w_kernel, g_cur_kernel, store ? "write after write" : "read after write", a >> 2);
  if (store) {
    for (int r = 0; r < kStreams; r++)
      if (r != s && sh.r_epoch[r] > me.c[r]) report_race(sh.r_kernel[r], g_cur_kernel, "write after read", a >> 2);
    sh.w_stream = s;
    sh.w_epoch = me.c[s];
    sh.w_kernel = g_cur_kernel;
  } else {
    sh.r_epoch[s] = me.c[s];
    sh.r_kernel[s] = g_cur_kernel;
  }
}
std::unordered_map<uintptr_t, uintptr_t> g_loops;  // head block -> latch block of every loop seen so far

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
  f.nf = f.nl = f.nlev = 0;
  f.no_frame = {0, 0};
  f.top = &f.no_frame;
  f.top_hi = 0;
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

const FrameRec* frame_of(const Fibre& f, uintptr_t fp) {
  for (int i = f.nf - 1; i >= 0; i--)
    if (f.fr[i].fp == fp) return &f.fr[i];
  return nullptr;
}

// < 0: a is behind b in the program (a runs first), > 0: b is behind a, 0: cannot tell them apart.
// Compared activation by activation from the kernel inwards: inside one activation first the loops (fewer back-edges taken =
// behind; not in the loop at all = before it or past it, by its block), then the addresses (lower = behind: forward code).
void read_chain(Fibre& f) {  // the parked fibre's stack is intact: frame pointers link its activations (fibre_main's saved one is 0)
  uintptr_t ra[kMaxLevels], rfp[kMaxLevels];
  int n = 0;
  for (const uintptr_t* fp = f.park_fp; fp != nullptr && fp[0] != 0 && n < kMaxLevels; fp = reinterpret_cast<const uintptr_t*>(fp[0])) {
    ra[n] = fp[1];
    rfp[n] = fp[0];
    n++;
  }
  f.nlev = n;
  for (int k = 0; k < n; k++) {
    f.ra[k] = ra[n - 1 - k];
    f.rfp[k] = rfp[n - 1 - k];
  }
}

int behind(Fibre& a, Fibre& b) {
  if (a.nlev < 0) read_chain(a);
  if (b.nlev < 0) read_chain(b);
  const int n = std::min(a.nlev, b.nlev);
  for (int k = 0; k < n; k++) {
    const uintptr_t fa = a.rfp[k], fb = b.rfp[k];
    const FrameRec* ba = frame_of(a, fa);
    const FrameRec* bb = frame_of(b, fb);
    int i = 0, j = 0;
    while (i < a.nl && a.lp[i].fp != fa) i++;
    while (j < b.nl && b.lp[j].fp != fb) j++;
    for (;;) {
      const bool ha = i < a.nl && a.lp[i].fp == fa, hb = j < b.nl && b.lp[j].fp == fb;
      if (!ha && !hb) break;
      if (ha && hb && a.lp[i].head == b.lp[j].head) {
        if (a.lp[i].iter != b.lp[j].iter) return a.lp[i].iter < b.lp[j].iter ? -1 : 1;
        i++, j++;
        continue;
      }
      // one of them is in a loop the other has no record of: the other is in front of it, in its first trip, or past it
      const bool a_has = ha && (!hb || a.lp[i].head < b.lp[j].head);
      const LoopRec& lr = a_has ? a.lp[i] : b.lp[j];
      const FrameRec* other = a_has ? bb : ba;
      const auto it = g_loops.find(lr.head);
      const uintptr_t latch = std::max(lr.latch, it == g_loops.end() ? (uintptr_t) 0 : it->second);
      const uintptr_t ob = other != nullptr ? other->block : 0;
      const bool other_first = ob <= latch;  // before the loop or in its first trip: the other is behind; past it: the one in the loop is
      return (a_has ? other_first : !other_first) ? 1 : -1;
    }
    if (a.ra[k] != b.ra[k]) return a.ra[k] < b.ra[k] ? -1 : 1;
  }
  return 0;
}

}  // namespace

// The compiler's basic-block instrumentation (-fsanitize-coverage=trace-pc on the kernel sources only): one call at the head of
// every block.  Keeps the running fibre's activations and loops up to date.
extern "C" void __sanitizer_cov_trace_pc() {
  Fibre* f = g_cur;
  if (f == nullptr) return;  // host code of the .hip files (launch wrappers)
  const uintptr_t* me = static_cast<const uintptr_t*>(__builtin_frame_address(0));
  const uintptr_t fp = me[0], pc = me[1];  // the instrumented function's frame and the block's address
  FrameRec* top = f->top;
  if (top->fp == fp && pc > top->block && pc <= f->top_hi) {
    top->block = pc;  // (the common case: further on in the same activation, inside the same loops)
    return;
  }
  while (f->nf > 0 && f->fr[f->nf - 1].fp < fp) f->nf--;  // activations that have returned (deeper = lower)
  while (f->nl > 0 && f->lp[f->nl - 1].fp < fp) f->nl--;
  if (f->nf > 0 && f->fr[f->nf - 1].fp == fp) {
    // (an activation of ANOTHER function at the same stack address -- one helper returned, the next was called -- lands here too:
    // what it inherits is dropped at its first block, which lies outside the other function's loops, and a loop it seems to have
    // taken is never compared with anything: two lanes in one helper at one call site have been told apart further out)
    const uintptr_t last = f->fr[f->nf - 1].block;
    while (f->nl > 0 && f->lp[f->nl - 1].fp == fp && (pc < f->lp[f->nl - 1].head || pc > f->lp[f->nl - 1].latch)) f->nl--;  // left that loop
    if (pc <= last) {  // a back-edge: from block `last` to block `pc`
      if (f->nl > 0 && f->lp[f->nl - 1].fp == fp && f->lp[f->nl - 1].head == pc) {
        f->lp[f->nl - 1].iter++;
        f->lp[f->nl - 1].latch = std::max(f->lp[f->nl - 1].latch, last);
      } else if (f->nl < kMaxLoops) {
        f->lp[f->nl++] = {fp, pc, last, 1};
      } else {
        g_counters[5]++;  // (a loop nest deeper than the record: the ordering falls back on addresses there; tests assert it never happens)
      }
      uintptr_t& g = g_loops[pc];
      g = std::max(g, last);
    }
    f->fr[f->nf - 1].block = pc;
  } else if (f->nf < kMaxFrames) {
    f->fr[f->nf++] = {fp, pc};
  } else {
    g_counters[5]++;
  }
  f->top = f->nf > 0 ? &f->fr[f->nf - 1] : &f->no_frame;
  f->top_hi = f->nl > 0 && f->lp[f->nl - 1].fp == fp ? f->lp[f->nl - 1].latch : ~(uintptr_t) 0;
}

namespace {

void run_block(int n_threads) {
  g_block_threads = n_threads;
  const int n_waves = (n_threads + 63) / 64;
  for (int t = 0; t < n_threads; t++) fibre_init(g_fibres[t], t, g_bdim);
  memset(g_dyn_lds, 0xCD, g_dyn_bytes);
  for (int round = 0;; round++) {
    for (int wi = 0; wi < n_waves; wi++) {
      // (the order in which the waves of a workgroup get to run between two barriers is the hardware's business: a kernel whose
      // results depend on it has a race, and shows it when the order is changed -- wemu_set_schedule)
      const int w = g_schedule == 0 ? wi : g_schedule == 1 ? n_waves - 1 - wi : (wi + round + (int) g_bid.x) % n_waves;
      Fibre* wave = g_fibres + 64 * w;
      const int n_lanes = std::min(64, n_threads - 64 * w);
      for (;;) {
        for (int l = 0; l < n_lanes; l++)
          if (wave[l].state == READY) run(wave[l]);
        // every lane of the wave is parked or done: the lanes that are furthest BEHIND in the program go next (with the lanes that
        // share their site, they form the EXEC mask) -- the order in which the hardware's reconvergence lets them run
        int best = -1;
        bool divergent = false;
        for (int l = 0; l < n_lanes; l++) {
          if (wave[l].state != PARKED || wave[l].op->kind >= BARRIER) continue;
          if (best < 0) best = l;
          else if (wave[l].op->site != wave[best].op->site) {
            divergent = true;
            if (behind(wave[l], wave[best]) < 0) best = l;
          }
        }
        if (best < 0) break;  // nothing but barriers and finished lanes
        const void* site = wave[best].op->site;
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
  f->park_fp = static_cast<const uintptr_t*>(__builtin_frame_address(0));  // (its call chain is read off the stack when it is needed)
  f->nlev = -1;
  // straight on to the wave's next runnable lane, if there is one (the scheduler would pick exactly that one)
  Fibre* wave = g_fibres + ((f - g_fibres) & ~63);
  const int n_lanes = std::min(64, g_block_threads - (int) (wave - g_fibres));
  for (int l = (int) (f - wave) + 1; l < n_lanes; l++)
    if (wave[l].state == READY) {
      Fibre& n = wave[l];
      g_cur = &n;
      g_tid = n.tid;
      g_lane = n.lane;
      wemu_switch(&f->sp, n.sp);
      return op.result;
    }
  wemu_switch(&f->sp, g_sched_sp);
  return op.result;
}

// DPP source lane of `lane` under dpp_ctrl (gfx9 encodings); -1: no valid source in the row
static inline int dpp_src(int lane, int ctrl) {
  const int row = lane & ~15, c = lane & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);                   // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl & 15; return c + n <= 15 ? row + c + n : -1; }     // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl & 15; return c - n >= 0 ? row + c - n : -1; }      // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl & 15; return row + ((c - n) & 15); }               // row_ror
  if (ctrl == 0x130) return lane + 1 <= 63 ? lane + 1 : -1;   // wave_shl:1
  if (ctrl == 0x134) return (lane + 1) & 63;                   // wave_rol:1
  if (ctrl == 0x138) return lane - 1 >= 0 ? lane - 1 : -1;    // wave_shr:1
  if (ctrl == 0x13C) return (lane - 1) & 63;                   // wave_ror:1
  if (ctrl == 0x140) return row + 15 - c;                      // row_mirror
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));    // row_half_mirror
  fprintf(stderr, "wave_emul: dpp_ctrl 0x%x\n", ctrl);        // (row_bcast15 / row_bcast31 write lanes of OTHER rows: not used here)
  abort();
}

static inline uint32_t permute(const void* site, uint32_t v, int src, uint32_t fallback) {
  Op op{};
  op.kind = PERMUTE; op.site = site; op.val = v; op.src = src; op.fallback = fallback;
  return (uint32_t) park(op);
}

// HIP's __shfl family (amd_warp_functions.h): a source outside the lane's segment of `width` reads the lane's own value
// (up / down / xor); a source lane that is inactive reads 0 (ds_bpermute_b32)
uint32_t xl_shfl(const void* site, uint32_t v, int mode, int arg, int width) {
  const int self = g_lane;
  int src = self;
  switch (mode) {
    case SHFL_IDX: src = (arg & (width - 1)) + (self & ~(width - 1)); break;
    case SHFL_UP: { const int i = self - arg; src = i < (self & ~(width - 1)) ? self : i; break; }
    case SHFL_DOWN: src = (self & (width - 1)) + arg >= width ? self : self + arg; break;
    case SHFL_XOR: { const int i = self ^ arg; src = i >= ((self + width) & ~(width - 1)) ? self : i; break; }
  }
  return permute(site, v, src, 0u);
}

// v_mov_b32_dpp: a lane whose row / bank is masked off keeps `old` (it still takes part: its register is a source); a lane whose
// source does not exist in the row or is inactive gets 0 under bound_ctrl, `old` otherwise
int xl_dpp(const void* site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = g_lane;
  const bool enabled = ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane & 15) >> 2)) & 1);
  const uint32_t fb = enabled && bound_ctrl ? 0u : (uint32_t) old;
  return (int) permute(site, (uint32_t) src, enabled ? dpp_src(lane, ctrl) : -1, fb);
}

unsigned long long xl_ballot(const void* site, int pred) {
  Op op{};
  op.kind = BALLOT; op.site = site; op.val = pred != 0;
  return park(op);
}

int xl_first(const void* site, int v) {
  Op op{};
  op.kind = FIRST; op.site = site; op.val = (uint32_t) v;
  return (int) park(op);
}

int xl_barrier(int kind, int pred) {
  Op op{};
  op.kind = kind; op.val = pred != 0;
  return (int) park(op);
}

void xl_mfma(const void* site, int kind, const void* a, const void* b, const void* c, void* d) {
  Op op{};
  op.kind = kind; op.site = site; op.a = a; op.b = b; op.c = c; op.d = d;
  park(op);
}

void* dyn_lds() { return g_dyn_lds; }

void launch(const char* kernel, const void* stream, dim3 grid, dim3 block, size_t dyn_lds_bytes, Body body, void* closure) {
  const size_t n_threads = (size_t) block.x * block.y * block.z;
  if (n_threads == 0 || n_threads > (size_t) kMaxThreads || dyn_lds_bytes > kDynLds) {
    fprintf(stderr, "wave_emul: launch of %zu work-items per group / %zu bytes of dynamic LDS\n", n_threads, dyn_lds_bytes);
    abort();
  }
  if (g_stacks == nullptr) {
    g_stacks = static_cast<char*>(mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (g_stacks == MAP_FAILED || posix_memalign(reinterpret_cast<void**>(&g_dyn_lds), 256, kDynLds) != 0) abort();
    for (int t = 0; t < kMaxThreads; t++) {
      g_fibres[t].stack = g_stacks + kStack * t;
      mprotect(g_fibres[t].stack, 4096, PROT_NONE);  // a work-item that outgrows its stack faults instead of writing into its neighbour's
    }
  }
  if (g_image_hi == 0) {
    dl_iterate_phdr([](dl_phdr_info* info, size_t, void*) {
      const uintptr_t me = reinterpret_cast<uintptr_t>(&g_image_lo);
      uintptr_t lo = ~(uintptr_t) 0, hi = 0;
      for (int i = 0; i < info->dlpi_phnum; i++)
        if (info->dlpi_phdr[i].p_type == PT_LOAD) {
          lo = std::min(lo, (uintptr_t) (info->dlpi_addr + info->dlpi_phdr[i].p_vaddr));
          hi = std::max(hi, (uintptr_t) (info->dlpi_addr + info->dlpi_phdr[i].p_vaddr + info->dlpi_phdr[i].p_memsz));
        }
      if (me >= lo && me < hi) {
        g_image_lo = lo;
        g_image_hi = hi;
        return 1;
      }
      return 0;
    }, nullptr);
  }
  {
    pthread_attr_t at;
    void* lo = nullptr;
    size_t sz = 0;
    if (pthread_getattr_np(pthread_self(), &at) == 0) {
      pthread_attr_getstack(&at, &lo, &sz);
      pthread_attr_destroy(&at);
      g_host_stack_lo = (uintptr_t) lo;
      g_host_stack_hi = (uintptr_t) lo + sz;
    }
  }
  if (g_hb_on) {
    g_cur_stream = (int) (reinterpret_cast<uintptr_t>(stream) & (kStreams - 1));
    VC& v = g_vc[g_cur_stream];
    for (int i = 0; i < kStreams; i++) v.c[i] = std::max(v.c[i], g_host_vc.c[i]);  // (the host issued this launch after what it had waited for)
    v.c[g_cur_stream]++;
    g_kernel_names.push_back(std::string(kernel) + " [stream " + std::to_string(g_cur_stream) + "]");
    g_cur_kernel = (int) g_kernel_names.size() - 1;
  }
  memset(g_tr, 0, sizeof(g_tr));
  g_ld_lines.clear();
  g_st_lines.clear();
  g_tracing = true;
  g_counters[0]++;
  g_dyn_bytes = std::min(kDynLds, (dyn_lds_bytes + 4095) & ~(size_t) 4095);  // (a page of poison behind what was asked for)
  g_body = body;
  g_closure = closure;
  g_bdim = block;
  g_gdim = grid;
  const size_t n_blocks = (size_t) grid.x * grid.y * grid.z;
  for (size_t i = 0; i < n_blocks; i++) {
    const size_t b = g_schedule == 0 ? i : g_schedule == 1 ? n_blocks - 1 - i : ((i & 1) ? n_blocks - 1 - i / 2 : i / 2);
    g_bid = {(unsigned) (b % grid.x), (unsigned) ((b / grid.x) % grid.y), (unsigned) (b / ((size_t) grid.x * grid.y))};
    run_block((int) n_threads);
    g_counters[4] += (long) n_threads;
  }
  g_cur = nullptr;
  g_tracing = false;
  if (g_tr[0] | g_tr[1] | g_tr[4] | g_tr[5]) {
    LaunchTraffic t;
    t.kernel = kernel;
    t.v[0] = g_tr[0]; t.v[1] = g_tr[1]; t.v[2] = g_ld_lines.size(); t.v[3] = g_st_lines.size(); t.v[4] = g_tr[4]; t.v[5] = g_tr[5];
    t.v[6] = n_blocks; t.v[7] = n_threads;
    g_traffic.push_back(t);
  }
}

extern "C" void wemu_hb_enable(int on) {
  g_hb_on = on != 0;
  memset(g_vc, 0, sizeof(g_vc));
  memset(&g_host_vc, 0, sizeof(g_host_vc));
  g_tokens.clear();
  g_shadow.clear();
  g_races.clear();
  g_kernel_names.clear();
}
extern "C" long wemu_hb_record(int stream) {
  if (!g_hb_on) return -1;
  VC v = g_vc[stream & (kStreams - 1)];
  for (int i = 0; i < kStreams; i++) v.c[i] = std::max(v.c[i], g_host_vc.c[i]);  // (recorded by the host after what it had waited for)
  g_tokens.push_back(v);
  return (long) g_tokens.size() - 1;
}
extern "C" void wemu_hb_wait(int stream, long token) {
  static const bool ignore = getenv("WEMU_HB_IGNORE_WAITS") != nullptr;  // (the detector's positive control: a host that never waits)
  if (!g_hb_on || token < 0 || ignore) return;
  VC& v = g_vc[stream & (kStreams - 1)];
  for (int i = 0; i < kStreams; i++) v.c[i] = std::max(v.c[i], g_tokens[token].c[i]);
}
extern "C" void wemu_hb_host_sync(long token) {  // the host waited for an event (token >= 0) or for the whole device (token < 0)
  if (!g_hb_on) return;
  if (token >= 0) {
    for (int i = 0; i < kStreams; i++) g_host_vc.c[i] = std::max(g_host_vc.c[i], g_tokens[token].c[i]);
  } else {
    for (int st = 0; st < kStreams; st++)
      for (int i = 0; i < kStreams; i++) g_host_vc.c[i] = std::max(g_host_vc.c[i], g_vc[st].c[i]);
  }
}
extern "C" int wemu_hb_races(void) { return (int) g_races.size(); }
// words that more than one stream touched (ordered or not), by the kernel that wrote them last: which buffers cross streams at all
extern "C" int wemu_hb_shared_words(char* out, int size) {
  std::unordered_map<std::string, long> by_writer;
  long total = 0, all = 0;
  for (auto& kv : g_shadow) {
    all++;
    if (__builtin_popcount(kv.second.streams) < 2) continue;
    total++;
    std::string w = kv.second.w_stream >= 0 ? g_kernel_names[kv.second.w_kernel] : std::string("(never written by a kernel: inputs)");
    by_writer[w.substr(0, w.find(" [stream"))]++;
  }
  std::vector<std::pair<long, std::string>> v;
  for (auto& kv : by_writer) v.push_back({kv.second, kv.first});
  std::sort(v.rbegin(), v.rend());
  std::string txt = std::to_string(total) + " of " + std::to_string(all) + " words touched by kernels were touched from more than one stream; last written by:";
  for (auto& e : v) txt += "\n#     " + std::to_string(e.first) + "  " + e.second;
  snprintf(out, (size_t) size, "%s", txt.c_str());
  return (int) v.size();
}
extern "C" long wemu_hb_race(int i, char* first, char* second, char* kind, int size, unsigned long long* addr) {
  if (i < 0 || i >= (int) g_races.size()) return -1;
  snprintf(first, (size_t) size, "%s", g_races[i].first.c_str());
  snprintf(second, (size_t) size, "%s", g_races[i].second.c_str());
  snprintf(kind, (size_t) size, "%s", g_races[i].kind.c_str());
  *addr = g_races[i].addr;
  return g_races[i].count;
}
extern "C" int wemu_traffic_launches(void) { return (int) g_traffic.size(); }
extern "C" void wemu_traffic_reset(void) { g_traffic.clear(); }
extern "C" int wemu_traffic_get(int i, char* name, int name_size, unsigned long long* out8) {
  if (i < 0 || i >= (int) g_traffic.size()) return -1;
  snprintf(name, (size_t) name_size, "%s", g_traffic[i].kernel.c_str());
  memcpy(out8, g_traffic[i].v, sizeof(g_traffic[i].v));
  return 0;
}

namespace {
inline void traffic(uintptr_t a, unsigned bytes, bool store) {
  if (!g_tracing || g_cur == nullptr) return;
  if (a >= (uintptr_t) g_stacks && a < (uintptr_t) g_stacks + kStack * kMaxThreads) return;  // registers / private memory
  if (a >= g_host_stack_lo && a < g_host_stack_hi) return;                                     // kernel arguments
  const bool lds = (a >= (uintptr_t) g_dyn_lds && a < (uintptr_t) g_dyn_lds + kDynLds) || (a >= g_image_lo && a < g_image_hi);
  if (lds) {
    g_tr[store ? 5 : 4] += bytes;
    return;
  }
  g_tr[store ? 1 : 0] += bytes;
  if (g_hb_on) {
    for (unsigned off = 0; off < bytes; off += 4) hb_access(a + off, store);  // (word by word: two kernels may share a line, not a word)
    if (((a + bytes - 1) >> 2) != ((a + ((bytes - 1) & ~3u)) >> 2)) hb_access(a + bytes - 1, store);
    return;
  }
  (store ? g_st_lines : g_ld_lines).insert(a >> 7);
  if (((a + bytes - 1) >> 7) != (a >> 7)) (store ? g_st_lines : g_ld_lines).insert((a + bytes - 1) >> 7);
}
}  // namespace
extern "C" void __sanitizer_cov_load1(void* p) { traffic((uintptr_t) p, 1, false); }
extern "C" void __sanitizer_cov_load2(void* p) { traffic((uintptr_t) p, 2, false); }
extern "C" void __sanitizer_cov_load4(void* p) { traffic((uintptr_t) p, 4, false); }
extern "C" void __sanitizer_cov_load8(void* p) { traffic((uintptr_t) p, 8, false); }
extern "C" void __sanitizer_cov_load16(void* p) { traffic((uintptr_t) p, 16, false); }
extern "C" void __sanitizer_cov_store1(void* p) { traffic((uintptr_t) p, 1, true); }
extern "C" void __sanitizer_cov_store2(void* p) { traffic((uintptr_t) p, 2, true); }
extern "C" void __sanitizer_cov_store4(void* p) { traffic((uintptr_t) p, 4, true); }
extern "C" void __sanitizer_cov_store8(void* p) { traffic((uintptr_t) p, 8, true); }
extern "C" void __sanitizer_cov_store16(void* p) { traffic((uintptr_t) p, 16, true); }

extern "C" void wemu_set_schedule(int mode) { g_schedule = mode; }
extern "C" long wemu_counter(int which) { return which >= 0 && which < 8 ? g_counters[which] : -1; }

}  // namespace wemu
