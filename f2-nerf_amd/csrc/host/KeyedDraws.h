// Counter-based random draws.  Every uniform a training run consumes is keyed by WHAT IT IS FOR -- (seed, purpose, sequence
// number) -- instead of by its position in one generator's stream: the march noise of batch k, the background colours and edge
// samples of step k and the rays of batch k are the k-th draw of their purpose whether they are made two steps ahead on a side
// stream, at the top of their step, a second time after a prefetched batch was dropped, or by a Train() call that starts at
// iteration k.  (Rounds 3-4 kept one running generator per purpose: a batch that was begun speculatively, dropped and begun again
// drew its noise twice and shifted every later batch's noise -- trainings then depended on the sampling schedule, and, through
// the timing-dependent decision to speculate, on what else ran on the GPU.)
//
// Implementation: one Philox generator per purpose, seeded from the default generator's seed (torch::manual_seed is followed
// lazily, as before); a draw sets the generator's Philox offset to seq * kStride and then draws -- ATen's uniform kernel consumes
// at most a few counter steps per call at the sizes used here (<= 1e6 values), kStride leaves room for 2^14 of them.
#pragma once
#include <ATen/hip/HIPGeneratorImpl.h>

#include "Common.h"

namespace f2n {

class KeyedUniforms {
 public:
  explicit KeyedUniforms(uint64_t purpose) : purpose_(purpose) {}
  static constexpr uint64_t kStride = 1ull << 16;   // Philox offset between consecutive sequence numbers (a multiple of 4)
  static constexpr int64_t kOwnBase = 1ll << 40;    // sequence numbers of draws nobody keyed (seq < 0): this object's own count
  // Data-parallel replicas (one process per GPU): rank r draws stream r of every purpose -- its own rays in ExpRunner::Train, its own
  // march noise, background colours and edge samples -- while the seed, and with it everything the replicas must agree on, stays the
  // same on every rank.  Set by DataParallel::Attach / parallel.attach; 0 on one GPU and on rank 0, whose draws are the single-GPU ones.
  static inline uint64_t replica_salt = 0;
  static uint64_t SaltOf(int rank) { return (uint64_t) rank * 0x9FB21C651E98DF25ull; }  // (odd multiplier: distinct for distinct ranks)
  static void SetReplica(int rank) { replica_salt = SaltOf(rank); }
  static uint64_t KeyOf(uint64_t seed, uint64_t purpose, uint64_t salt) { return seed ^ purpose ^ salt; }
  // n uniforms in [0, 1) on the current stream: draw number `seq` of this purpose under the default generator's current seed.
  // seq < 0: the next of this object's own running sequence (callers outside a training loop: tests, the plugin entry points).
  Tensor Draw(int64_t n, int64_t seq) {
    const int dev = c10::hip::current_device();
    const uint64_t seed = at::cuda::detail::getDefaultCUDAGenerator(dev).current_seed();
    if (!gen_.defined() || seed != seed_ || replica_salt != salt_ || gen_.device().index() != dev) {
      gen_ = at::cuda::detail::createCUDAGenerator(dev);
      gen_.set_current_seed(KeyOf(seed, purpose_, replica_salt));
      seed_ = seed;
      salt_ = replica_salt;
      own_seq_ = 0;  // (a new seed starts this object's own sequence again: torch::manual_seed(s) replays unkeyed draws)
    }
    const uint64_t s = seq >= 0 ? (uint64_t) seq : (uint64_t) (kOwnBase + own_seq_++);
    gen_.set_offset(s * kStride);
    return torch::rand({n}, gen_, DevF32());
  }
  // The same key without a draw: {seed ^ purpose, sequence number} for kernels that make their own uniforms (Philox4x32-10 in
  // csrc/f2n_dev.h: f2n_draw_ray_batch_keyed, f2n_sampler_prologue_keyed) -- one launch less than rand + consumer.
  struct Key {
    uint64_t key, seq;
  };
  Key KeyFor(int64_t seq) {
    const int dev = c10::hip::current_device();
    const uint64_t seed = at::cuda::detail::getDefaultCUDAGenerator(dev).current_seed();
    if (!keyed_ || seed != key_seed_) {
      key_seed_ = seed;
      keyed_ = true;
      own_seq_ = 0;
    }
    return {KeyOf(seed, purpose_, replica_salt), seq >= 0 ? (uint64_t) seq : (uint64_t) (kOwnBase + own_seq_++)};
  }
  void Rewind() { own_seq_ = 0; }

 private:
  uint64_t purpose_;
  at::Generator gen_;
  uint64_t seed_ = 0, key_seed_ = 0, salt_ = 0;
  bool keyed_ = false;
  int64_t own_seq_ = 0;
};

}  // namespace f2n
