// Plugin interface of the point sampler (mirrors src/PtsSampler/PtsSampler.h:13-48).
#pragma once
#include "GlobalDataPool.h"
#include "Pipe.h"

namespace f2n {

struct SampleResultFlex {
  Tensor pts;             // [ n_all_pts, 3 ]  warped coordinates
  Tensor dirs;            // [ n_all_pts, 3 ]
  Tensor dt;              // [ n_all_pts ]
  Tensor t;               // [ n_all_pts ]
  Tensor anchors;         // [ n_all_pts, 3 ]  (trans_idx, leaf node idx, 0)
  Tensor pts_idx_bounds;  // [ n_rays, 2 ]     start, end
  Tensor first_oct_dis;   // [ n_rays, 1 ]
  // rows allocated (not filled) IN FRONT of pts and anchors (which are views starting extra_rows rows into their storage): a
  // training step parks its 2E edge samples there so that the density pre-pass gathers their hash features in the same
  // launches -- in front, because that offset is known before the sample count is
  int extra_rows = 0;
  // Streaming training steps (Renderer::PreGenerateStepDraws): the step's background colours and its 2E edge samples, generated
  // on the sampler's side stream right behind the pack -- the edge points already sit in the extra rows above and at the head of
  // pts_all / vol_all (worst-case-sized homes of the grad pass's point / warp-index arrays) -- so that the step's main queue
  // starts with the hash gather instead of a draw launch and an edge-sample launch.
  bool step_draws_ready = false;
  Tensor bg_color, pts_all, vol_all;
};

class PtsSampler : public Pipe {
 public:
  virtual SampleResultFlex GetSamples(const Tensor& rays_o, const Tensor& rays_d, const Tensor& bounds) = 0;
  virtual std::tuple<Tensor, Tensor> GetEdgeSamples(int n_pts) = 0;
  virtual void UpdateOctNodes(const SampleResultFlex& sample_result, const Tensor& sampled_weights,
                              const Tensor& sampled_alpha) = 0;
  GlobalDataPool* global_data_pool_ = nullptr;
};

std::unique_ptr<PtsSampler> ConstructPtsSampler(GlobalDataPool* global_data_pool);

}  // namespace f2n
