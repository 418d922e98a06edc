// Ray source (mirrors the ray-generation half of src/Dataset/Dataset.h).  Cameras AND images are
// resident in HBM; camera / pixel indices are drawn on the device, rays come from f2n_img2world_rays and colours from
// f2n_gather_pixels -- no per-iteration upload (the reference draws on the CPU and copies rays and colours every
// iteration, Dataset.cpp:275-298).  File IO (cams_meta.npy, JPEG decoding, NormalizeScene) is out of scope: the
// constructor takes the tensors the reference's constructor would have produced.
#pragma once
#include <map>
#include <tuple>

#include "GlobalDataPool.h"
#include "KeyedDraws.h"

namespace f2n {

#define DATA_TRAIN_SET 1
#define DATA_TEST_SET 2
#define DATA_VAL_SET 4

struct Rays {
  Tensor origins, dirs;
};
struct BoundedRays {
  Tensor origins, dirs, bounds;  // bounds: near, far
};

Tensor PoseInterpolate(const Tensor& pose_a, const Tensor& pose_b, float alpha);  // Utils/CameraUtils.cpp:11-44

class Dataset {
 public:
  // poses [C,3,4] c2w, intri [C,3,3], dist_params [C,4], bounds [C,2]; images fp32 [C,H,W,3] in [0,1] (may be
  // undefined: then ground-truth colours are not available and only ray generation works)
  Dataset(const Tensor& poses, const Tensor& intri, const Tensor& dist_params, const Tensor& bounds, const Tensor& images,
          int height, int width, const std::vector<int>& train_set, const std::vector<int>& test_set,
          const std::vector<int>& val_set);

  BoundedRays RaysOfCamera(int idx, int reso_level = 1);                      // Dataset.cpp:174-192
  BoundedRays RaysFromPose(const Tensor& pose, int reso_level = 1);           // :194-214
  BoundedRays RandRaysFromPose(int batch_size, const Tensor& pose);           // :216-230
  BoundedRays RaysInterpolate(int idx_0, int idx_1, float alpha, int reso_level = 1);
  BoundedRays RandRaysWholeSpace(int batch_size);                             // :245-255
  std::tuple<BoundedRays, Tensor, Tensor> RandRaysDataOfCamera(int idx, int batch_size);
  // seq >= 0: the batch's sequence number in a training run -- its three uniforms per ray are then draw `seq` of the ray purpose
  // (KeyedDraws.h), whenever and however often the batch is drawn; seq < 0: the next draw of this data set's own sequence
  std::tuple<BoundedRays, Tensor, Tensor> RandRaysData(int batch_size, int sets, int64_t seq = -1);  // :275-298

  Rays Img2WorldRay(int cam_idx, const Tensor& ij);
  Rays Img2WorldRay(const Tensor& pose, const Tensor& intri, const Tensor& dist_params, const Tensor& ij);
  Rays Img2WorldRayFlex(const Tensor& cam_indices, const Tensor& ij);         // Dataset.cu:125-152
  Tensor GatherColors(const Tensor& cam_indices, const Tensor& ij);           // the index expression of Dataset.cpp:293

  int n_images_ = 0, height_ = 0, width_ = 0;
  Tensor poses_, intri_, dist_params_, bounds_;  // device
  Tensor poses_cpu_;                             // for the host-side pose blending of RandRaysWholeSpace
  Tensor image_tensors_;                         // device, [C,H,W,3] fp32
  std::vector<int> train_set_, test_set_, val_set_;
  std::map<int, Tensor> set_on_device_;  // RandRaysData: the image indices of a `sets` combination, uploaded once
  Tensor last_cam_indices_, last_ij_;            // the draws behind the most recent Rand* batch (tests, logging)
  KeyedUniforms ray_draws_{0xA0761D6478BD642Full};

 private:
  Tensor PixelGrid(int H_out, int W_out);  // all (row, col) of the full-resolution image sub-sampled to H_out x W_out
  float bounds_min_near_ = 0.f, bounds_max_far_ = 0.f;
};

}  // namespace f2n
