// Dataset: ray generation + data residency.
#include "Dataset.h"

#include <cmath>

namespace f2n {

// Utils/CameraUtils.cpp:11-44: rotation slerp (Eigen::Quaternionf semantics) + translation lerp.  Host arithmetic on a
// 3x4 pose; not part of the bit-exact contract (the reference's random poses come from a different RNG stream anyway).
Tensor PoseInterpolate(const Tensor& pose_a, const Tensor& pose_b, float alpha) {
  Tensor a = pose_a.to(torch::kCPU).to(torch::kFloat32).contiguous(), b = pose_b.to(torch::kCPU).to(torch::kFloat32).contiguous();
  auto to_quat = [](const float* m, float* q /*w,x,y,z*/) {  // Eigen's rotation-matrix -> quaternion (Shepperd)
    auto M = [&](int i, int j) { return m[4 * i + j]; };
    float t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.f) {
      t = std::sqrt(t + 1.f);
      q[0] = .5f * t;
      t = .5f / t;
      q[1] = (M(2, 1) - M(1, 2)) * t;
      q[2] = (M(0, 2) - M(2, 0)) * t;
      q[3] = (M(1, 0) - M(0, 1)) * t;
    } else {
      int i = 0;
      if (M(1, 1) > M(0, 0)) i = 1;
      if (M(2, 2) > M(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.f);
      q[1 + i] = .5f * t;
      t = .5f / t;
      q[0] = (M(k, j) - M(j, k)) * t;
      q[1 + j] = (M(j, i) + M(i, j)) * t;
      q[1 + k] = (M(k, i) + M(i, k)) * t;
    }
  };
  float qa[4], qb[4];
  to_quat(a.data_ptr<float>(), qa);
  to_quat(b.data_ptr<float>(), qb);
  // Eigen::QuaternionBase::slerp
  const float one = 1.f - 1.1920928955078125e-07f;
  const float d = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
  const float ad = std::fabs(d);
  float s0, s1;
  if (ad >= one) {
    s0 = 1.f - alpha;
    s1 = alpha;
  } else {
    const float theta = std::acos(ad), st = std::sin(theta);
    s0 = std::sin((1.f - alpha) * theta) / st;
    s1 = std::sin(alpha * theta) / st;
  }
  if (d < 0.f) s1 = -s1;
  float q[4];
  float nrm = 0.f;
  for (int i = 0; i < 4; i++) {
    q[i] = s0 * qa[i] + s1 * qb[i];
    nrm += q[i] * q[i];
  }
  nrm = std::sqrt(nrm);
  for (int i = 0; i < 4; i++) q[i] /= nrm;
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  Tensor ret = torch::zeros({3, 4}, CpuF32());
  float* r = ret.data_ptr<float>();
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
              tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz;       r[2] = txz + twy;
  r[4] = txy + twz;       r[5] = 1 - (txx + tzz); r[6] = tyz - twx;
  r[8] = txz - twy;       r[9] = tyz + twx;       r[10] = 1 - (txx + tyy);
  const float *pa = a.data_ptr<float>(), *pb = b.data_ptr<float>();
  for (int i = 0; i < 3; i++) r[4 * i + 3] = pa[4 * i + 3] * (1.f - alpha) + pb[4 * i + 3] * alpha;
  return ret.to(pose_a.device());
}

Dataset::Dataset(const Tensor& poses, const Tensor& intri, const Tensor& dist_params, const Tensor& bounds, const Tensor& images,
                 int height, int width, const std::vector<int>& train_set, const std::vector<int>& test_set,
                 const std::vector<int>& val_set) {
  n_images_ = poses.size(0);
  height_ = height;
  width_ = width;
  TORCH_CHECK(poses.dim() == 3 && poses.size(1) == 3 && poses.size(2) == 4, "poses must be [C,3,4]");
  TORCH_CHECK(intri.size(0) == n_images_ && dist_params.size(0) == n_images_ && bounds.size(0) == n_images_, "camera count mismatch");
  poses_ = poses.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  intri_ = intri.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  dist_params_ = dist_params.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  bounds_ = bounds.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  poses_cpu_ = poses_.to(torch::kCPU);
  Tensor bc = bounds_.to(torch::kCPU);
  bounds_min_near_ = bc.index({Slc(), 0}).min().item<float>();
  bounds_max_far_ = bc.index({Slc(), 1}).max().item<float>();
  if (images.defined() && images.numel() > 0) {
    TORCH_CHECK(images.dim() == 4 && images.size(0) == n_images_ && images.size(1) == height && images.size(2) == width &&
                    images.size(3) == 3, "images must be [C,H,W,3]");
    image_tensors_ = images.to(torch::kCUDA).to(torch::kFloat32).contiguous();
  }
  train_set_ = train_set;
  test_set_ = test_set;
  val_set_ = val_set;
}

Rays Dataset::Img2WorldRayFlex(const Tensor& cam_indices, const Tensor& ij) {
  Tensor cam = cam_indices.to(torch::kCUDA).to(torch::kInt32).contiguous();
  Tensor px = ij.to(torch::kCUDA).to(torch::kInt32).contiguous();
  const int n = cam.size(0);
  TORCH_CHECK(px.numel() == 2 * (int64_t) n, "ij must be [n,2]");
  Tensor rays_o = torch::empty({n, 3}, DevF32()), rays_d = torch::empty({n, 3}, DevF32());
  F2N_CALL(f2n_img2world_rays(CurStream(), n, F32P(poses_), F32P(intri_), F32P(dist_params_), I32P(cam), I32P(px), F32P(rays_o),
                              F32P(rays_d)));
  return {rays_o, rays_d};
}

Rays Dataset::Img2WorldRay(int cam_idx, const Tensor& ij) {
  return Img2WorldRayFlex(torch::full({ij.size(0)}, cam_idx, DevI32()), ij);
}

Rays Dataset::Img2WorldRay(const Tensor& pose, const Tensor& intri, const Tensor& dist_params, const Tensor& ij) {
  // a one-camera table: same arithmetic as the per-camera kernel (the reference spells this variant with ATen ops,
  // Dataset.cpp:152-172)
  Tensor p = pose.to(torch::kCUDA).to(torch::kFloat32).reshape({1, 3, 4}).contiguous();
  Tensor k = intri.to(torch::kCUDA).to(torch::kFloat32).reshape({1, 3, 3}).contiguous();
  Tensor d = dist_params.to(torch::kCUDA).to(torch::kFloat32).reshape({1, 4}).contiguous();
  Tensor px = ij.to(torch::kCUDA).to(torch::kInt32).contiguous();
  const int n = px.size(0);
  Tensor cam = torch::zeros({n}, DevI32());
  Tensor rays_o = torch::empty({n, 3}, DevF32()), rays_d = torch::empty({n, 3}, DevF32());
  F2N_CALL(f2n_img2world_rays(CurStream(), n, F32P(p), F32P(k), F32P(d), I32P(cam), I32P(px), F32P(rays_o), F32P(rays_d)));
  return {rays_o, rays_d};
}

Tensor Dataset::GatherColors(const Tensor& cam_indices, const Tensor& ij) {
  TORCH_CHECK(image_tensors_.defined(), "no images resident");
  Tensor cam = cam_indices.to(torch::kCUDA).to(torch::kInt32).contiguous();
  Tensor px = ij.to(torch::kCUDA).to(torch::kInt32).contiguous();
  const int n = cam.size(0);
  Tensor colors = torch::empty({n, 3}, DevF32());
  F2N_CALL(f2n_gather_pixels(CurStream(), n, height_, width_, F32P(image_tensors_), nullptr, I32P(cam), I32P(px), F32P(colors),
                             nullptr));
  return colors;
}

Tensor Dataset::PixelGrid(int H_out, int W_out) {  // linspace(0, H-1, H_out) x linspace(0, W-1, W_out), truncated to pixels
  Tensor ii = torch::linspace(0.f, height_ - 1.f, H_out, DevF32()).to(torch::kInt32);
  Tensor jj = torch::linspace(0.f, width_ - 1.f, W_out, DevF32()).to(torch::kInt32);
  auto g = torch::meshgrid({ii, jj}, "ij");
  return torch::stack({g[0].reshape({-1}), g[1].reshape({-1})}, -1).contiguous();
}

// (reso_level is accepted and ignored, as in the reference: Dataset.cpp:177-179 renders H = height_, W = width_)
BoundedRays Dataset::RaysOfCamera(int idx, int /*reso_level*/) {
  TORCH_CHECK(idx >= 0 && idx < n_images_, "camera index out of range");
  Tensor ij = PixelGrid(height_, width_);
  auto rays = Img2WorldRay(idx, ij);
  Tensor b = bounds_.index({idx}).reshape({1, 2}).repeat({ij.size(0), 1}).contiguous();
  return {rays.origins, rays.dirs, b};
}

BoundedRays Dataset::RaysFromPose(const Tensor& pose, int reso_level) {
  const int H = height_ / reso_level, W = width_ / reso_level;
  Tensor ij = PixelGrid(H, W);
  auto rays = Img2WorldRay(pose, intri_[0], dist_params_[0], ij);
  Tensor b = torch::stack({torch::full({H * W}, bounds_min_near_, DevF32()), torch::full({H * W}, bounds_max_far_, DevF32())}, -1)
                 .contiguous();
  return {rays.origins, rays.dirs, b};
}

BoundedRays Dataset::RandRaysFromPose(int batch_size, const Tensor& pose) {
  Tensor i = torch::randint(0, height_, {batch_size}, DevI32()), j = torch::randint(0, width_, {batch_size}, DevI32());
  last_ij_ = torch::stack({i, j}, -1).contiguous();
  last_cam_indices_ = Tensor();
  auto rays = Img2WorldRay(pose, intri_[0], dist_params_[0], last_ij_);
  Tensor b = torch::stack({torch::full({batch_size}, bounds_min_near_, DevF32()), torch::full({batch_size}, bounds_max_far_, DevF32())},
                          -1)
                 .contiguous();
  return {rays.origins, rays.dirs, b};
}

BoundedRays Dataset::RaysInterpolate(int idx_0, int idx_1, float alpha, int reso_level) {
  return RaysFromPose(PoseInterpolate(poses_cpu_[idx_0], poses_cpu_[idx_1], alpha), reso_level);
}

BoundedRays Dataset::RandRaysWholeSpace(int batch_size) {
  const int window_size = 10;
  TORCH_CHECK(n_images_ > window_size, "RandRaysWholeSpace needs more than 10 cameras");
  Tensor weights = torch::rand({3}, CpuF32()) + 1e-7f;
  Tensor indices = torch::randint(0, window_size, {3}, CpuI32()) + torch::randint(0, n_images_ - window_size, {1}, CpuI32());
  const int a = indices[0].item<int>(), b = indices[1].item<int>(), c = indices[2].item<int>();
  const float wa = weights[0].item<float>(), wb = weights[1].item<float>(), wc = weights[2].item<float>();
  Tensor pose = PoseInterpolate(poses_cpu_[a], poses_cpu_[b], wb / (wb + wa));
  pose = PoseInterpolate(pose, poses_cpu_[c], wc / (wa + wb + wc));
  return RandRaysFromPose(batch_size, pose);
}

std::tuple<BoundedRays, Tensor, Tensor> Dataset::RandRaysDataOfCamera(int idx, int batch_size) {
  Tensor cam = torch::full({batch_size}, idx, DevI32());
  Tensor i = torch::randint(0, height_, {batch_size}, DevI32()), j = torch::randint(0, width_, {batch_size}, DevI32());
  Tensor ij = torch::stack({i, j}, -1).contiguous();
  last_cam_indices_ = cam;
  last_ij_ = ij;
  auto rays = Img2WorldRayFlex(cam, ij);
  Tensor colors = torch::empty({batch_size, 3}, DevF32()), b = torch::empty({batch_size, 2}, DevF32());
  F2N_CALL(f2n_gather_pixels(CurStream(), batch_size, height_, width_, image_tensors_.defined() ? F32P(image_tensors_) : nullptr,
                             F32P(bounds_), I32P(cam), I32P(ij), image_tensors_.defined() ? F32P(colors) : nullptr, F32P(b)));
  return {{rays.origins, rays.dirs, b}, image_tensors_.defined() ? colors : Tensor(), cam};
}

std::tuple<BoundedRays, Tensor, Tensor> Dataset::RandRaysData(int batch_size, int sets, int64_t seq) {
  // The image list of a set lives on the device, uploaded once per `sets` value: the upload of a pageable host array is a
  // SYNCHRONOUS copy -- issued every iteration (as a first version did) it made the host wait for the whole previous training
  // step before it could queue the next one: ~0.2 ms of idle device at the head of every step of ExpRunner::Train
  // (profiles/r04_native_loop_timeline.txt).
  auto it = set_on_device_.find(sets);
  if (it == set_on_device_.end()) {
    std::vector<int> img_idx;
    if ((sets & DATA_TRAIN_SET) != 0) img_idx.insert(img_idx.end(), train_set_.begin(), train_set_.end());
    if ((sets & DATA_VAL_SET) != 0) img_idx.insert(img_idx.end(), val_set_.begin(), val_set_.end());
    if ((sets & DATA_TEST_SET) != 0) img_idx.insert(img_idx.end(), test_set_.begin(), test_set_.end());
    TORCH_CHECK(!img_idx.empty(), "empty image set");
    it = set_on_device_.emplace(sets, torch::from_blob(img_idx.data(), {(int64_t) img_idx.size()}, CpuI32()).to(torch::kCUDA)).first;
  }
  const Tensor& cur_set = it->second;
  // every draw on the device: uniform image of the set, uniform pixel (Dataset.cpp:286-291) -- one uniform launch and ONE kernel
  // that maps the draws, generates the rays and gathers colours and bounds (f2n_draw_ray_batch)
  Tensor cam = torch::empty({batch_size}, DevI32()), ij = torch::empty({batch_size, 2}, DevI32());
  Tensor rays_o = torch::empty({batch_size, 3}, DevF32()), rays_d = torch::empty({batch_size, 3}, DevF32());
  Tensor colors = torch::empty({batch_size, 3}, DevF32()), b = torch::empty({batch_size, 2}, DevF32());
  const auto key = ray_draws_.KeyFor(seq);  // (the kernel draws its own three uniforms per ray: no rand launch on the step's main queue)
  F2N_CALL(f2n_draw_ray_batch_keyed(CurStream(), batch_size, key.key, key.seq, I32P(cur_set), (int) cur_set.size(0), height_, width_, F32P(poses_),
                              F32P(intri_), F32P(dist_params_), image_tensors_.defined() ? F32P(image_tensors_) : nullptr, F32P(bounds_),
                              I32P(cam), I32P(ij), F32P(rays_o), F32P(rays_d), image_tensors_.defined() ? F32P(colors) : nullptr, F32P(b)));
  last_cam_indices_ = cam;
  last_ij_ = ij;
  return {{rays_o, rays_d, b}, image_tensors_.defined() ? colors : Tensor(), cam};
}

}  // namespace f2n
