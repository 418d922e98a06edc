// Shared mutable context handed to every plugin (mirrors src/Utils/GlobalDataPool.h:10-32).  The reference
// keeps a YAML::Node here; yaml-cpp is not available, so the config is a flat "a.b.c" -> string map produced
// by the Python-side composer of the reference's hydra configs (f2-nerf_amd/config.py).
#pragma once
#include "Common.h"

namespace f2n {

enum RunningMode { TRAIN, VALIDATE };

class Config {
 public:
  std::map<std::string, std::string> kv;
  bool Has(const std::string& k) const { return kv.count(k) > 0; }
  const std::string& Str(const std::string& k) const {
    auto it = kv.find(k);
    TORCH_CHECK(it != kv.end(), "missing config key: ", k);
    return it->second;
  }
  float Float(const std::string& k) const { return std::stof(Str(k)); }
  int Int(const std::string& k) const { return (int) std::lround(std::stod(Str(k))); }
  bool Bool(const std::string& k) const {
    const std::string& s = Str(k);
    return s == "true" || s == "True" || s == "1";
  }
  std::vector<int> IntList(const std::string& k) const {
    std::vector<int> out;
    std::stringstream ss(Str(k));
    std::string item;
    while (std::getline(ss, item, ',')) {
      if (!item.empty()) out.push_back((int) std::lround(std::stod(item)));
    }
    return out;
  }
};

class GlobalDataPool {
 public:
  Config config_;
  RunningMode mode_ = TRAIN;
  std::string base_exp_dir_;
  void *dataset_ = nullptr, *renderer_ = nullptr, *scene_field_ = nullptr, *shader_ = nullptr, *pts_sampler_ = nullptr;

  int n_volumes_ = 1;
  int iter_step_ = 0;
  float sampled_oct_per_ray_ = 16.f;
  float sampled_pts_per_ray_ = 512.f;
  float meaningful_sampled_pts_per_ray_ = 512.f;
  float learning_rate_ = 1.f;
  float distortion_weight_ = 0.f;
  float ray_march_fineness_ = 1.f;
  float near_ = 0.1f;
  float gradient_scaling_progress_ = 1.f;
  bool backward_nan_ = false;
};

}  // namespace f2n
