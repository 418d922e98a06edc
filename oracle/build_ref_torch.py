#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Compiles the reference's own HOST code for perspective-warp construction --
``DistanceSummary``, ``GetVisiCams``, ``PCA`` and the body of ``PersOctree::ConstructTrans``
(PtsSampler/PersSampler.cpp:16-66, 423-612) -- against the libtorch that ships in this image's torch wheel, on the CPU, into
``oracle/_ref/libf2n_ref_torch.so``.  The sources are read in place from /root/reference and never copied into the repo;
the only edits are textual: ``torch::kCUDA`` -> ``torch::kCPU`` / ``CUDAFloat`` -> ``CPUFloat`` (no GPU here), the member
function becomes a free function, glog's CHECKs become counters.  tests/test_oracle_vs_ref.py uses the library to pin
``oracle/octree_construct.py::construct_trans`` / ``get_visi_cams`` / ``distance_summary`` -- the comparators of the device-side
octree / warp builder -- against the reference's code instead of against a reading of it.

What this cannot pin: LibTorch 1.13's CUDA kernels for linalg_inv / linalg_eigh / matmul (the reference runs this function
on the GPU); both sides here run the same ops of the same libtorch on the CPU, so the comparison is about the ALGORITHM
(camera selection, frame construction, PCA weights, step normalisation), not about LAPACK-vs-cuSOLVER ulps."""
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from build_ref import _read, extract_between, extract_function  # noqa: E402


def build(reference="/root/reference", keep_tu=False, verbose=True):
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(reference, "src")
    if not os.path.isdir(src):
        raise FileNotFoundError("reference sources not found at %s" % src)
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)
    tu = ["#include <torch/torch.h>", "#include <Eigen/Eigen>", "#include <cmath>", "#include <cstring>", "#include <vector>",
          "using Tensor = torch::Tensor;",
          "static int g_ref_check_failures = 0;",
          "#define CHECK(x) do { if (!(x)) g_ref_check_failures++; } while (false)",
          "#define CHECK_GT(a, b) CHECK((a) > (b))", "#define CHECK_GE(a, b) CHECK((a) >= (b))",
          "#define CHECK_LT(a, b) CHECK((a) < (b))", "#define CHECK_LE(a, b) CHECK((a) <= (b))",
          '#include "%s/Common.h"' % src,
          "#undef CUDAFloat", "#define CUDAFloat CPUFloat", "#undef CUDAInt", "#define CUDAInt CPUInt",
          "#undef PRINT_VAL", "#define PRINT_VAL(x) do { } while (false)", ""]
    h = _read(os.path.join(src, "PtsSampler/PersSampler.h"))
    tu += extract_between(h, r"^#define INIT_NODE_STAT", r"^#define TransWetType")
    tu += extract_between(h, r"^struct alignas\(32\) TransInfo", r"^};")
    tu += extract_between(h, r"^struct alignas\(32\) TreeNode", r"^};")
    cpp = _read(os.path.join(src, "PtsSampler/PersSampler.cpp"))
    host = []
    host += extract_function(cpp, r"^float DistanceSummary\(")
    host += [""]
    host += extract_function(cpp, r"^std::vector<int> GetVisiCams\(")
    host += [""]
    host += extract_function(cpp, r"^std::tuple<Tensor, Tensor> PCA\(")
    host += [""]
    body = extract_function(cpp, r"^TransInfo PersOctree::ConstructTrans\(")
    body[0] = body[0].replace("TransInfo PersOctree::ConstructTrans(", "static TransInfo ref_construct_trans_body(")
    host += body
    # PersOctree::ConstructTreeNode (:359-421) as a free function over file-scope copies of the members it uses
    host += ["", "#define ConstructTrans ref_construct_trans_body",
             "static std::vector<TreeNode> tree_nodes_;", "static std::vector<TransInfo> pers_trans_;",
             "static Tensor c2w_, intri_, bound_;", "static int max_depth_ = 0;", "static float split_dist_thres_ = 0.f;",
             "static void ConstructTreeNode(int u, int depth, Wec3f center, float side_len);"]
    node = extract_function(cpp, r"^void PersOctree::ConstructTreeNode\(")
    node[0] = node[0].replace("void PersOctree::ConstructTreeNode(", "static void ConstructTreeNode(")
    host += node
    # Dataset::NormalizeScene (Dataset/Dataset.cpp:127-146) over file-scope copies of the members it touches
    ds = _read(os.path.join(src, "Dataset/Dataset.cpp"))
    norm = extract_function(ds, r"^void Dataset::NormalizeScene\(\)")
    norm[0] = norm[0].replace("void Dataset::NormalizeScene()", "static void NormalizeScene()")
    host += ["", "namespace Utils { static void TensorExportPCD(const std::string&, const Tensor&) {} }",
             "struct RefCfgNode { RefCfgNode operator[](const char*) const { return RefCfgNode(); } };",
             "struct RefDsGdp { RefCfgNode config_; std::string base_exp_dir_; };",
             "static RefDsGdp ref_ds_gdp; static RefDsGdp* global_data_pool_ = &ref_ds_gdp;",
             "static Tensor poses_, w2c_, bounds_, center_; static float radius_ = 0.f; static int n_images_ = 0;"]
    host += norm
    # PoseInterpolate (Utils/CameraUtils.cpp:11-44): Eigen quaternion slerp between two camera poses
    cu = _read(os.path.join(src, "Utils/CameraUtils.cpp"))
    host += [""] + extract_function(cu, r"^Tensor PoseInterpolate\(")
    text = "\n".join(host).replace("torch::kCUDA", "torch::kCPU")
    tu += text.split("\n")
    tu += ["", r'''
extern "C" {
int ref_torch_check_failures() { return g_ref_check_failures; }
float ref_distance_summary(int n, const float* dis) {
  return DistanceSummary(torch::from_blob(const_cast<float*>(dis), {n}, CPUFloat).clone());
}
// returns the number of visible cameras; out_idx [n_cams]
int ref_get_visi_cams(float side_len, const float* center3, int n_cams, const float* c2w34, const float* intri33, const float* bound2,
                      int* out_idx) {
  Tensor c = torch::from_blob(const_cast<float*>(center3), {3}, CPUFloat).clone();
  Tensor c2w = torch::from_blob(const_cast<float*>(c2w34), {n_cams, 3, 4}, CPUFloat).clone();
  Tensor intri = torch::from_blob(const_cast<float*>(intri33), {n_cams, 3, 3}, CPUFloat).clone();
  Tensor bound = torch::from_blob(const_cast<float*>(bound2), {n_cams, 2}, CPUFloat).clone();
  std::vector<int> v = GetVisiCams(side_len, c, c2w, intri, bound);
  for (size_t i = 0; i < v.size(); i++) out_idx[i] = v[i];
  return (int) v.size();
}
// out: one TransInfo (544 bytes)
// PersOctree::PersOctree up to ConstructEdgePool (:70-82): returns the node count; counts = {n_nodes, n_trans}
int ref_build_octree(int max_depth, float bbox_side_len, float split_dist_thres, int n_cams, const float* c2w34, const float* intri33,
                     const float* bound2, void* out_nodes, int cap_nodes, void* out_trans, int cap_trans, int* counts) {
  max_depth_ = max_depth;
  split_dist_thres_ = split_dist_thres;
  c2w_ = torch::from_blob(const_cast<float*>(c2w34), {n_cams, 3, 4}, CPUFloat).clone();
  intri_ = torch::from_blob(const_cast<float*>(intri33), {n_cams, 3, 3}, CPUFloat).clone();
  bound_ = torch::from_blob(const_cast<float*>(bound2), {n_cams, 2}, CPUFloat).clone();
  tree_nodes_.clear();
  pers_trans_.clear();
  TreeNode root;
  root.parent = -1;
  tree_nodes_.push_back(root);
  ConstructTreeNode(0, 0, Wec3f::Zero(), bbox_side_len);
  counts[0] = (int) tree_nodes_.size();
  counts[1] = (int) pers_trans_.size();
  static_assert(sizeof(TreeNode) == 64, "TreeNode layout");
  if ((int) tree_nodes_.size() > cap_nodes || (int) pers_trans_.size() > cap_trans) return -1;
  std::memcpy(out_nodes, tree_nodes_.data(), tree_nodes_.size() * sizeof(TreeNode));
  std::memcpy(out_trans, pers_trans_.data(), pers_trans_.size() * sizeof(TransInfo));
  return (int) tree_nodes_.size();
}
// Dataset::NormalizeScene + the bounds relaxation of the constructor (Dataset.cpp:73-76): poses [n,3,4] and bounds [n,2] in place;
// w2c [n,3,4], center [3], radius out
void ref_normalize_scene(int n, float* poses34, float* bounds2, float f0, float f1, float* w2c34, float* center3, float* radius) {
  poses_ = torch::from_blob(poses34, {n, 3, 4}, CPUFloat).clone();
  bounds_ = torch::from_blob(bounds2, {n, 2}, CPUFloat).clone();
  n_images_ = n;
  NormalizeScene();
  bounds_ = torch::stack({bounds_.index({"...", 0}) * f0, bounds_.index({"...", 1}) * f1}, -1).contiguous();  // :73-75, as spelled there
  bounds_.clamp_(1e-2f, 1e9f);
  std::memcpy(poses34, poses_.contiguous().data_ptr(), sizeof(float) * n * 12);
  std::memcpy(bounds2, bounds_.contiguous().data_ptr(), sizeof(float) * n * 2);
  std::memcpy(w2c34, w2c_.contiguous().data_ptr(), sizeof(float) * n * 12);
  std::memcpy(center3, center_.contiguous().data_ptr(), sizeof(float) * 3);
  *radius = radius_;
}
void ref_pose_interpolate(const float* a34, const float* b34, float alpha, float* out34) {
  Tensor a = torch::from_blob(const_cast<float*>(a34), {3, 4}, CPUFloat).clone();
  Tensor b = torch::from_blob(const_cast<float*>(b34), {3, 4}, CPUFloat).clone();
  Tensor r = PoseInterpolate(a, b, alpha).contiguous();
  std::memcpy(out34, r.data_ptr(), sizeof(float) * 12);
}
void ref_construct_trans(int n_pts, const float* rand_pts, int n_cams, const float* c2w34, const float* intri33, const float* center3,
                         void* out) {
  Tensor pts = torch::from_blob(const_cast<float*>(rand_pts), {n_pts, 3}, CPUFloat).clone();
  Tensor c2w = torch::from_blob(const_cast<float*>(c2w34), {n_cams, 3, 4}, CPUFloat).clone();
  Tensor intri = torch::from_blob(const_cast<float*>(intri33), {3, 3}, CPUFloat).clone();
  Tensor c = torch::from_blob(const_cast<float*>(center3), {3}, CPUFloat).clone();
  TransInfo t = ref_construct_trans_body(pts, c2w, intri, c);
  static_assert(sizeof(TransInfo) == 544, "TransInfo layout");
  std::memcpy(out, &t, sizeof(TransInfo));
}
}
''']
    tmp = tempfile.mkdtemp(prefix="f2n_ref_torch_")
    tu_path = os.path.join(tmp, "ref_torch_tu.cpp")
    with open(tu_path, "w") as f:
        f.write("\n".join(tu))
    so = os.path.join(out_dir, "libf2n_ref_torch.so")
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = (["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
           + ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"],
                                                        "-I", os.path.join(reference, "External/eigen-3.4.0"),
                                                        tu_path, "-o", so, "-L" + libdir, "-Wl,-rpath," + libdir, "-lc10", "-ltorch_cpu",
                                                        "-ltorch"])
    if verbose:
        print(" ".join(cmd[:6]), "...", so)
    try:
        subprocess.check_call(cmd)
    finally:
        if keep_tu:
            print("kept TU at", tu_path)
        else:
            shutil.rmtree(tmp, ignore_errors=True)
    return so


if __name__ == "__main__":
    print(build(keep_tu="--keep-tu" in sys.argv))
