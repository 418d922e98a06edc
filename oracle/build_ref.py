#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- builds oracle/_ref/libf2n_ref.so from the reference's own kernels.

The reference (Totoro97/f2-nerf) has no CPU path and cannot be built here (needs nvcc, CUDA LibTorch,
tiny-cuda-nn, yaml-cpp).  What CAN be done cheaply: its first-party ``__global__`` kernels are plain
C++ + Eigen.  This recipe reads them IN PLACE from ``/root/reference/src`` (nothing is copied into the
repo; the generated translation unit lives in a temp dir and is deleted), prepends the CUDA-emulation
shim ``oracle/ref_shim/cuda_emul.h`` (serial thread loops, Eigen::half for __half), appends the
``extern "C"`` launch drivers in ``oracle/ref_driver.inc`` and compiles the lot with g++ into
``oracle/_ref/libf2n_ref.so`` (git-ignored, but it travels to the GPU box with the snapshot).

It pins the hand-written restatement (oracle/f2n_oracle.c) and generates tests/golden/*.  It is never
imported by the product.  tiny-cuda-nn (the MLP) is NOT in the reference tree, so the MLP stays
"parity unpinned" (see DESIGN.md).

Usage: python oracle/build_ref.py [--reference /root/reference] [--keep-tu]
"""
import argparse
import os
import re
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def _read(path):
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        return f.read().split("\n")


def extract_function(lines, start_pat, with_template=True):
    """Return the source lines of one top-level function: from the line matching start_pat (plus a
    directly preceding ``template`` line) to the first following line that is exactly ``}``."""
    rx = re.compile(start_pat)
    for i, ln in enumerate(lines):
        if rx.search(ln):
            s = i
            if with_template and i > 0 and lines[i - 1].lstrip().startswith("template"):
                s = i - 1
            for j in range(i, len(lines)):
                if lines[j].rstrip() == "}":
                    return lines[s:j + 1]
            raise RuntimeError("no closing brace for " + start_pat)
    raise RuntimeError("pattern not found: " + start_pat)


def extract_between(lines, start_pat, end_pat, include_end=True):
    rs, re_ = re.compile(start_pat), re.compile(end_pat)
    for i, ln in enumerate(lines):
        if rs.search(ln):
            for j in range(i + 1, len(lines)):
                if re_.search(lines[j]):
                    return lines[i:(j + 1 if include_end else j)]
            raise RuntimeError("end pattern not found: " + end_pat)
    raise RuntimeError("start pattern not found: " + start_pat)


def build(reference="/root/reference", keep_tu=False, verbose=True):
    src = os.path.join(reference, "src")
    if not os.path.isdir(src):
        raise FileNotFoundError("reference sources not found at %s" % src)
    out_dir = os.path.join(HERE, "_ref")
    os.makedirs(out_dir, exist_ok=True)

    tu = []
    add = tu.extend
    add(['#include "cuda_emul.h"', '#include "%s/Common.h"' % src, ""])

    # ---- sampler: types (PersSampler.h) and kernels (PersSampler.cu) -----------------------------
    h = _read(os.path.join(src, "PtsSampler/PersSampler.h"))
    add(extract_between(h, r"^#define INIT_NODE_STAT", r"^#define TransWetType"))
    add(extract_between(h, r"^struct alignas\(32\) TransInfo", r"^};"))
    add(extract_between(h, r"^struct alignas\(32\) TreeNode", r"^};"))
    add(extract_between(h, r"^struct alignas\(32\) EdgePool", r"^};"))
    cu = _read(os.path.join(src, "PtsSampler/PersSampler.cu"))
    add(extract_between(cu, r"^#define MAX_STACK_SIZE", r"^#define REL_ALPHA_THRES"))
    for pat in [r"^inline __device__ void GetIntersection",
                r"^__global__ void FindRayOctreeIntersectionKernel",
                r"^void __device__ QueryFrameTransform\(",
                r"^void __device__ QueryFrameTransformJac\(",
                r"^__global__ void RayMarchKernel",
                r"^__global__ void GetEdgeSamplesKernel",
                r"^__global__ void MarkVistNodeKernel",
                r"^__global__ void MarkInvalidNodes",
                r"^__device__ int CheckVisible",
                r"^__global__ void MarkInvisibleNodesKernel"]:
        add(extract_function(cu, pat))
        add([""])
    # child search-order table (host code inside PersOctree::PersOctree, PersSampler.cpp:106-117)
    cpp = _read(os.path.join(src, "PtsSampler/PersSampler.cpp"))
    body = extract_between(cpp, r"^\s*std::vector<int> search_order;", r"^\s*node_search_order_ = ",
                           include_end=False)
    add(["static std::vector<int> ref_build_search_order() {"] + body + ["  return search_order;", "}", ""])

    # ---- field: Hash3DAnchored ------------------------------------------------------------------
    h = _read(os.path.join(src, "Field/Hash3DAnchored.h"))
    add(extract_between(h, r"^#define N_CHANNELS", r"^#define RES_BASE_POW_2"))
    cu = _read(os.path.join(src, "Field/Hash3DAnchored.cu"))
    add(extract_function(cu, r"^__global__ void Hash3DAnchoredForwardKernel"))
    add(extract_function(cu, r"^__global__ void Hash3DAnchoredBackwardKernel"))

    # ---- shader: SH encoding ---------------------------------------------------------------------
    cu = _read(os.path.join(src, "Shader/SHShader.cu"))
    add(extract_function(cu, r"^__global__ void SHKenerl"))

    # ---- renderer / custom ops ------------------------------------------------------------------
    cu = _read(os.path.join(src, "Renderer/Renderer.cu"))
    add(extract_function(cu, r"^__global__ void CountValidPts"))
    cu = _read(os.path.join(src, "Utils/CustomOps/FlexOps.cu"))
    for k in ["FlexSumForwardKernel", "FlexSumBackwardKernel", "FlexSumVecForwardKernel",
              "FlexSumVecBackwardKernel", "FlexAccumulateSumForwardKernel",
              "FlexAccumulateSumBackwardKernel"]:
        add(extract_function(cu, r"^__global__ void %s" % k))
    cu = _read(os.path.join(src, "Utils/CustomOps/CustomOps.cu"))
    add([ln for ln in cu if ln.startswith("#define SCALE")][:1])
    for k in ["WeightVarLossForwardKernel", "WeightVarLossBackwardKernel", "GradientScalingBackwardKernel"]:
        add(extract_function(cu, r"^__global__ void %s" % k))
    cu = _read(os.path.join(src, "Utils/CustomOps/Scatter.cu"))
    for k in ["ScatterAddFuncForward", "ScatterAddFuncBackwardBlock", "ScatterIdxKernal"]:
        add(extract_function(cu, r"^__global__ void %s" % k))

    # ---- dataset ray generation (Dataset.cu) ----------------------------------------------------
    cu = _read(os.path.join(src, "Dataset/Dataset.cu"))
    add(extract_function(cu, r"^__device__ __host__ inline void apply_camera_distortion"))
    add(extract_function(cu, r"^__device__ __host__ inline void iterative_camera_undistortion"))
    add(extract_function(cu, r"^__global__ void Img2WorldRayKernel"))

    # ---- octree maintenance + edge pool: host code of PersOctree (PersSampler.cpp), plain C++ over std::vector<TreeNode> --
    # ProcOctree: the statements between its tensor -> vector prologue and its vector -> tensor epilogue (:138-319), pasted
    # into a function whose parameters carry the names the body uses.  ConstructEdgePool: its whole body (:615-659).
    body = extract_between(cpp, r"^\s*// First, compact tree nodes;", r"^\s*CHECK_EQ\(new_nodes.size\(\), new_weight_stats.size\(\)\);",
                           include_end=False)
    add(["#undef PRINT_VAL", "#define PRINT_VAL(x) do { } while (false)", "#ifndef CHECK",
         "#define CHECK(x) do { if (!(x)) ref_check_failed(#x); } while (false)", "#endif",
         "static int g_ref_check_failures = 0;", "static void ref_check_failed(const char*) { g_ref_check_failures++; }",
         "static void ref_proc_octree_body(std::vector<TreeNode>& tree_nodes_before, const int* weight_stats_before,",
         "                                 const int* alpha_stats_before, const std::vector<int>& visit_cnt, bool compact,",
         "                                 bool subdivide, bool brute_force, std::vector<TreeNode>& out_nodes_v,",
         "                                 std::vector<int>& out_w_v, std::vector<int>& out_a_v) {",
         "  int n_nodes_before = tree_nodes_before.size();"] + body +
        ["  out_nodes_v = std::move(new_nodes);", "  out_w_v = std::move(new_weight_stats);", "  out_a_v = std::move(new_alpha_stats);", "}", ""])
    body = extract_between(cpp, r"^void PersOctree::ConstructEdgePool\(\) \{", r"^\}", include_end=False)[2:]  # drop signature + ScopeWatch
    add(["static void ref_construct_edge_pool_body(const std::vector<TreeNode>& tree_nodes_, std::vector<EdgePool>& edge_pool_) {"]
        + body + ["}", ""])

    # ---- training schedules: the body of ExpRunner::UpdateAdaParams (ExpRunner.cpp:222-253) and the variance-loss ramp of
    # ExpRunner::Train (:108-114), pasted into a function whose locals carry the member names and types of ExpRunner.h:34-45 ----
    ex = _read(os.path.join(src, "ExpRunner.cpp"))
    ada = extract_between(ex, r"^void ExpRunner::UpdateAdaParams\(\) \{", r"^\}", include_end=False)[1:]
    ramp = extract_between(ex, r"^\s*float var_loss_weight = 0.f;", r"^\s*Tensor loss = color_loss", include_end=False)
    add(["struct RefGdp { float ray_march_fineness_ = 0.f, gradient_scaling_progress_ = 0.f; };",
         "struct RefOptGroup { float* lr; RefOptGroup& options() { return *this; } void set_lr(float v) { *lr = v; } };",
         "struct RefOpt { std::vector<RefOptGroup> g; std::vector<RefOptGroup>& param_groups() { return g; } };",
         "static void ref_schedules_body(unsigned iter_step_, unsigned end_iter_, float ray_march_init_fineness_,",
         "                               int ray_march_fineness_decay_end_iter_, int var_loss_start_, int var_loss_end_,",
         "                               int gradient_scaling_start_, int gradient_scaling_end_, float learning_rate_,",
         "                               float learning_rate_alpha_, float learning_rate_warm_up_end_iter_, float var_loss_weight_,",
         "                               float* out4) {",
         "  RefGdp gdp_obj; RefGdp* global_data_pool_ = &gdp_obj;",
         "  float lr_seen = 0.f; RefOpt opt_obj; opt_obj.g.push_back(RefOptGroup{&lr_seen}); RefOpt* optimizer_ = &opt_obj;"]
        + ada + ramp +
        ["  out4[0] = gdp_obj.ray_march_fineness_; out4[1] = lr_seen; out4[2] = gdp_obj.gradient_scaling_progress_; out4[3] = var_loss_weight;",
         "}", ""])

    add(["", '#include "%s"' % os.path.join(HERE, "ref_driver.inc"), ""])

    tmp = tempfile.mkdtemp(prefix="f2n_ref_")
    tu_path = os.path.join(tmp, "ref_tu.cpp")
    with open(tu_path, "w") as f:
        f.write("\n".join(tu))
    so = os.path.join(out_dir, "libf2n_ref.so")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Wno-invalid-offsetof", "-Wno-deprecated-declarations",
           "-I", os.path.join(HERE, "ref_shim"),
           "-I", os.path.join(reference, "External/eigen-3.4.0"),
           tu_path, "-o", so]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        if keep_tu:
            print("kept TU at", tu_path)
        else:
            shutil.rmtree(tmp, ignore_errors=True)
    return so


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--keep-tu", action="store_true")
    a = ap.parse_args()
    print(build(a.reference, a.keep_tu))
