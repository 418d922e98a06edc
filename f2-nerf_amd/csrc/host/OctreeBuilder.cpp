// Construction of the perspective-warped octree from the training cameras.
//
// What it computes is what PersOctree's constructor computes in the reference (PtsSampler/PersSampler.cpp:16-66
// DistanceSummary / GetVisiCams, :70-118 constructor, :359-421 ConstructTreeNode, :423-612 PCA + ConstructTrans,
// :614-659 ConstructEdgePool); how it is computed is not: the reference recurses node by node and evaluates the camera
// visibility of every node with broadcast ATen ops; here the tree is grown LEVEL BY LEVEL -- one
// f2n_oct_visible_cams launch classifies every box of a depth against every camera -- and the nodes are numbered
// afterwards in the depth-first order the recursion would have produced, so the serialised tree (checkpoint layout)
// is the same.  Leaf warps are built on the device except for the 12x12 eigen-decomposition.
#include <algorithm>
#include <cstring>

#include "PersSampler.h"

namespace f2n {

namespace {

// PersSampler.cpp:16-25: robust "typical camera distance": exp of the mean log-distance of the closest quartile.
float DistanceSummary(const Tensor& dis_in) {
  Tensor dis = dis_in.to(torch::kCPU).to(torch::kFloat32).reshape({-1});
  if (dis.numel() <= 0) return 1e8f;
  Tensor log_dis = torch::log(dis);
  const float thres = torch::quantile(log_dis, 0.25).item<float>();
  Tensor mask = (log_dis < thres).to(torch::kFloat32);
  if (mask.sum().item<float>() < 1e-3f) return std::exp(log_dis.mean().item<float>());
  return std::exp(((log_dis * mask).sum() / mask.sum()).item<float>());
}

// Rodrigues rotation, row-major 3x3.
void AngleAxis(float angle, const float* axis, float* R) {
  const float x = axis[0], y = axis[1], z = axis[2];
  const float c = std::cos(angle), s = std::sin(angle), C = 1.f - c;
  R[0] = c + C * x * x; R[1] = C * x * y - s * z; R[2] = C * x * z + s * y;
  R[3] = C * x * y + s * z; R[4] = c + C * y * y; R[5] = C * y * z - s * x;
  R[6] = C * x * z - s * y; R[7] = C * y * z + s * x; R[8] = c + C * z * z;
}

}  // namespace

// One leaf's warp (PersSampler.cpp:438-612): six well-spread visible cameras (greedy farthest-point selection on the
// unit sphere around the leaf), re-aimed at the leaf centre and pulled in to the typical distance, give 12 projection
// rows (x and y image axes); the 3 x 12 mixing weights are the top principal axes of the projected coordinates of random
// points of the leaf, rescaled so that a unit step in warp space is on average a unit-Jacobian step in world space.
// rand_pts [n,3] (device), c2w_vis [m,3,4], intri0 [3,3], center [3]; first_cam = the one random draw of the algorithm.
TransInfo ConstructTrans(const Tensor& rand_pts, const Tensor& c2w_vis, const Tensor& intri0, const Tensor& center_in, int first_cam) {
  const auto dev = rand_pts.device();
  const int n_virt = N_PROS / 2;
  Tensor c2w = c2w_vis.to(torch::kCPU).to(torch::kFloat32).contiguous();
  Tensor center = center_in.to(torch::kCPU).to(torch::kFloat32).contiguous();
  const int n_cur = (int) c2w.size(0);
  TORCH_CHECK(n_cur > 0 && first_cam >= 0 && first_cam < n_cur, "ConstructTrans: bad camera set");
  Tensor cam_pos = c2w.index({Slc(), Slc(0, 3), 3}).contiguous();
  Tensor cam_axes = torch::linalg_inv(c2w.index({Slc(), Slc(0, 3), Slc(0, 3)})).contiguous();
  Tensor dis = torch::linalg_vector_norm(cam_pos - center.unsqueeze(0), 2, {-1});
  const float dis_summary = DistanceSummary(dis);
  Tensor normed = (cam_pos - center.unsqueeze(0)) / dis.unsqueeze(-1);
  Tensor dis_pairs = torch::linalg_vector_norm(normed.unsqueeze(0) - normed.unsqueeze(1), 2, {-1}).contiguous();
  const float* dp = dis_pairs.data_ptr<float>();
  // greedy farthest-point selection (:461-488)
  std::vector<int> good{first_cam};
  std::vector<char> mark(n_cur, 0);
  mark[first_cam] = 1;
  for (int cnt = 1; cnt < n_virt && cnt < n_cur; cnt++) {
    int candi = -1;
    float best = -1.f;
    for (int i = 0; i < n_cur; i++) {
      if (mark[i]) continue;
      float nearest = 1e8f;
      for (int j = 0; j < n_cur; j++)
        if (mark[j]) nearest = std::min(nearest, dp[i * n_cur + j]);
      if (nearest > best) {
        best = nearest;
        candi = i;
      }
    }
    mark[candi] = 1;
    good.push_back(candi);
  }
  for (int i = 0; (int) good.size() < n_virt; i++) good.push_back(good[i]);

  Tensor cam_scale = (dis / dis_summary).clamp(1.f, 1e9f);
  Tensor rel = (cam_pos - center.unsqueeze(0)) / dis.unsqueeze(-1) * dis.unsqueeze(-1).clamp(dis_summary, 1e9f);
  Tensor g = torch::from_blob(good.data(), {n_virt}, CpuI32()).to(torch::kInt64);
  Tensor good_rel = rel.index({g}).contiguous();
  Tensor good_cam_pos = good_rel + center.unsqueeze(0);
  Tensor good_axis = cam_axes.index({g}).clone().contiguous();
  Tensor good_scale = cam_scale.index({g}).contiguous();
  Tensor expect_z = (good_rel / torch::linalg_vector_norm(good_rel, 2, {-1}, true)).contiguous();
  Tensor rots = torch::zeros({n_virt, 3, 3}, CpuF32());
  for (int k = 0; k < n_virt; k++) {  // turn each camera frame so that its z axis looks along leaf -> camera
    const float* fz = good_axis.data_ptr<float>() + 9 * k + 6;
    const float* tz = expect_z.data_ptr<float>() + 3 * k;
    float cr[3] = {fz[1] * tz[2] - fz[2] * tz[1], fz[2] * tz[0] - fz[0] * tz[2], fz[0] * tz[1] - fz[1] * tz[0]};
    const float cos_v = fz[0] * tz[0] + fz[1] * tz[1] + fz[2] * tz[2];
    const float sin_v = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    float angle = std::asin(std::min(sin_v, 1.f));
    if (cos_v < 0.f) angle = float(M_PI) - angle;
    if (sin_v > 0.f)
      for (float& v : cr) v /= sin_v;
    AngleAxis(angle, cr, rots.data_ptr<float>() + 9 * k);
  }
  good_axis = torch::matmul(good_axis, rots.transpose(1, 2));
  const float focal = (intri0.index({0, 0}) / intri0.index({0, 2})).item<float>();
  Tensor scale = (good_scale * focal).unsqueeze(-1);
  Tensor x_axis = good_axis.index({Slc(), 0, Slc()}) * scale, y_axis = good_axis.index({Slc(), 1, Slc()}) * scale;
  Tensor z_axis = good_axis.index({Slc(), 2, Slc()});
  Tensor xs = torch::cat({x_axis, y_axis}, 0), zs = torch::cat({z_axis, z_axis}, 0);
  Tensor wp = torch::cat({good_cam_pos, good_cam_pos}, 0);
  Tensor frame = torch::zeros({N_PROS, 2, 4}, CpuF32());
  frame.index_put_({Slc(), 0, Slc(0, 3)}, xs);
  frame.index_put_({Slc(), 1, Slc(0, 3)}, zs);
  frame.index_put_({Slc(), 0, 3}, -(xs * wp).sum(-1));
  frame.index_put_({Slc(), 1, 3}, -(zs * wp).sum(-1));

  // PCA mixing weights over the leaf's random points (:568-597), on the device
  Tensor fd = frame.to(dev);
  Tensor pts = rand_pts.to(torch::kFloat32);
  Tensor A = fd.index({Slc(), Slc(), Slc(0, 3)}).reshape({N_PROS * 2, 3});                        // [24,3]
  Tensor tp = (torch::matmul(pts, A.transpose(0, 1)) + fd.index({Slc(), Slc(), 3}).reshape({1, N_PROS * 2}))
                  .reshape({-1, N_PROS, 2});                                                      // [n,12,2] = (x, z)
  Tensor tx = tp.index({Slc(), Slc(), 0}), tz = tp.index({Slc(), Slc(), 1});
  TORCH_CHECK(tz.max().item<float>() < 0.f, "ConstructTrans: a leaf point lies behind one of its cameras");
  Tensor dv_da = 1.f / tz, dv_db = tx / -tz.square();
  Tensor fx3 = fd.index({Slc(), 0, Slc(0, 3)}), fz3 = fd.index({Slc(), 1, Slc(0, 3)});             // [12,3]
  Tensor dv_dxyz = dv_da.unsqueeze(-1) * fx3.unsqueeze(0) + dv_db.unsqueeze(-1) * fz3.unsqueeze(0);  // [n,12,3]
  Tensor tv = tx / tz;                                                                             // [n,12]
  Tensor moved = tv - tv.mean(0, true);
  Tensor cov = (torch::matmul(moved.transpose(0, 1), moved) / float(tv.size(0))).to(torch::kCPU);
  auto eig = torch::linalg_eigh(cov);
  Tensor L = std::get<0>(eig).to(torch::kFloat32), Vall = std::get<1>(eig).to(torch::kFloat32);
  Tensor order = std::get<1>(torch::sort(L, 0, /*descending=*/true));
  Tensor V = Vall.transpose(0, 1).index({order}).index({Slc(0, 3)}).contiguous().to(dev);          // [3,12] top principal axes
  Tensor jac = torch::matmul(V.unsqueeze(0), dv_dxyz);                                             // [n,3,3]
  // batched 3x3 inverse by the adjugate (no solver library needed)
  auto J = [&](int r, int c) { return jac.index({Slc(), r, c}); };
  Tensor c00 = J(1, 1) * J(2, 2) - J(1, 2) * J(2, 1), c01 = J(0, 2) * J(2, 1) - J(0, 1) * J(2, 2), c02 = J(0, 1) * J(1, 2) - J(0, 2) * J(1, 1);
  Tensor c10 = J(1, 2) * J(2, 0) - J(1, 0) * J(2, 2), c11 = J(0, 0) * J(2, 2) - J(0, 2) * J(2, 0), c12 = J(0, 2) * J(1, 0) - J(0, 0) * J(1, 2);
  Tensor c20 = J(1, 0) * J(2, 1) - J(1, 1) * J(2, 0), c21 = J(0, 1) * J(2, 0) - J(0, 0) * J(2, 1), c22 = J(0, 0) * J(1, 1) - J(0, 1) * J(1, 0);
  Tensor det = J(0, 0) * c00 + J(0, 1) * c10 + J(0, 2) * c20;
  Tensor inv = torch::stack({torch::stack({c00, c01, c02}, -1), torch::stack({c10, c11, c12}, -1), torch::stack({c20, c21, c22}, -1)}, -2) /
               det.unsqueeze(-1).unsqueeze(-1);
  Tensor jac_w2i = torch::matmul(dv_dxyz, inv);                                                    // [n,12,3]
  Tensor jac_max = std::get<0>(jac_w2i.abs().max(1));                                              // [n,3]
  Tensor mean_step = (1.f / jac_max).mean(0);                                                      // [3]
  Tensor W = (V / mean_step.unsqueeze(-1)).to(torch::kCPU).contiguous();
  TORCH_CHECK(torch::isfinite(W).all().item<bool>() && torch::isfinite(frame).all().item<bool>(), "ConstructTrans: non-finite warp");

  TransInfo t;
  std::memset(&t, 0, sizeof(t));
  std::memcpy(t.w2xz, frame.contiguous().data_ptr<float>(), sizeof(t.w2xz));
  std::memcpy(t.weight, W.data_ptr<float>(), sizeof(t.weight));
  std::memcpy(t.center, center.data_ptr<float>(), sizeof(t.center));
  t.dis_summary = dis_summary;
  return t;
}

// Shared faces between valid leaves (PersSampler.cpp:614-659): for every pair, the face centres of the smaller (or
// equal, first) leaf that lie on the other leaf's surface.  O(n^2) over a few hundred leaves at construction time.
std::vector<EdgePool> ConstructEdgePool(const std::vector<TreeNode>& nodes) {
  std::vector<int> valid;
  for (int i = 0; i < (int) nodes.size(); i++)
    if (nodes[i].trans_idx >= 0) valid.push_back(i);
  std::vector<EdgePool> out;
  static const int face_axis[6] = {0, 0, 1, 1, 2, 2};
  static const float face_sign[6] = {1.f, -1.f, 1.f, -1.f, 1.f, -1.f};
  for (size_t ai = 0; ai < valid.size(); ai++)
    for (size_t bi = ai + 1; bi < valid.size(); bi++) {
      const int a = valid[ai], b = valid[bi];
      const bool a_small = !(nodes[a].side_len > nodes[b].side_len);
      const TreeNode& u = a_small ? nodes[a] : nodes[b];
      const TreeNode& v = a_small ? nodes[b] : nodes[a];
      const float len_u = u.side_len * .5f;
      for (int f = 0; f < 6; f++) {
        float p[3] = {u.center[0], u.center[1], u.center[2]};
        p[face_axis[f]] = p[face_axis[f]] + face_sign[f] * len_u;
        float mx = 0.f;
        for (int k = 0; k < 3; k++) mx = std::max(mx, std::fabs((p[k] - v.center[k]) / v.side_len * 2.f));
        if (!(mx < 1.f + 1e-4f)) continue;
        EdgePool e;
        std::memset(&e, 0, sizeof(e));
        e.t_idx_a = nodes[a].trans_idx;
        e.t_idx_b = nodes[b].trans_idx;
        for (int k = 0; k < 3; k++) e.center[k] = p[k];
        int d0 = -1, d1 = -1;
        for (int k = 0; k < 3; k++)
          if (k != face_axis[f]) (d0 < 0 ? d0 : d1) = k;
        e.dir_0[d0] = len_u;
        e.dir_1[d1] = len_u;
        out.push_back(e);
      }
    }
  return out;
}

// Level-synchronous construction.  c2w / w2c [C,3,4], intri [C,3,3], bounds [C,2]: the TRAINING cameras.
OctreeBuildResult BuildPersOctree(const Tensor& c2w_in, const Tensor& intri_in, const Tensor& bounds_in, int max_depth, float bbox_side_len,
                                  float split_dist_thres, int n_rand_pts) {
  Tensor c2w = c2w_in.to(torch::kCPU).to(torch::kFloat32).contiguous(), intri = intri_in.to(torch::kCPU).to(torch::kFloat32).contiguous();
  Tensor bounds = bounds_in.to(torch::kCPU).to(torch::kFloat32).contiguous();
  const int n_cams = (int) c2w.size(0);
  TORCH_CHECK(n_cams > 0 && c2w.dim() == 3 && c2w.size(1) == 3 && c2w.size(2) == 4, "c2w must be [C,3,4]");
  Tensor c2w_d = c2w.to(torch::kCUDA), bounds_d = bounds.to(torch::kCUDA);
  // GetVisiCams' pixel bundle (:33-47): 128 columns, rows to match the aspect ratio, pixel centres by linspace
  const float half_w = intri.index({0, 0, 2}).item<float>(), half_h = intri.index({0, 1, 2}).item<float>();
  const float fx = intri.index({0, 0, 0}).item<float>(), fy = intri.index({0, 1, 1}).item<float>();
  const int res_w = 128, res_h = (int) std::lround(float(res_w) / half_w * half_h);
  Tensor pix_i = torch::linspace(.5f, half_h * 2.f - .5f, res_h, CpuF32()).to(torch::kCUDA).contiguous();
  Tensor pix_j = torch::linspace(.5f, half_w * 2.f - .5f, res_w, CpuF32()).to(torch::kCUDA).contiguous();
  Tensor cam_pos = c2w.index({Slc(), Slc(0, 3), 3}).contiguous();

  struct Build {
    float center[3], side;
    int depth, parent;
    int childs[8];
    bool leaf = false, valid = false;
    std::vector<int> visi;
  };
  std::vector<Build> tree(1);
  tree[0].center[0] = tree[0].center[1] = tree[0].center[2] = 0.f;
  tree[0].side = bbox_side_len;
  tree[0].depth = 0;
  tree[0].parent = -1;
  for (int& c : tree[0].childs) c = -1;
  std::vector<int> frontier{0};
  while (!frontier.empty()) {
    // nodes beyond max_depth are invalid leaves without further tests (:368-372)
    std::vector<int> test;
    for (int u : frontier) {
      if (tree[u].depth > max_depth) tree[u].leaf = true;
      else test.push_back(u);
    }
    std::vector<int> next;
    if (!test.empty()) {
      Tensor boxes = torch::empty({(int64_t) test.size(), 4}, CpuF32());
      for (size_t k = 0; k < test.size(); k++) {
        float* b = boxes.data_ptr<float>() + 4 * k;
        b[0] = tree[test[k]].center[0]; b[1] = tree[test[k]].center[1]; b[2] = tree[test[k]].center[2]; b[3] = tree[test[k]].side;
      }
      Tensor boxes_d = boxes.to(torch::kCUDA);
      Tensor visible = torch::empty({(int64_t) test.size(), n_cams}, DevU8());
      F2N_CALL(f2n_oct_visible_cams(CurStream(), (int) test.size(), n_cams, F32P(boxes_d), F32P(c2w_d), F32P(bounds_d), fx, fy, half_w, half_h,
                                    res_h, res_w, F32P(pix_i), F32P(pix_j), visible.data_ptr<uint8_t>()));
      Tensor vis_cpu = visible.to(torch::kCPU);
      const uint8_t* vp = vis_cpu.data_ptr<uint8_t>();
      for (size_t k = 0; k < test.size(); k++) {
        Build& nd = tree[test[k]];
        for (int c = 0; c < n_cams; c++)
          if (vp[k * n_cams + c]) nd.visi.push_back(c);
        Tensor center = torch::from_blob(nd.center, {3}, CpuF32()).clone();
        float dsum = 1e8f;
        if (!nd.visi.empty()) {
          Tensor idx = torch::from_blob(nd.visi.data(), {(int64_t) nd.visi.size()}, CpuI32()).to(torch::kInt64);
          dsum = DistanceSummary(torch::linalg_vector_norm(cam_pos.index({idx}) - center.unsqueeze(0), 2, {-1}));
        }
        const bool enough = (int) nd.visi.size() >= N_PROS / 2;
        if (enough && dsum < nd.side * split_dist_thres) {  // cameras closer than the box can resolve: subdivide (:396-410)
          const int u = test[k];
          for (int st = 0; st < 8; st++) {
            Build ch;
            const float off[3] = {float((st >> 2) & 1) - .5f, float((st >> 1) & 1) - .5f, float(st & 1) - .5f};
            for (int a = 0; a < 3; a++) ch.center[a] = tree[u].center[a] + tree[u].side * .5f * off[a];
            ch.side = tree[u].side * .5f;
            ch.depth = tree[u].depth + 1;
            ch.parent = u;
            for (int& c : ch.childs) c = -1;
            tree[u].childs[st] = (int) tree.size();
            next.push_back((int) tree.size());
            tree.push_back(ch);
          }
        } else {
          nd.leaf = true;
          nd.valid = enough;
        }
      }
    }
    frontier.swap(next);
  }

  // number the nodes as the reference's recursion would (a node's subtree is complete before its next sibling is created)
  std::vector<int> order, new_id(tree.size(), -1), stack{0};
  while (!stack.empty()) {
    const int u = stack.back();
    stack.pop_back();
    new_id[u] = (int) order.size();
    order.push_back(u);
    for (int st = 7; st >= 0; st--)
      if (tree[u].childs[st] >= 0) stack.push_back(tree[u].childs[st]);
  }
  // ... except that the recursion numbers all eight children's slots in creation order: child st is created (numbered)
  // right before ITS subtree, which is exactly pre-order.
  OctreeBuildResult res;
  res.nodes.resize(tree.size());
  std::vector<TransInfo> trans;
  for (size_t k = 0; k < order.size(); k++) {
    const Build& b = tree[order[k]];
    TreeNode& nd = res.nodes[k];
    std::memset((void*) &nd, 0, sizeof(nd));
    for (int a = 0; a < 3; a++) nd.center[a] = b.center[a];
    nd.side_len = b.side;
    nd.parent = b.parent >= 0 ? new_id[b.parent] : -1;
    for (int st = 0; st < 8; st++) nd.childs[st] = b.childs[st] >= 0 ? new_id[b.childs[st]] : -1;
    nd.is_leaf_node = b.leaf;
    nd.trans_idx = -1;
    if (b.leaf && b.valid) {  // a warp per valid leaf, in pre-order like pers_trans_.push_back (:417-420)
      nd.trans_idx = (int) trans.size();
      Tensor center = torch::from_blob((void*) b.center, {3}, CpuF32()).clone();
      Tensor rand_pts = (torch::rand({n_rand_pts, 3}, DevF32()) - .5f) * b.side + center.to(torch::kCUDA).unsqueeze(0);
      Tensor idx = torch::from_blob((void*) b.visi.data(), {(int64_t) b.visi.size()}, CpuI32()).to(torch::kInt64);
      const int first = torch::randint((int64_t) b.visi.size(), {1}, CpuI32()).item<int>();
      trans.push_back(ConstructTrans(rand_pts, c2w.index({idx}), intri[0], center, first));
    }
  }
  res.trans = std::move(trans);
  res.edges = ConstructEdgePool(res.nodes);
  return res;
}

}  // namespace f2n
