// PersOctree::ProcOctree on gfx950 (PtsSampler/PersSampler.cpp:120-330): pruning of dead leaves, path compression,
// renumbering and subdivision of the occupancy octree WITHOUT the reference's device -> host -> device round trip of the
// node array (which stalls every data-parallel replica at each milestone / compaction).
//
// The reference code is a sequence of index-ordered loops over std::vector<TreeNode>; each has an order-free
// characterisation, which is what the kernels compute (equality with the sequential result is checked bit for bit in
// tests/test_gpu_e2e.py::test_proc_octree_matches_restatement, whose comparator is pinned to the reference's own code
// compiled for the CPU, tests/test_oracle_vs_ref.py):
//   compact loop (:139-178)   a node survives iff its subtree holds a leaf with trans_idx >= 0: valid leaves mark their
//                             ancestor chain; dead children are unhooked, childless interior nodes become (dead) leaves;
//   path compression (:181-215) every non-root interior node with exactly one child is spliced out; a node's new parent is
//                             its nearest ancestor that is not spliced, a child slot points at the first non-spliced node
//                             down the single-child chain below it (child counts do not change while splicing, and parents
//                             precede children in index order, so the sequential loop produces exactly this);
//   renumbering (:217-252)    exclusive prefix sum over the kept flags (f2n_segment_scan) + gather;
//   subdivision (:255-318)    the depth-first renumbering is a preorder position: 1 + the sizes of the elder siblings'
//                             subtrees, summed along the ancestor chain, where a subdivided leaf counts 9 nodes.
// Trees have at most a few 1e5 nodes and <= 24 levels; every kernel is one thread per node with an ancestor walk.
#include "f2n_dev.h"

#define F2N_INIT_NODE_STAT 1000  // PersSampler.h:10

// alive[u] = 1 for every node whose subtree contains a valid leaf (benign races: all writers store 1).
__global__ void oct_mark_alive_kernel(int n, const F2nTreeNode* __restrict__ nodes, int32_t* __restrict__ alive) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const F2nTreeNode& nd = nodes[u];
  if (!nd.is_leaf_node || nd.trans_idx < 0) return;
  int v = u;
  while (v >= 0 && alive[v] == 0) {  // (an already marked ancestor has marked the rest of the chain, or is doing so)
    alive[v] = 1;
    v = nodes[v].parent;
  }
}

// Applies the compact loop's fixed point to a working copy: childs of dead nodes unhooked, dead interior nodes (never the
// root) flagged as leaves, and the number of children left per node.
__global__ void oct_prune_kernel(int n, const int32_t* __restrict__ alive, F2nTreeNode* __restrict__ work, int32_t* __restrict__ n_child) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  F2nTreeNode& nd = work[u];
  int cnt = 0;
#pragma unroll
  for (int st = 0; st < 8; st++) {
    const int c = nd.childs[st];
    if (c >= 0) {
      // only LEAVES that are dead are unhooked by :141-151, interior nodes once they have turned into leaves (:153-172):
      // in the fixed point that is every child whose subtree holds no valid leaf
      if (alive[c] == 0) nd.childs[st] = -1;
      else cnt++;
    }
  }
  n_child[u] = cnt;
  if (u >= 1 && cnt == 0 && !nd.is_leaf_node) nd.is_leaf_node = 1;  // (its trans_idx is < 0: an interior node never carries a warp)
}

// Path compression.  spliced(v): interior, not the root, exactly one child.  Reads the pruned copy `work`, writes `out`.
__device__ __forceinline__ bool f2n_oct_spliced(const F2nTreeNode* __restrict__ work, const int32_t* __restrict__ n_child, int v) {
  return !work[v].is_leaf_node && work[v].parent >= 0 && n_child[v] == 1;
}

__global__ void oct_compress_kernel(int n, const F2nTreeNode* __restrict__ work, const int32_t* __restrict__ n_child,
                                    F2nTreeNode* __restrict__ out, int32_t* __restrict__ keep) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  F2nTreeNode nd = work[u];
  const bool dead_leaf = nd.is_leaf_node && nd.trans_idx < 0;
  if (!dead_leaf) {
    if (f2n_oct_spliced(work, n_child, u)) {  // :209-210 "the flag to remove it"
      nd.trans_idx = -1;
      nd.is_leaf_node = 1;
    } else {
      int v = nd.parent;  // nearest ancestor that stays
      while (v >= 0 && f2n_oct_spliced(work, n_child, v)) v = work[v].parent;
      nd.parent = v;
      if (!nd.is_leaf_node) {
#pragma unroll
        for (int st = 0; st < 8; st++) {
          int c = nd.childs[st];
          while (c >= 0 && f2n_oct_spliced(work, n_child, c)) {  // down the single-child chain
            int only = -1;
#pragma unroll
            for (int k = 0; k < 8; k++)
              if (work[c].childs[k] >= 0) only = work[c].childs[k];
            c = only;
          }
          nd.childs[st] = c;
        }
      }
    }
  }
  out[u] = nd;
  keep[u] = (!nd.is_leaf_node || nd.trans_idx >= 0) ? 1 : 0;  // :219-224
}

// Renumbering: new_pos[2u] = exclusive prefix of keep (f2n_segment_scan layout [n,2]).
__global__ void oct_gather_kept_kernel(int n, const F2nTreeNode* __restrict__ src, const int32_t* __restrict__ keep,
                                       const int32_t* __restrict__ new_pos, const int32_t* __restrict__ w_stats,
                                       const int32_t* __restrict__ a_stats, const int32_t* __restrict__ visit,
                                       F2nTreeNode* __restrict__ dst, int32_t* __restrict__ dst_w, int32_t* __restrict__ dst_a,
                                       int32_t* __restrict__ dst_visit) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || keep[u] == 0) return;
  F2nTreeNode nd = src[u];
  if (nd.parent >= 0) nd.parent = new_pos[2 * nd.parent];
#pragma unroll
  for (int st = 0; st < 8; st++)
    if (nd.childs[st] >= 0) nd.childs[st] = new_pos[2 * nd.childs[st]];
  const int k = new_pos[2 * u];
  dst[k] = nd;
  dst_w[k] = w_stats[u];
  dst_a[k] = a_stats[u];
  dst_visit[k] = visit[u];
}

// Subdivision, pass 1: subtree sizes in the new numbering (a leaf that splits counts 9 nodes), bottom-up one depth at a
// time -- plain loads and stores (ancestor-chain atomics would pile ~1e5 same-address atomics onto the root).
__global__ void oct_depth_kernel(int n, const F2nTreeNode* __restrict__ nodes, int32_t* __restrict__ depth) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  int d = 0;
  for (int v = nodes[u].parent; v >= 0; v = nodes[v].parent) d++;
  depth[u] = d;
}
__global__ void oct_size_level_kernel(int n, int level, const F2nTreeNode* __restrict__ nodes, const int32_t* __restrict__ depth,
                                      const int32_t* __restrict__ visit, int brute_force, int32_t* __restrict__ size) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n || depth[u] != level) return;
  const F2nTreeNode& nd = nodes[u];
  int sz = 1;
  if (nd.is_leaf_node) {
    if (brute_force || visit[u] > 4) sz = 9;  // :279
  } else {
#pragma unroll
    for (int st = 0; st < 8; st++)
      if (nd.childs[st] >= 0) sz += size[nd.childs[st]];  // written by the previous (deeper) launch
  }
  size[u] = sz;
}

// Subdivision, pass 2: preorder position of every node, and the nodes themselves in the new numbering.
__global__ void oct_subdivide_emit_kernel(int n, const F2nTreeNode* __restrict__ nodes, const int32_t* __restrict__ visit, int brute_force,
                                          const int32_t* __restrict__ size, const int32_t* __restrict__ w_stats,
                                          const int32_t* __restrict__ a_stats, int32_t* __restrict__ new_idx,
                                          F2nTreeNode* __restrict__ dst, int32_t* __restrict__ dst_w, int32_t* __restrict__ dst_a) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  auto position = [&](int x) {
    int pos = 0;
    int c = x;
    for (int p = nodes[x].parent; p >= 0; c = p, p = nodes[p].parent) {
      pos += 1;  // the parent itself precedes its subtree
#pragma unroll
      for (int st = 0; st < 8; st++) {
        const int s = nodes[p].childs[st];
        if (s == c) break;
        if (s >= 0) pos += size[s];
      }
    }
    return pos;
  };
  const int me = position(u);
  new_idx[u] = me;
  F2nTreeNode nd = nodes[u];
  const int old_parent = nd.parent;
  nd.parent = old_parent >= 0 ? position(old_parent) : -1;
  const bool split = nd.is_leaf_node && (brute_force || visit[u] > 4);
  if (!nd.is_leaf_node) {
    int next = me + 1;  // children follow their parent in slot order (:305-311)
#pragma unroll
    for (int st = 0; st < 8; st++) {
      const int s = nd.childs[st];
      if (s >= 0) {
        nd.childs[st] = next;
        next += size[s];
      }
    }
    dst[me] = nd;
    dst_w[me] = w_stats[u];
    dst_a[me] = a_stats[u];
  } else if (!split) {
    dst[me] = nd;
    dst_w[me] = w_stats[u];
    dst_a[me] = a_stats[u];
  } else {  // :280-303
    F2nTreeNode pr = nd;
#pragma unroll
    for (int st = 0; st < 8; st++) {
      const float off[3] = {float((st >> 2) & 1) - .5f, float((st >> 1) & 1) - .5f, float(st & 1) - .5f};
      F2nTreeNode ch;
      for (int k = 0; k < 3; k++) ch.center[k] = nd.center[k] + nd.side_len * .5f * off[k];
      ch.side_len = nd.side_len * .5f;
      ch.parent = me;
      for (int k = 0; k < 8; k++) ch.childs[k] = -1;
      ch.is_leaf_node = 1;
      ch.pad0[0] = ch.pad0[1] = ch.pad0[2] = 0;
      ch.trans_idx = nd.trans_idx;
      ch.pad1[0] = ch.pad1[1] = ch.pad1[2] = ch.pad1[3] = 0;
      dst[me + 1 + st] = ch;
      dst_w[me + 1 + st] = w_stats[u];
      dst_a[me + 1 + st] = a_stats[u];
      pr.childs[st] = me + 1 + st;
    }
    pr.is_leaf_node = 0;
    pr.trans_idx = -1;
    dst[me] = pr;
    dst_w[me] = F2N_INIT_NODE_STAT;
    dst_a[me] = F2N_INIT_NODE_STAT;
  }
}

extern "C" {

int f2n_oct_prune_compress(void* stream, int n_nodes, const void* tree_nodes, void* work_nodes, void* out_nodes, int32_t* alive,
                           int32_t* n_child, int32_t* keep) {
  if (n_nodes < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipStream_t st = (hipStream_t) stream;
  const dim3 grid(f2n_div_up(n_nodes, 256)), block(256);
  if (hipMemsetAsync(alive, 0, sizeof(int32_t) * (size_t) n_nodes, st) != hipSuccess) return f2n_launch_status();
  if (hipMemcpyAsync(work_nodes, tree_nodes, sizeof(F2nTreeNode) * (size_t) n_nodes, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return f2n_launch_status();
  hipLaunchKernelGGL(oct_mark_alive_kernel, grid, block, 0, st, n_nodes, (const F2nTreeNode*) tree_nodes, alive);
  hipLaunchKernelGGL(oct_prune_kernel, grid, block, 0, st, n_nodes, alive, (F2nTreeNode*) work_nodes, n_child);
  hipLaunchKernelGGL(oct_compress_kernel, grid, block, 0, st, n_nodes, (const F2nTreeNode*) work_nodes, n_child,
                     (F2nTreeNode*) out_nodes, keep);
  return f2n_launch_status();
}

int f2n_oct_gather_kept(void* stream, int n_nodes, const void* nodes, const int32_t* keep, const int32_t* new_pos,
                        const int32_t* w_stats, const int32_t* a_stats, const int32_t* visit_cnt, void* dst_nodes, int32_t* dst_w,
                        int32_t* dst_a, int32_t* dst_visit) {
  if (n_nodes < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_gather_kept_kernel, dim3(f2n_div_up(n_nodes, 256)), dim3(256), 0, (hipStream_t) stream, n_nodes,
                     (const F2nTreeNode*) nodes, keep, new_pos, w_stats, a_stats, visit_cnt, (F2nTreeNode*) dst_nodes, dst_w, dst_a,
                     dst_visit);
  return f2n_launch_status();
}

#define F2N_OCT_MAX_DEPTH 40  // the reference's own DFS walks at most 24 levels (PersSampler.cu:7)
int f2n_oct_subtree_sizes(void* stream, int n_nodes, const void* nodes, const int32_t* visit_cnt, int brute_force, int32_t* depth,
                          int32_t* size) {
  if (n_nodes < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipStream_t st = (hipStream_t) stream;
  const dim3 grid(f2n_div_up(n_nodes, 256)), block(256);
  hipLaunchKernelGGL(oct_depth_kernel, grid, block, 0, st, n_nodes, (const F2nTreeNode*) nodes, depth);
  for (int level = F2N_OCT_MAX_DEPTH - 1; level >= 0; level--)
    hipLaunchKernelGGL(oct_size_level_kernel, grid, block, 0, st, n_nodes, level, (const F2nTreeNode*) nodes, depth, visit_cnt,
                       brute_force, size);
  return f2n_launch_status();
}

int f2n_oct_subdivide(void* stream, int n_nodes, const void* nodes, const int32_t* visit_cnt, int brute_force, const int32_t* size,
                      const int32_t* w_stats, const int32_t* a_stats, int32_t* new_idx, void* dst_nodes, int32_t* dst_w, int32_t* dst_a) {
  if (n_nodes < 0) return F2N_ERR_INVALID_ARG;
  if (n_nodes == 0) return F2N_OK;
  hipLaunchKernelGGL(oct_subdivide_emit_kernel, dim3(f2n_div_up(n_nodes, 256)), dim3(256), 0, (hipStream_t) stream, n_nodes,
                     (const F2nTreeNode*) nodes, visit_cnt, brute_force, size, w_stats, a_stats, new_idx, (F2nTreeNode*) dst_nodes, dst_w,
                     dst_a);
  return f2n_launch_status();
}

}  // extern "C"
