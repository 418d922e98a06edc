/* f2n_abi.h -- C-ABI of the MI355X-native (gfx950, HIP) implementation of F2-NeRF's per-ray hot path.
 *
 * The reference (Totoro97/f2-nerf) has no FFI: its hot path is a set of first-party CUDA kernels launched
 * from C++ plugin classes (PersSampler / Hash3DAnchored / SHShader / Renderer) plus the third-party
 * tiny-cuda-nn `tcnn::cpp::Module`.  This header declares, seam for seam, what a binding for that path
 * would bind; each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/src).  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Every call only ENQUEUES work on
 *     that stream; nothing here allocates or frees caller-visible memory.
 *   - Internal scratch: the backward / loss / large-batch field entry points (f2n_field_fwd for n >= 8192, f2n_field_bwd*,
 *     f2n_mlp_bwd, f2n_shade_bwd*, f2n_hash_bwd with level_entries > 0, f2n_train_loss) keep per-block partial sums, feature
 *     planes and scatter queues in ONE library-owned workspace per device and slot.  Consequences: (a) calls that share a
 *     slot must be issued on ONE stream per device (or be ordered by the caller) -- two streams running f2n_field_bwd at
 *     the same time would share partials; the host layer issues all of them on the compute stream and only sampler kernels
 *     (which use no workspace) on its side stream; (b) the first calls, and a call that needs more scratch than any before
 *     it, grow the workspace: that path drains the device (hipDeviceSynchronize) before the old buffer is released,
 *     because queued kernels may still hold it.  Sizes settle after the first iterations; steady-state calls never
 *     synchronise.
 *   - All pointers are DEVICE pointers to contiguous buffers in exactly the layouts named below, unless
 *     the parameter is documented as host data.  TreeNode = 64 B, TransInfo = 544 B, EdgePool = 64 B as in
 *     PtsSampler/PersSampler.h:15-37.  "h16" = IEEE binary16.
 *   - Return value: 0 on success, F2N_ERR_* (< 0) otherwise; a HIP launch error e is returned as
 *     -(1000 + e).  Callable from any host thread (e.g. the autograd engine thread); the caller selects
 *     the device (hipSetDevice) before calling.
 *   - Variable-length results use count -> f2n_segment_scan -> fill, so no host read-back is needed:
 *     callers may size outputs for the worst case (n_rays * 1024) and read totals lazily.
 */
#ifndef F2N_ABI_H
#define F2N_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F2N_OK 0
#define F2N_ERR_INVALID_ARG (-1)
#define F2N_ERR_UNSUPPORTED (-2)

#define F2N_N_LEVELS 16            /* Field/Hash3DAnchored.h:16 */
#define F2N_N_CHANNELS 2           /* Field/Hash3DAnchored.h:15 */
#define F2N_MAX_SAMPLE_PER_RAY 1024 /* PtsSampler/PersSampler.cu:9 */
#define F2N_MLP_OUT_PAD 16         /* tcnn FullyFusedMLP pads the output width to 16 */

/* Library / device introspection (host-only). */
int f2n_abi_version(void);
/* 0 = product numerics; 1 = the reference-numerics build of this library (libf2n_hip_refnum.so, -DF2N_REFERENCE_NUMERICS=1):
 * the two places where the product deviates from the reference's arithmetic ON PURPOSE are switched back -- the hash
 * gradient is accumulated by per-addend packed-f16 atomics in arrival order (Hash3DAnchored.cu:145-153) instead of fp32 /
 * fp64 owner sums rounded once, and the MLP forward products use an f16 accumulator fragment (the other plausible reading
 * of tcnn's FullyFusedMLP).  Same ABI; used to A/B whole trainings (bench.py "psnr_numerics_ab"). */
int f2n_numerics_mode(void);
const char* f2n_build_info(void);
/* Data-parallel runs (one all-reduce of the hash-table gradient per step): the owner launch of the binned scatter behind
 * f2n_hash_bwd / f2n_field_bwd[_dyn] can be cut into n_buckets launches over contiguous ranges of table slices -- bucket b = slices
 * [b * S / n, (b + 1) * S / n) of the S = 17 * (level_entries / 8192) slices of 4096 entries (8192 halves) that the active prefix of
 * the table spans -- and fn(user, b, n) is called on the calling host thread right behind bucket b's launch: that range of the
 * gradient table is final once the launch has run, so its all-reduce can start (on another stream, behind an event) while the
 * later buckets' owners still run, instead of the whole 17 MiB waiting for the step's last kernel.  Per device; n_buckets <= 1 or
 * fn == NULL removes the hook.  A scatter that does not take the binned path (small batches, tiny tables) calls nothing: the caller
 * sends what is left when the backward returns (host/DataParallel.cpp). */
typedef void (*f2n_bucket_fn)(void* user, int bucket, int n_buckets);
int f2n_set_scatter_buckets(int n_buckets, f2n_bucket_fn fn, void* user);
/* ABI v13 (round-5 advisor): the hook is for ONE gradient table -- only a scatter into grad_table_h16 is cut into buckets and
 * reports them; any other f2n_hash_bwd / f2n_field_bwd on the device (a second runner, a test, a taped backward into another
 * buffer) runs its owner launch in one piece and calls nobody.  grad_table_h16 == NULL: whatever table (the v12 behaviour). */
int f2n_set_scatter_buckets_for(int n_buckets, f2n_bucket_fn fn, void* user, const void* grad_table_h16);
/* Diagnostics (host-only, synchronous; no reference counterpart): eight device-side event counters copied to host memory,
 * optionally reset.  [0] = scatter records of f2n_hash_bwd's owner-binned path that found their queue segment full and were
 * applied by a packed-f16 atomic instead (the only order-dependent addition of that path); [1] = table slices whose owner left
 * its packed fixed-point image for exact fp64 sums because the slice's addends could have left the 32-bit fields' range (same
 * bits either way: a cost counter, not an error counter); [2] (debug variant only) = the largest sum of |addend| an owner saw in
 * a slice, as float bits; [3] = scatter records that found their queue segment full and travelled through their producer block's
 * overflow list instead (summed by the owners like every other record: order-free; [0] counts only what found that list full
 * too).  The rest are reserved (0). */
int f2n_debug_counters(int32_t* out8_host /* or NULL */, int reset);
/* (The two debugging launches of rounds 3-4 -- a delay on a stream, a launch that leaves garbage in every CU's LDS and registers
 * -- are not part of this ABI: include/f2n_debug.h, compiled into the debug variant of the library only.) */

/* ---------------------------------------------------------------------------------------------------
 * Sampler -- replaces PersSampler::GetSamples' kernels (PtsSampler/PersSampler.cu:21-434).
 * ------------------------------------------------------------------------------------------------- */

/* rays_d / ||rays_d|| (PersSampler.cu:319, torch::linalg_norm there).  Fixed fp32 order: sqrt((x*x + y*y) + z*z),
 * IEEE division, so that the CPU oracle restates it bit for bit.  out may alias dirs. */
int f2n_normalize_dirs(void* stream, int n, const float* dirs /*[n,3]*/, float* out /*[n,3]*/);
/* f2n_normalize_dirs + zero[0..n_zero) = 0 (the hit / sample totals of f2n_oct_intersect_strided / f2n_segment_scan) +
 * f2n_march_noise (u -> noise_out, n_noise values; 0: none) in ONE launch: the head of a GetSamples call. */
int f2n_sampler_prologue(void* stream, int n_rays, const float* dirs, float* out, int32_t* zero, int n_zero, int n_noise,
                         const float* u, float fineness, float* noise_out);
/* The same with the march noise DRAWN by the launch (round 5): element i of the noise is uniform number i of the batch with sequence
 * number `seq` under `key` (Philox4x32-10, counter = (i / 4, 0, seq), key = seed ^ purpose: host/KeyedDraws.h) -- a function of what
 * the draw is for, not of how many draws were made before it -- mapped like f2n_march_noise.  No rand launch in front of the chain. */
int f2n_sampler_prologue_keyed(void* stream, int n_rays, const float* dirs, float* out, int32_t* zero, int n_zero, int n_noise,
                               uint64_t key, uint64_t seq, float fineness, float* noise_out /*[n_noise]*/);

/* FindRayOctreeIntersectionKernel<false> (PersSampler.cu:53-152, launched :342-351).
 * rays_d must already be unit length (GetSamples normalises at :319); [near, far] is the global bound the
 * reference substitutes for its `bounds` argument (:322-323: near = pts_sampler.near, far = 1e8).
 * hit_counts[r] = number of valid leaves hit by ray r, capped at max_hits. */
int f2n_oct_intersect_count(void* stream, int n_rays, int max_hits, const uint8_t* search_order /*[64]*/,
                            const float* rays_o /*[R,3]*/, const float* rays_d /*[R,3]*/, float near_, float far_,
                            const void* tree_nodes, int32_t* hit_counts /*[R]*/, const void* child_blocks /*or NULL*/);

/* Ray-ordered segment allocation: start_end[i] = (sum_{j<i} counts[j], sum_{j<=i} counts[j]);
 * total[0] = sum.  Replaces the racing atomicAdd allocator (PersSampler.cu:144) and the
 * torch::cumsum + .item() host sync (:395-397); also FilterIdxBounds' cumsum (Renderer/Renderer.cu:44-48). */
int f2n_segment_scan(void* stream, int n, const int32_t* counts, int32_t* start_end /*[n,2]*/, int32_t* total /*[1]*/);
/* The same, and the host's copy of the result from the same launch: `mirror` (or NULL) is a DEVICE pointer to MAPPED HOST
 * memory (hipHostMallocMapped + hipHostGetDevicePointer) that receives also[0..n_also) followed by the total -- e.g. the hit
 * total of f2n_oct_intersect_strided next to the sample total, the pair the reference reads back with two .item() calls
 * (PersSampler.cu:353,397).  The host reads it after an event recorded behind this call; no device-to-host copy is queued
 * (on an in-order queue that copy is one more dependent launch between the scan and the kernel that consumes it).
 * n_also <= 4. */
int f2n_segment_scan_ex(void* stream, int n, const int32_t* counts, int32_t* start_end /*[n,2]*/, int32_t* total /*[1]*/,
                        int32_t* mirror /*mapped host [n_also+1] or NULL*/, const int32_t* also /*device [n_also] or NULL*/,
                        int n_also);

/* FindRayOctreeIntersectionKernel<true> (PersSampler.cu:357-366): fills each ray's segment with
 * (leaf node index, t_near, t_far), front to back. */
int f2n_oct_intersect_fill(void* stream, int n_rays, const uint8_t* search_order, const float* rays_o,
                           const float* rays_d, float near_, float far_, const void* tree_nodes,
                           const int32_t* oct_start_end /*[R,2]*/, int32_t* oct_idx /*[K]*/,
                           float* oct_near_far /*[K,2]*/, const void* child_blocks /*or NULL*/);

/* Single-pass variant of count + scan + fill for callers that do not need a compact list: ray r owns the fixed
 * slot range [r*max_hits, r*max_hits + hits_r) of oct_idx / oct_near_far (both sized n_rays*max_hits);
 * oct_start_end[r] = that range, total[0] += sum of hits (must be zeroed by the caller).  Per-ray contents are
 * identical to the count/fill pair; the march kernels accept either segment layout. */
int f2n_oct_intersect_strided(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                              const float* rays_d, float near_, float far_, const void* tree_nodes,
                              int32_t* oct_start_end /*[R,2]*/, int32_t* oct_idx /*[R*max_hits]*/,
                              float* oct_near_far /*[R*max_hits,2]*/, int32_t* total /*[1]*/,
                              int32_t* oct_trans /*[R*max_hits] or NULL: trans_idx of every listed leaf*/,
                              const void* child_blocks /*or NULL*/);
/* The same walk with the records of every node that HAS a child (the only ones a walk expands) copied into LDS when a block
 * starts: interior_nodes[r] = index of the r-th such node in index order, rank_of[node] = its r (both derived from the node
 * array; they change only when child indices do, i.e. with f2n_oct_build_child_blocks).  For trees of up to
 * f2n_oct_lds_max_interior() such nodes (F2N_ERR_UNSUPPORTED beyond).  Output bit-identical to f2n_oct_intersect_strided; the
 * point is latency: the reference's one-thread-per-ray DFS (PersSampler.cu:53-152) and the walk above are chains of dependent
 * reads, which take ~6x longer when the kernel runs underneath a kernel that saturates the L2s (as the prefetched sampling of
 * the next batch does, under the hash gather). */
int f2n_oct_lds_max_interior(void);
int f2n_oct_intersect_strided_lds(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                                  const float* rays_d, float near_, float far_, const void* tree_nodes,
                                  int32_t* oct_start_end /*[R,2]*/, int32_t* oct_idx /*[R*max_hits]*/,
                                  float* oct_near_far /*[R*max_hits,2]*/, int32_t* total /*[1]*/,
                                  int32_t* oct_trans /*[R*max_hits] or NULL*/, const void* child_blocks,
                                  const int32_t* interior_nodes /*[n_interior]*/, const int32_t* rank_of /*[n_nodes]*/,
                                  int n_interior);


/* Optional acceleration structure for the three intersection entry points: child_blocks [n_nodes][8] x 32 B, entry
 * [u][c] = {center xyz, side_len, child index (-1: none), child's trans_idx, child has children, pad} of child slot c of
 * node u.  With it, expanding a node is one coalesced 256-byte read instead of the reference's dependent pair
 * "childs[] of the node, then the child nodes" (PersSampler.cu:93-120); results are identical.  Rebuild after every
 * host-side change of the tree; f2n_oct_update_stats keeps the trans_idx copies current when given the pointer. */
int f2n_oct_build_child_blocks(void* stream, int n_nodes, const void* tree_nodes, void* child_blocks);

/* GetVisiCams (PtsSampler/PersSampler.cpp:27-66) for n_boxes candidate octree nodes at once:
 * visible[b, c] = 1 iff any ray of camera c's res_h x res_w pixel-grid bundle (pixel centres pix_i [res_h] rows,
 * pix_j [res_w] columns; direction ((j-cx)/fx, -(i-cy)/fy, -1) rotated by c2w[c]) hits box b = (centre xyz, side)
 * within the camera's [near, far] bounds.  boxes [n_boxes,4], c2w [n_cams,3,4], bounds [n_cams,2]. */
int f2n_oct_visible_cams(void* stream, int n_boxes, int n_cams, const float* boxes, const float* c2w,
                         const float* bounds, float fx, float fy, float cx, float cy, int res_h, int res_w,
                         const float* pix_i, const float* pix_j, uint8_t* visible /*[n_boxes,n_cams]*/);

/* The march noise of PersSampler.cu:372-381 from uniform draws u in [0,1): out = ((u - 0.5) + 1) * fineness, in that
 * fp32 order (the reference: three ATen launches). */
int f2n_march_noise(void* stream, int n, const float* u, float fineness, float* out);

/* RayMarchKernel<false> (PersSampler.cu:189-314, launched :383-393).  noise has
 * F2N_MAX_SAMPLE_PER_RAY + n_rays + 10 floats, already multiplied by ray_march_fineness (:372-381), and is
 * indexed [ray + k] exactly as in the reference (:203,:266). */
int f2n_ray_march_count(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o,
                        const float* rays_d, const float* noise, const int32_t* oct_start_end,
                        const int32_t* oct_idx, const float* oct_near_far, const void* tree_nodes,
                        const void* transes, int32_t* pts_counts /*[R]*/);

/* RayMarchKernel<true> (PersSampler.cu:407-423).  Outputs are the SampleResultFlex tensors
 * (PtsSampler/PtsSampler.h:13-22): pts = WARPED coordinates, dirs = unit world direction, dt = warped-space
 * step, t = world distance, anchors[:,0] = trans_idx, anchors[:,1] = leaf node index, anchors[:,2] = 0. */
int f2n_ray_march_fill(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o,
                       const float* rays_d, const float* noise, const int32_t* oct_start_end,
                       const int32_t* oct_idx, const float* oct_near_far, const void* tree_nodes,
                       const void* transes, const int32_t* pts_start_end /*[R,2]*/, float* pts /*[N,3]*/,
                       float* dirs /*[N,3]*/, float* dt /*[N]*/, float* t /*[N]*/, int32_t* anchors /*[N,3]*/,
                       float* first_oct_dis /*[R]*/);

/* Single-pass variant of count + scan + fill: ONE march writes every ray's samples into its fixed-stride slot
 * [r*F2N_MAX_SAMPLE_PER_RAY, ...) of s_pts [R*1024,3], s_dt / s_t [R*1024], s_anchors [R*1024,2] = (trans_idx, node)
 * and its sample count into pts_counts[r]; after f2n_segment_scan(pts_counts), f2n_pack_samples copies the filled
 * prefixes into the ray-ordered compact SampleResultFlex arrays (dirs come from rays_d, anchors[:,2] = 0).
 * Per-ray contents are identical to the count/fill pair (same march, executed once instead of twice); the slot
 * buffers are scratch of which only the filled prefixes are ever touched. */
int f2n_ray_march_strided(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o,
                          const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                          const float* oct_near_far, const void* tree_nodes, const void* transes,
                          int32_t* pts_counts /*[R]*/, float* s_pts, float* s_dt, float* s_t, int32_t* s_anchors,
                          float* first_oct_dis /*[R]*/,
                          const int32_t* oct_trans /*NULL, or f2n_oct_intersect_strided's: spares a node read per leaf*/);
/* s_pts may be NULL in both calls: the march then skips the warped point (one division and three reductions less on its
 * sequential path) and f2n_pack_samples computes pts = warp(o + d*t) for all samples in parallel -- same expressions,
 * same bits (PersSampler.cu:155-169); rays_o and transes are only read in that case. */
int f2n_pack_samples(void* stream, int n_rays, const int32_t* pts_start_end /*[R,2]*/, const float* rays_o,
                     const float* rays_d, const void* transes, const float* s_pts, const float* s_dt, const float* s_t,
                     const int32_t* s_anchors, float* pts /*[N,3]*/, float* dirs /*[N,3]*/, float* dt /*[N]*/,
                     float* t /*[N]*/, int32_t* anchors /*[N,3]*/);

/* GetEdgeSamplesKernel (PersSampler.cu:436-452).  edge_idx/edge_coords are the random draws of :456-457. */
int f2n_edge_samples(void* stream, int n_pts, const void* edge_pool, const void* transes, const int32_t* edge_idx,
                     const float* edge_coords /*[n,2]*/, float* out_pts /*[n,2,3]*/, int32_t* out_idx /*[n,2]*/);
/* The same with (a) the draws optionally given as u01 [n,3] uniform in [0,1) -- idx = min(floor(u0 * n_edges), n_edges - 1),
 * coords = 2 u - 1: one random launch for :456-457 instead of two (edge_idx / edge_coords are then ignored, may be NULL) --,
 * (b) a stride for the index output (3: the transform column of an anchors array [.,3]) and (c) an optional second
 * destination (out_pts2 / out_idx2 both NULL or both given): a streaming training step appends the edge samples to the
 * sampler's arrays so that the density pre-pass gathers their hash features too, and heads the grad pass's arrays with them. */
int f2n_edge_samples_ex(void* stream, int n_pts, const void* edge_pool, int n_edges, const void* transes, const int32_t* edge_idx,
                        const float* edge_coords, const float* u01, float* out_pts, int32_t* out_idx, int idx_stride,
                        float* out_pts2, int32_t* out_idx2, int idx_stride2);

/* MarkVistNodeKernel (PersSampler.cu:475-526).  oct node index of sample i = anchors[i*anchor_stride + 1].
 * w_adder/a_adder must be pre-filled with -1, mark with 0 (:555-557); visit_cnt is persistent state. */
int f2n_oct_mark_visit(void* stream, int n_rays, int n_nodes, const int32_t* pts_start_end, const int32_t* anchors,
                       int anchor_stride, const float* weights, const float* alphas, int32_t* w_adder,
                       int32_t* a_adder, int32_t* mark, int32_t* visit_cnt);

/* The integer stat update of UpdateOctNodes (PersSampler.cu:579-593) fused with MarkInvalidNodes
 * (:528-534, :595-603): stats = clamp(max(stats, pos vote) + visited negative vote, -100, 2^20);
 * trans_idx = -1 where either stat < 0.  reset_votes != 0: the three vote buffers are re-initialised after use
 * (adders = -1, mark = 0: what the reference re-creates with torch::full / zeros every iteration, :555-557), so a
 * caller can keep one persistent buffer. */
int f2n_oct_update_stats(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                         int32_t* a_stats, void* tree_nodes, void* child_blocks /*or NULL*/, int reset_votes);

/* ---- Speculative sampling (ABI v8) -------------------------------------------------------------------------------------
 * The reference samples a batch AFTER the previous iteration's UpdateOctNodes (ExpRunner.cpp:88-93 -> Renderer.cpp:100 ->
 * PersSampler.cu:317, behind :536-615 of the iteration before): intersection and march of batch k+1 wait for the density
 * pre-pass of batch k.  Outside the ProcOctree iterations the only thing that update changes in the tree is trans_idx -> -1 of
 * the leaves that die (MarkInvalidNodes, :528-534), so batch k+1 can be sampled EARLY, against the tree as it stands, and
 * repaired afterwards: a ray whose leaf list holds no node that died since is already exact (see oct_intersect_coop_kernel
 * MODE 3 in csrc/sampler.hip for the argument), the others are walked and marched again into their fixed-stride slots.
 *   f2n_oct_update_stats_ex   as f2n_oct_update_stats; a leaf that dies in this call gets died_at[node] = epoch and
 *                             death_epoch[0] = epoch (both or neither may be NULL; epochs must increase from call to call)
 *   f2n_oct_intersect_repair  on the outputs of f2n_oct_intersect_strided computed while / before the updates of epochs
 *                             >= spec_epoch ran: rays whose list holds a node with died_at >= spec_epoch (or is at the
 *                             max_hits cap) are walked again on the updated tree, total[0] is adjusted, repair_flags[r] = 1 for
 *                             those rays and 0 for all others, n_repaired[0] (or NULL) += their number.  Returns immediately on
 *                             the device when death_epoch[0] < spec_epoch (then repair_flags is NOT written).
 *   f2n_ray_march_repair      f2n_ray_march_strided for the flagged rays only (same early exit).
 *   f2n_pack_samples_repair   f2n_pack_samples once more over the whole batch, but only when a leaf died since (same early exit):
 *                             a batch may be scanned and packed OPTIMISTICALLY right behind its speculative march -- long before
 *                             the stat update, in a stretch of the step where the memory system has room -- and is scanned again
 *                             and conditionally packed again behind the two repair calls.
 * After both, slots / counts / first_oct_dis are bit-identical to a fresh f2n_oct_intersect_strided + f2n_ray_march_strided on
 * the updated tree (tests/test_gpu_parity.py::test_speculative_sampling_repair). */
int f2n_oct_update_stats_ex(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                            int32_t* a_stats, void* tree_nodes, void* child_blocks /*or NULL*/, int reset_votes,
                            int32_t* died_at /*[n_nodes] or NULL*/, int epoch, int32_t* death_epoch /*[1] or NULL*/,
                            int32_t* death_epoch_host /*[1] device-accessible HOST word or NULL: also receives the epoch of a
                            death -- a hint the host may read without synchronising (whether speculating pays right now)*/);
/* f2n_oct_update_stats_ex and f2n_segment_scan_ex in ONE launch (round 5): the stat update of a streaming training step
 * (PersSampler.cu:579-593, :528-534) and the scan of the rays' surviving-sample counts (FilterIdxBounds, Renderer.cu:20-50) are
 * independent and used to be two dependent launches of the step's main queue.  Same results as the two calls. */
int f2n_oct_update_stats_scan(void* stream, int n_nodes, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* w_stats,
                              int32_t* a_stats, void* tree_nodes, void* child_blocks, int reset_votes, int32_t* died_at, int epoch,
                              int32_t* death_epoch, int32_t* death_epoch_host, int n, const int32_t* counts, int32_t* start_end /*[n,2]*/,
                              int32_t* total /*[1]*/, int32_t* mirror, const int32_t* also, int n_also);
int f2n_oct_intersect_repair(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                             const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                             int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans, const void* child_blocks,
                             const int32_t* died_at, int spec_epoch, const int32_t* death_epoch, int32_t* repair_flags /*[R]*/,
                             int32_t* n_repaired /*[1] or NULL*/);
int f2n_ray_march_repair(void* stream, int n_rays, float sample_l, int scale_by_dis, const float* rays_o, const float* rays_d,
                         const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx, const float* oct_near_far,
                         const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts, float* s_dt, float* s_t,
                         int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans, const int32_t* repair_flags,
                         const int32_t* death_epoch, int spec_epoch);
int f2n_pack_samples_repair(void* stream, int n_rays, const int32_t* pts_start_end, const float* rays_o, const float* rays_d,
                            const void* transes, const float* s_pts, const float* s_dt, const float* s_t, const int32_t* s_anchors,
                            float* pts, float* dirs, float* dt, float* t, int32_t* anchors, const int32_t* death_epoch /*[1]*/,
                            int spec_epoch);

/* ---- Tail repair of a speculatively sampled batch (ABI v10) -------------------------------------------------------------
 * The repair above walks and marches an invalidated ray again from its origin: as latency-bound as the sampler itself
 * (the longest invalidated ray), and it sits on the step's critical cycle.  Neither is necessary.  (1) The walk reads trans_idx
 * only to decide whether a leaf it has reached is listed, so the list a fresh walk produces is the old list minus its dead
 * entries (unless the old list was cut at max_hits).  (2) A march iteration looks at list entries only when it crosses out of
 * its current leaf, so every iteration before the one that first reached the first removed entry is untouched.
 *   f2n_ray_march_strided_rec  f2n_ray_march_strided that also records, per list entry it crosses into or past, the state the
 *                              iteration started from -- leaf_state[ray * max_hits + entry] = {bits of t, samples emitted |
 *                              list position << 11 | first-point flag << 22} -- and reached[ray] = the last list position it
 *                              looked at.  max_hits <= 2048.
 *   f2n_oct_list_repair        removes the entries whose node has died_at >= spec_epoch from every list IN PLACE (order, near /
 *                              far, trans kept), adjusts oct_start_end / total[0]; repair_from[ray] = position of the first
 *                              removed entry if the march had reached it, F2N_REPAIR_NONE (-1) if the ray needs no new march,
 *                              F2N_REPAIR_FULL (-2) if the list was cut at max_hits and lost an entry (n_full[0] += 1; the list is
 *                              left as it was).  Same early exit as the calls above (repair_from is then NOT written).
 *   f2n_oct_intersect_repair_flagged  walks the F2N_REPAIR_FULL rays again (returns at once when n_full[0] == 0) and sets their
 *                              repair_from to 0.
 *   f2n_ray_march_repair_tail  resumes the march of every ray with repair_from >= 0 from the state recorded at that entry (0: from
 *                              the origin) on the repaired list, recording again.
 * Result: bit-identical to a fresh f2n_oct_intersect_strided + f2n_ray_march_strided on the updated tree
 * (tests/test_gpu_scale.py::test_speculative_tail_repair). */
#define F2N_REPAIR_NONE (-1)
#define F2N_REPAIR_FULL (-2)
int f2n_ray_march_strided_rec(void* stream, int n_rays, int max_hits, float sample_l, int scale_by_dis, const float* rays_o,
                              const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                              const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                              float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                              void* leaf_state /*[R * max_hits] 8-byte records*/, int32_t* reached /*[R]*/);
/* f2n_ray_march_strided[_rec] on a small persistent grid: n_blocks persistent WAVES (in workgroups of block_waves waves, 1..16:
 * ABI v13 -- a workgroup's waves land on one CU, so 512 waves in blocks of 8 leave three quarters of the chip's CUs to the
 * register-hungry kernels of the main queue instead of parking one or two march waves on every CU) take groups of four rays off counter[0], rays
 * ordered by leaf count, longest first (order[R]: scratch, filled here; counter[1]: scratch, zeroed here).  For batches that are
 * sampled well ahead of their use (two-deep pipeline): a few hundred resident waves instead of R / 4, so that the occupancy-bound
 * kernels the march runs underneath keep their wave slots.  leaf_state / reached both NULL: no recording.  Same slots, same
 * bits (tests/test_gpu_scale.py::test_persistent_march). */
int f2n_ray_march_persistent(void* stream, int n_rays, int max_hits, int n_blocks, int block_waves, float sample_l, int scale_by_dis, const float* rays_o,
                             const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                             const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                             float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                             void* leaf_state /*or NULL*/, int32_t* reached /*or NULL*/, int32_t* order /*[R]*/, int32_t* counter /*[1]*/);
int f2n_oct_list_repair(void* stream, int n_rays, int max_hits, int32_t* oct_start_end, int32_t* oct_idx, float* oct_near_far,
                        int32_t* oct_trans /*or NULL*/, int32_t* total, const int32_t* died_at, int spec_epoch,
                        const int32_t* death_epoch, const int32_t* reached, int32_t* repair_from /*[R]*/,
                        int32_t* n_repaired /*[1] or NULL*/, int32_t* n_full /*[1], zeroed by the caller*/);
int f2n_oct_intersect_repair_flagged(void* stream, int n_rays, int max_hits, const uint8_t* search_order, const float* rays_o,
                                     const float* rays_d, float near_, float far_, const void* tree_nodes, int32_t* oct_start_end,
                                     int32_t* oct_idx, float* oct_near_far, int32_t* total, int32_t* oct_trans,
                                     const void* child_blocks, const int32_t* death_epoch, int spec_epoch, int32_t* repair_from,
                                     const int32_t* n_full);
int f2n_ray_march_repair_tail(void* stream, int n_rays, int max_hits, float sample_l, int scale_by_dis, const float* rays_o,
                              const float* rays_d, const float* noise, const int32_t* oct_start_end, const int32_t* oct_idx,
                              const float* oct_near_far, const void* tree_nodes, const void* transes, int32_t* pts_counts, float* s_pts,
                              float* s_dt, float* s_t, int32_t* s_anchors, float* first_oct_dis, const int32_t* oct_trans,
                              void* leaf_state, int32_t* reached, const int32_t* repair_from, const int32_t* death_epoch,
                              int spec_epoch);

/* MarkInvisibleNodesKernel (PersSampler.cu:618-680). */
int f2n_oct_mark_invisible(void* stream, int n_nodes, int n_cams, void* tree_nodes, const float* intris /*[C,3,3]*/,
                           const float* w2cs /*[C,3,4]*/, const float* bounds /*[C,2]*/);

/* ---------------------------------------------------------------------------------------------------
 * Hash grid -- replaces Hash3DAnchoredFunction::forward/backward (Field/Hash3DAnchored.cu:11-233).
 * `table_h` is the binary16 table [pool,2]; level l lives at table_h + local_idx[l] HALVES and is
 * addressed as pos*2+k with pos < local_size[l] (the 50 % level-overlap quirk of :37/:74-77 is kept).
 * level_scale[16] are the per-level multipliers exp2f(7*l/15+3) (:28) supplied by the host so that both
 * sides of a parity test use identical bits.  If pts_are_warped != 0 the kernel applies the
 * ((p + 1) * .5) of Hash3DAnchored.cpp:91 itself.  volume index of point i = volume_idx[i*vol_stride].
 * ------------------------------------------------------------------------------------------------- */
int f2n_hash_fwd(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool /*[16,V,3]*/,
                 const int32_t* local_idx /*[16]*/, const int32_t* local_size /*[16]*/,
                 const float* bias_pool /*[16*V,3]*/, const float* level_scale /*[16]*/, const float* pts /*[n,3]*/,
                 int pts_are_warped, const int32_t* volume_idx, int vol_stride, void* out_h /*[n,32] h16*/);

/* grad_in_h: dL/dfeat already scaled by 128 and rounded to h16 (Hash3DAnchored.cu:220).  grad_table_h is
 * ACCUMULATED into (the reference's half2 atomicAdd, :151) and must be zeroed by the caller (:222).  The /128 and
 * fp32 widening of :232 are left to the optimiser step (f2n_adam_step_h16grad).
 * level_entries: 0, or the HOST-side statement that the table has the reference's layout (Hash3DAnchored.cpp:60-70):
 * local_size[l] = level_entries (a power of two) and local_idx[l] = l * level_entries halves for every level
 * (the device arrays cannot be read without a sync).  It lets large batches use the owner-binned scatter: records
 * binned per 4096-entry table slice, summed in LDS in fp64 by one owner block per slice and added to the table with
 * plain stores -- no global atomics (measured 3x faster at 8e5 samples).  0 = always packed-h16 global atomics,
 * any layout. */
int f2n_hash_bwd(void* stream, int n, int n_volumes, const int32_t* prim_pool, const int32_t* local_idx,
                 const int32_t* local_size, const float* bias_pool, const float* level_scale, const float* pts,
                 int pts_are_warped, const int32_t* volume_idx, int vol_stride, const void* grad_in_h /*[n,32]*/,
                 void* grad_table_h, int level_entries);

/* ---------------------------------------------------------------------------------------------------
 * Fully-fused MLP -- replaces the tcnn::cpp::Module surface used by Field/TCNNWP.cpp:94-97,150-154,
 * 217-227 ("FullyFusedMLP", ReLU, no output activation, no bias).  Supported shapes: d_in = 32,
 * d_hidden = 64, n_hidden in {1,2}, d_out <= 16 (padded to 16) -- the two networks of the reference
 * configs.  Parameter layout (fp32 master and h16 working copy alike): layers first->last, each
 * [n_out, n_in] row-major, last layer has 16 rows.  The h16 parameter block handed to the *_bwd entry points must be
 * 16-byte aligned (F2N_ERR_INVALID_ARG otherwise).
 * ------------------------------------------------------------------------------------------------- */
int f2n_mlp_n_params(int d_in, int d_hidden, int n_hidden);               /* Module::n_params() */
/* Module::initialize_params(seed, float*): Xavier-uniform from a counter-based generator (tcnn's pcg32
 * stream cannot be reproduced; weights are exchanged as flat fp32 arrays instead). */
int f2n_mlp_init_params(void* stream, uint64_t seed, int d_in, int d_hidden, int n_hidden, float* params_f32);
int f2n_params_to_h16(void* stream, int n, const float* params_f32, void* params_h); /* TCNNWP.cpp:111 */
/* Module::inference / forward: x is fp32 [n,32] (tcnn input precision), out is h16 [n,16]
 * (output_precision()).  No activations are stored: the backward recomputes them on the matrix cores. */
int f2n_mlp_fwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, const void* params_h, const float* x,
                void* out_h);
/* Module::backward: dy fp32 [n,16] is multiplied by loss_scale and rounded to h16 (TCNNWP.cpp:174);
 * dparams_f32 [n_params] is ACCUMULATED (fp32 atomics) in the SCALED domain -- divide by loss_scale
 * afterwards (:232); dx_f32 (may be NULL) receives dL/dx / loss_scale (:231). */
int f2n_mlp_bwd(void* stream, int n, int d_in, int d_hidden, int n_hidden, float loss_scale, const void* params_h,
                const float* x, const float* dy, float* dparams_f32_scaled, float* dx_f32);

/* ---------------------------------------------------------------------------------------------------
 * Fused field = hash gather + density MLP (Hash3DAnchored::AnchoredQuery, Field/Hash3DAnchored.cpp:84-99).
 * Features never leave the register file between the gather and the matrix cores.
 *   out_feat_f32 [n,16]   (may be NULL) the AnchoredQuery result (h16 values widened, TCNNWP.cpp:112)
 *   out_f0       [n]      (may be NULL) channel 0 only: the density pre-activation for the no-grad
 *                         pre-pass of Renderer::Render (Renderer/Renderer.cpp:105-137)
 *   save_x_h     [n,32]   (may be NULL) the h16 hash features, kept for f2n_field_bwd
 * ------------------------------------------------------------------------------------------------- */
int f2n_field_fwd(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                  const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                  const float* level_scale, const float* pts_warped, const int32_t* volume_idx, int vol_stride,
                  const void* mlp_params_h, float* out_feat_f32, float* out_f0, void* save_x_h);
/* f2n_field_fwd's one-kernel path (hash gather -> MLP inside one wave, features never in HBM) at ANY batch size: the A/B
 * comparator of the XCD-partitioned gather + plane-fed MLP that f2n_field_fwd uses for large batches.  Same results. */
int f2n_field_fwd_fused(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool, const int32_t* local_idx,
                        const int32_t* local_size, const float* bias_pool, const float* level_scale, const float* pts_warped,
                        const int32_t* volume_idx, int vol_stride, const void* mlp_params_h, float* out_feat_f32 /*or NULL*/,
                        float* out_f0 /*or NULL*/, void* save_x_h /*or NULL*/);

/* The two halves of f2n_field_fwd's large-batch path, callable on their own.
 * f2n_hash_gather_planes: the 16-level gather with an XCD-aware partition -- workgroup b serves levels (2p, 2p+1),
 * p = b % 8, so that each of the 8 private 4 MiB L2s of an MI355X only ever sees a 3 MiB slice of the table
 * (~2.6x the rate of one wave gathering all 16 levels).  Output: h16 "planes" [8][n][4], plane p = features
 * 4p..4p+3 (levels 2p, 2p+1) of every sample; numerically the same features as f2n_hash_fwd (same fp32 order).
 * f2n_field_mlp_planes: the density MLP on such planes; outputs as f2n_field_fwd. */
int f2n_hash_gather_planes(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                           const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                           const float* level_scale, const float* pts, int pts_are_warped, const int32_t* volume_idx,
                           int vol_stride, void* planes_h /*[8,n,4] h16*/);
/* The same gather with run combining and a cost-balanced split (ABI v6).  step01 = typical distance of consecutive
 * samples of a ray in the [0,1] hash space (the march's warped-space step / 2: sample_l * fineness / 2,
 * PersSampler.cu:262-270 and Hash3DAnchored.cpp:91); level_scale_host = the 16 level scales as a HOST array.  Samples
 * must be ray-ordered (as PersSampler::GetSamples emits them).  Consecutive samples in one cell of a level share their
 * eight table reads; the level pairs, which then differ in cost by up to 10x, are spread over the XCDs by expected cost
 * instead of one pair per XCD.  Bit-identical planes; step01 <= 0 or level_scale_host == NULL: exactly
 * f2n_hash_gather_planes. */
int f2n_hash_gather_planes_balanced(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                                    const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                                    const float* level_scale, const float* pts, int pts_are_warped, const int32_t* volume_idx,
                                    int vol_stride, void* planes_h /*[8,n,4] h16*/, float step01, const float* level_scale_host);
/* Introspection (host only, no device work): the split f2n_hash_gather_planes_balanced would use for n_tiles tiles of 256
 * samples.  out [8][1 + 3*8] int32: per XCD the number of segments, then (level pair, first tile, tile count) per
 * segment (-1, 0, 0 for unused slots); cost8_out [8] (may be NULL): the modelled cost of one tile of each level pair. */
int f2n_gather_plan_query(int n_tiles, float step01, const float* level_scale_host, int32_t* out, float* cost8_out);
/* f2n_hash_gather_planes for tables that have left the L2s (Field/Hash3DAnchored.cu:11-79 at confs/wanjinyou_big.yaml's sizes and
 * beyond; level_entries = local_size[l] = a power of two, 8 * 8192 ... 2^23): level pairs >= first_binned_pair (0 ... 8) go through
 * a slice-binned pipeline -- requests binned by 8192-entry table slice per 1536-sample chunk, every slice read once into LDS and
 * its requests answered in place, values blended per sample in the reference's order -- instead of one 128-byte fabric line per
 * 4-byte read; pairs below it keep the partitioned gather.  Planes bit-identical to f2n_hash_gather_planes.
 * F2N_ERR_UNSUPPORTED: level_entries outside that range or n > 1536 * 1024.  Scratch: library workspace, 56 B per sample and
 * binned level (~0.65 GB at 8e5 samples and 14 levels). */
int f2n_hash_gather_planes_binned(void* stream, int n, int n_volumes, const void* table_h, const int32_t* prim_pool,
                                  const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                                  const float* level_scale, const float* pts, int pts_are_warped, const int32_t* volume_idx,
                                  int vol_stride, void* planes_h, int level_entries, int first_binned_pair);
/* ... and the kernel variant it would launch for these sizes (host only, ABI v8): bit 0 = hash constants staged in LDS
 * (2 * n_volumes * 24 B <= 20000 and n >= 16384), bit 1 = run combining + balanced split (the cost model predicts a balanced
 * share below 0.8 x the costliest level pair).  Negative = error.  What the parity tests assert before they claim to have
 * exercised a variant. */
int f2n_hash_gather_variant(int n, int n_volumes, float step01, const float* level_scale_host);
int f2n_field_mlp_planes(void* stream, int n, const void* planes_h, const void* mlp_params_h, float* out_feat_f32,
                         float* out_f0, void* save_x_h);

/* AnchoredQuery for samples whose hash features are already known: Renderer::Render queries the field twice per
 * step with an unchanged table -- the no-grad density pre-pass over every marched sample (Renderer.cpp:115-123)
 * and the grad pass over the survivors (:152-166) -- so the 128 gathers per surviving sample of the second query
 * are a pure recomputation.  x_cache_h [n_cache,32] is the save_x_h of the pre-pass; sample i of this call takes
 * row src_rows[i] (NULL = row i).  Outputs as f2n_field_fwd: bit-identical to it by construction (same h16
 * features through the same MLP kernel).  save_x_h [n,32] (may be NULL) receives the gathered rows in call order,
 * ready for f2n_field_bwd. */
int f2n_field_fwd_cached(void* stream, int n, int n_cache, const int32_t* src_rows, const void* x_cache_h,
                         const void* mlp_params_h, float* out_feat_f32, float* out_f0, void* save_x_h);

/* Backward of the fused field: MLP backward (dparams accumulated, scaled domain) chained into the hash scatter.
 * dfeat fp32 [n,16].  level_entries as in f2n_hash_bwd (0: dL/dx goes straight from registers into global atomics;
 * > 0 and a large batch: through 64 B/sample of f16 workspace into the owner-binned scatter). */
int f2n_field_bwd(void* stream, int n, int n_volumes, const int32_t* prim_pool, const int32_t* local_idx,
                  const int32_t* local_size, const float* bias_pool, const float* level_scale,
                  const float* pts_warped, const int32_t* volume_idx, int vol_stride, const void* mlp_params_h,
                  const void* saved_x_h, const float* dfeat, float loss_scale, float* dparams_f32_scaled,
                  void* grad_table_h, int level_entries);

/* ---------------------------------------------------------------------------------------------------
 * Shader -- replaces SHShader::Query (Shader/SHShader.cpp:23-29) and SHKenerl (Shader/SHShader.cu:10-118).
 * ------------------------------------------------------------------------------------------------- */
int f2n_sh_encode(void* stream, int n, int degree /*1..8*/, const float* dirs /*[n,3]*/, float* out /*[n,degree^2]*/);

/* ScatterIdxKernal (Utils/CustomOps/Scatter.cu:110-120): out[i] = ray_val[r] for every sample i of ray r. */
int f2n_scatter_idx(void* stream, int n_rays, const int32_t* start_end, const int32_t* ray_val, int32_t* out /*[n]*/);

/* Fused colour path of Renderer::Render (Renderer/Renderer.cpp:181-189):
 *   shading_feat = [1 | feat[:,1:16]] (+ app_emb[sample_emb_idx[i]] when app_emb != NULL, Scatter.cu:10-18),
 *   input = [shading_feat | SH4(dir)] -> colour MLP 32->64->64->16 -> rgb = (1+2e-3)/(1+exp(-o)) - 1e-3.
 * sample_emb_idx [n] is the ScatterIdx result (image index per sample), only read when app_emb != NULL.
 * save_x_h [n,32] (may be NULL) keeps the h16 MLP input for the backward. */
int f2n_shade_fwd(void* stream, int n, const float* feat /*[n,16]*/, const float* dirs /*[n,3]*/,
                  const float* app_emb /*[n_img,16] or NULL*/, const int32_t* sample_emb_idx /*[n] or NULL*/,
                  const void* mlp_params_h, float* rgb /*[n,3]*/, void* save_x_h);

/* drgb [n,3] -> dfeat[:,1:16] written; column 0 belongs to the density path: left untouched when df0 is NULL, or
 * filled from the compact array df0 [n] (f2n_composite_bwd with df0_stride 1) so that every dfeat row is written once,
 * as whole cache lines.  dparams accumulated (scaled domain), dapp_emb [n_img,16] accumulated UNSCALED or NULL.
 * The network output needed for the sigmoid derivative is recomputed from saved_x_h on the matrix cores.
 * n_emb <= 480: the embedding gradient is accumulated per block in LDS and reduced (no same-address global atomics);
 * larger image counts fall back to global atomics like the reference's ScatterAddFuncBackward (Scatter.cu:20-40).
 * sample_emb_idx values outside [0, n_emb) contribute nothing (f2n_shade_fwd trusts its indices: it has no n_emb). */
int f2n_shade_bwd(void* stream, int n, const float* drgb, const int32_t* sample_emb_idx, const void* mlp_params_h,
                  const void* saved_x_h, float loss_scale, float* dfeat /*[n,16]*/, float* dparams_f32_scaled,
                  float* dapp_emb /*[n_emb,16] or NULL*/, int n_emb, const float* df0 /*[n] or NULL*/);

/* ---------------------------------------------------------------------------------------------------
 * PersOctree::ProcOctree on the device (PtsSampler/PersSampler.cpp:120-330; ABI v6): the reference copies the node array
 * to the host, edits it with index-ordered loops and copies it back; these four calls produce the same arrays without
 * leaving HBM.  All buffers are caller-allocated; node arrays are TreeNode[] (64 B).
 *   f2n_oct_prune_compress  compact loop + path compression (:139-215) on a copy: out_nodes [n] = edited nodes in the OLD
 *                           numbering, keep [n] = 1 for the nodes that survive (:219-224).  work_nodes [n], alive [n],
 *                           n_child [n] are scratch.
 *   f2n_oct_gather_kept     renumbering (:226-252): new_pos = f2n_segment_scan(keep) start_end layout [n,2]; gathers nodes
 *                           (parent / child indices rewritten), both statistics and the visit counts.
 *   f2n_oct_subtree_sizes + f2n_oct_subdivide   subdivision (:255-318): leaves with visit_cnt > 4 (or all: brute_force)
 *                           split into 8 children that inherit warp and statistics, the whole tree renumbered depth-first;
 *                           size[0] after the first call is the new node count (read it to allocate dst_*).
 * ------------------------------------------------------------------------------------------------- */
int f2n_oct_prune_compress(void* stream, int n_nodes, const void* tree_nodes, void* work_nodes, void* out_nodes, int32_t* alive,
                           int32_t* n_child, int32_t* keep);
int f2n_oct_gather_kept(void* stream, int n_nodes, const void* nodes, const int32_t* keep, const int32_t* new_pos,
                        const int32_t* w_stats, const int32_t* a_stats, const int32_t* visit_cnt, void* dst_nodes, int32_t* dst_w,
                        int32_t* dst_a, int32_t* dst_visit);
int f2n_oct_subtree_sizes(void* stream, int n_nodes, const void* nodes, const int32_t* visit_cnt, int brute_force, int32_t* depth,
                          int32_t* size);
int f2n_oct_subdivide(void* stream, int n_nodes, const void* nodes, const int32_t* visit_cnt, int brute_force, const int32_t* size,
                      const int32_t* w_stats, const int32_t* a_stats, int32_t* new_idx, void* dst_nodes, int32_t* dst_w, int32_t* dst_a);

/* ---------------------------------------------------------------------------------------------------
 * Device-side sample counts (ABI v6).  Renderer::Render learns the number of samples that survive the early stop from a
 * blocking read-back (Renderer.cpp:128-135: boolean-mask indexing) and only then launches the grad pass.  The _dyn entry
 * points take the count where it is produced: n_dev points at a DEVICE int32 (the `total` of f2n_segment_scan); the kernels
 * process min(n_max, *n_dev + n_off) rows, n_max sizes grids / workspaces / plane strides (buffers must hold n_max rows).
 * The host can therefore queue a whole training step without waiting for the device.  n_dev == NULL: exactly the plain
 * entry point with n = n_max.  Results for the rows processed are identical to the plain calls.
 * ------------------------------------------------------------------------------------------------- */
int f2n_field_fwd_cached_dyn(void* stream, int n_max, const int32_t* n_dev, int n_cache, const int32_t* src_rows,
                             const void* x_cache_h, const void* mlp_params_h, float* out_feat_f32, float* out_f0, void* save_x_h);
int f2n_field_bwd_dyn(void* stream, int n_max, const int32_t* n_dev, int n_off, int n_volumes, const int32_t* prim_pool,
                      const int32_t* local_idx, const int32_t* local_size, const float* bias_pool, const float* level_scale,
                      const float* pts_warped, const int32_t* volume_idx, int vol_stride, const void* mlp_params_h,
                      const void* saved_x_h, const float* dfeat, float loss_scale, float* dparams_f32_scaled, void* grad_table_h,
                      int level_entries, int defer_reduce);
int f2n_shade_fwd_dyn(void* stream, int n_max, const int32_t* n_dev, const float* feat, const float* dirs, const float* app_emb,
                      const int32_t* sample_emb_idx, const void* mlp_params_h, float* rgb, void* save_x_h);
/* f2n_field_fwd_cached + f2n_shade_fwd in one launch for the surviving samples of the grad pass (Renderer.cpp:152-189): the
 * field network's outputs go straight into the colour network's input fragment, `feat` [n,16] is never written.  Outputs:
 * out_f0 [n] (density pre-activation, for compositing), save_field_x_h / save_shade_x_h [n,32] (h16 inputs of the two
 * networks, for f2n_field_bwd / f2n_shade_bwd; either may be NULL), rgb [n,3].  Bit-identical to the two separate calls.
 * src_rows NULL: row i of x_cache_h.  n_dev as above (may be NULL). */
int f2n_field_shade_fwd_dyn(void* stream, int n_max, const int32_t* n_dev, const int32_t* src_rows, const void* x_cache_h,
                            const void* field_params_h, const float* dirs, const float* app_emb, const int32_t* sample_emb_idx,
                            const void* color_params_h, float* out_f0, void* save_field_x_h, void* save_shade_x_h, float* rgb);
/* ... plus n_extra rows that only go through the field MLP (x_extra_h [n_extra,32] -> feat_extra fp32 [n_extra,16], inputs
 * saved to save_x_extra_h or NULL): f2n_field_fwd_cached for the edge (TV) samples of Renderer.cpp:159-166 riding in the same
 * launch. */
int f2n_field_shade_fwd_extra(void* stream, int n_max, const int32_t* n_dev, const int32_t* src_rows, const void* x_cache_h,
                              const void* field_params_h, const float* dirs, const float* app_emb, const int32_t* sample_emb_idx,
                              const void* color_params_h, float* out_f0, void* save_field_x_h, void* save_shade_x_h, float* rgb,
                              int n_extra, const void* x_extra_h, float* feat_extra, void* save_x_extra_h);
int f2n_shade_bwd_dyn(void* stream, int n_max, const int32_t* n_dev, const float* drgb, const int32_t* sample_emb_idx,
                      const void* mlp_params_h, const void* saved_x_h, float loss_scale, float* dfeat, float* dparams_f32_scaled,
                      float* dapp_emb, int n_emb, const float* df0, int defer_reduce);
/* defer_reduce != 0 (f2n_shade_bwd_dyn, f2n_field_bwd_dyn): the per-block partial sums of the parameter gradients (and of the
 * appearance-embedding gradient) stay in the library's workspace; ONE later f2n_reduce_deferred(stream) folds everything
 * that was deferred on this device into the destinations named at the time (at most four pending reductions; same stream
 * rule as the workspace itself).  Saves two dependent launches per training step. */
int f2n_reduce_deferred(void* stream);
/* Drops every reduction registered on this device and not yet folded: call it at the start of a backward pass (the host's
 * ZeroGrad does), so that a step that failed between a deferring launch and f2n_reduce_deferred cannot leak its partial sums
 * into the next step's gradients.  No launch. */
int f2n_deferred_reset(void);

/* ---------------------------------------------------------------------------------------------------
 * Renderer -- replaces the per-ray glue of Renderer::Render (Renderer/Renderer.cpp:105-208), i.e.
 * TruncExp (CustomOps.cpp:9-18), FlexOps::Sum/AccumulateSum (FlexOps.cu:5-93), CountValidPts /
 * FilterIdxBounds (Renderer.cu:8-50), GradientScaling (CustomOps.cu:68-80).  One sequential left-to-right
 * walk per ray, as in the reference (the add order is part of the parity contract).
 * ------------------------------------------------------------------------------------------------- */

/* No-grad pre-pass (Renderer.cpp:115-137): sigma = exp(f0-3), sec = sigma*dt, alpha = 1-exp(-sec),
 * T = exp(-exclusive_cumsum(sec)), w = T*alpha, mask = T > 1e-4.  f0 of sample i = f0[i*f0_stride].
 * kept[r] = number of samples of ray r with mask set. */
int f2n_early_stop(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride,
                   const float* dt, float* weights /*[N]*/, float* alphas /*[N]*/, int32_t* mask /*[N]*/,
                   int32_t* kept /*[R]*/);
/* f2n_early_stop followed by f2n_oct_mark_visit (the occupancy votes of PersSampler.cu:475-526 over the weights / alphas the
 * early stop has just produced) in ONE launch: same outputs, bit for bit.  anchors: node index of sample i at
 * anchors[i*anchor_stride + 1]; vote buffers as f2n_oct_mark_visit expects them. */
int f2n_early_stop_votes(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride, const float* dt,
                         float* weights, float* alphas, int32_t* mask, int32_t* kept, int n_nodes, const int32_t* anchors,
                         int anchor_stride, int32_t* w_adder, int32_t* a_adder, int32_t* mark, int32_t* visit_cnt);

/* Compaction of the surviving samples (Renderer.cpp:128-135: the five index-gathers).  Sample order is
 * preserved; new_start_end comes from f2n_segment_scan(kept). */
int f2n_compact_samples(void* stream, int n_rays, const int32_t* old_start_end, const int32_t* new_start_end,
                        const int32_t* mask, const float* pts, const float* dirs, const float* dt, const float* t,
                        const int32_t* anchors, float* o_pts, float* o_dirs, float* o_dt, float* o_t,
                        int32_t* o_anchors);
/* Same, and o_src[k] = index of surviving sample k in the uncompacted arrays (the row of the pre-pass feature
 * cache that f2n_field_fwd_cached reuses for it). */
int f2n_compact_samples_src(void* stream, int n_rays, const int32_t* old_start_end, const int32_t* new_start_end,
                            const int32_t* mask, const float* pts, const float* dirs, const float* dt, const float* t,
                            const int32_t* anchors, float* o_pts, float* o_dirs, float* o_dt, float* o_t,
                            int32_t* o_anchors, int32_t* o_src /*[M]*/,
                            int32_t* o_vol /*[M] or NULL: anchors[:,0] of the survivors as a unit-stride array*/,
                            const int32_t* ray_val /*[R] or NULL*/,
                            int32_t* o_ray_val /*[M] or NULL: ray_val of each survivor's ray = f2n_scatter_idx on the new bounds*/);

/* Compositing (Renderer.cpp:196-208): colors = sum w*c + T_last*bg, disparity = sum w/(t+.01),
 * depth = sum w*(t+.01) / (1 - T_last + 1e-4); weights [M] is also returned (RenderResult, Renderer.h:18-27).
 * f0 of sample i = f0[i * f0_stride]: stride 16 reads column 0 of the field output [M,16] as the reference does,
 * stride 1 reads a compact density array (f2n_field_fwd*'s out_f0: a quarter of the cache lines). */
int f2n_composite_fwd(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride,
                      const float* dt, const float* t, const float* rgb /*[M,3]*/, const float* bg /*[R,3]*/,
                      float* colors /*[R,3]*/, float* disparity /*[R]*/, float* depth /*[R]*/, float* weights /*[M]*/,
                      float* out_vars /*[R] or NULL: f2n_weight_var_fwd of the weights, same arithmetic, same launch*/);

/* Backward of the above through TruncExp; gradient scaling (CustomOps.cu:68-80) is applied to dsigma and
 * drgb when grad_scaling_progress < 1.  Any of dcolors/ddisparity/ddepth/dweights may be NULL (= zero).
 * Writes drgb [M,3] and d f0: df0[i * df0_stride] (stride 16 = column 0 of dfeat [M,16], other columns untouched;
 * stride 1 = a compact array that f2n_shade_bwd merges into the dfeat rows it writes anyway). */
int f2n_composite_bwd(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride,
                      const float* dt, const float* t, const float* rgb, const float* bg, const float* dcolors,
                      const float* ddisparity, const float* ddepth, const float* dweights,
                      float grad_scaling_progress, float* drgb, float* df0, int df0_stride,
                      const float* var_weights /*[M] or NULL*/, const float* dvars /*[R] or NULL*/);
/* var_weights + dvars (both or neither): the gradient f2n_weight_var_bwd would produce from them is added to dweights
 * inside this launch (same arithmetic), so a training step needs neither the extra launch nor the [M] buffer. */

/* WeightVarLoss forward/backward (CustomOps.cu:12-66). */
int f2n_weight_var_fwd(void* stream, int n_rays, const float* weights, const int32_t* pts_start_end, float* out_vars);
int f2n_weight_var_bwd(void* stream, int n_rays, const float* weights, const int32_t* pts_start_end,
                       const float* dvars, float* dweights);

/* Seam-level FlexOps (FlexOps.cu:5-93) for callers that keep the reference's op-by-op structure. */
int f2n_flex_sum_fwd(void* stream, int n_rays, int vec, const float* val, const int32_t* start_end, float* sum);
int f2n_flex_sum_bwd(void* stream, int n_rays, int vec, const float* dsum, const int32_t* start_end, float* dval);
int f2n_flex_acc_fwd(void* stream, int n_rays, int include_this, const float* val, const int32_t* start_end, float* sum);
int f2n_flex_acc_bwd(void* stream, int n_rays, int include_this, const float* dsum, const int32_t* start_end, float* dval);

/* ---------------------------------------------------------------------------------------------------
 * Ray generation -- replaces Dataset::Img2WorldRayFlex / CameraUndistort
 * (Dataset/Dataset.cu:13-152) and the colour / bounds gathers of Dataset::RandRaysData (Dataset/Dataset.cpp:275-298).
 * ------------------------------------------------------------------------------------------------- */
/* Img2WorldRayKernel (Dataset.cu:93-123): ij int32 [n,2] = (row, column); the half-pixel shift of :126 is applied
 * inside.  poses [C,3,4] row-major c2w, intri [C,3,3], dist_params [C,4] = (k1,k2,p1,p2); Newton undistortion with
 * central differences, <= 100 iterations (:30-72).  rays_d is NOT normalised (GetSamples does that, :319). */
int f2n_img2world_rays(void* stream, int n_rays, const float* poses, const float* intri, const float* dist_params,
                       const int32_t* cam_indices /*[n]*/, const int32_t* ij /*[n,2]*/, float* rays_o /*[n,3]*/,
                       float* rays_d /*[n,3]*/);
/* gt_colors[r] = images[cam][i][j][0:3] (images fp32 [C,H,W,3], resident in HBM) and bounds[r] = cam_bounds[cam]
 * ([C,2]); either output may be NULL. */
int f2n_gather_pixels(void* stream, int n_rays, int height, int width, const float* images, const float* cam_bounds,
                      const int32_t* cam_indices, const int32_t* ij, float* gt_colors /*[n,3]*/, float* bounds /*[n,2]*/);
/* Dataset::RandRaysData (Dataset/Dataset.cpp:275-298) in one launch: u01 [R,3] uniforms in [0,1) pick an image of image_set
 * (n_set entries) and a pixel (i = floor(u1 * height), j = floor(u2 * width)); then f2n_img2world_rays and f2n_gather_pixels for
 * those draws.  Outputs: cam_indices [R], ij [R,2], rays_o / rays_d [R,3], gt_colors [R,3] (or NULL), bounds [R,2]. */
int f2n_draw_ray_batch(void* stream, int n_rays, const float* u01, const int32_t* image_set, int n_set, int height, int width,
                       const float* poses, const float* intri, const float* dist_params, const float* images /*or NULL*/,
                       const float* cam_bounds, int32_t* cam_indices, int32_t* ij, float* rays_o, float* rays_d,
                       float* gt_colors /*or NULL*/, float* bounds);
/* The same with the three uniforms of ray r drawn by the launch itself: Philox4x32-10, counter = (r, 0, seq), key = seed ^ purpose
 * (host/KeyedDraws.h): batch `seq` is the same batch whenever, however often and with whatever else in between it is drawn. */
int f2n_draw_ray_batch_keyed(void* stream, int n_rays, uint64_t key, uint64_t seq, const int32_t* image_set, int n_set, int height,
                             int width, const float* poses, const float* intri, const float* dist_params, const float* images /*or NULL*/,
                             const float* cam_bounds, int32_t* cam_indices, int32_t* ij, float* rays_o, float* rays_d,
                             float* gt_colors /*or NULL*/, float* bounds);

/* ---------------------------------------------------------------------------------------------------
 * Optimiser -- replaces torch::optim::Adam::step over the groups of Hash3DAnchored::OptimParamGroups
 * (Field/Hash3DAnchored.cpp:124-150), SHShader (Shader/SHShader.cpp:44-56), Renderer (Renderer.cpp:238-258):
 * beta = (0.9, 0.99), eps = 1e-15, L2 weight decay added to the gradient (torch Adam semantics).
 * beta1 / beta2 are DOUBLES, as in torch::optim::AdamOptions (`opt->betas() = {0.9, 0.99}`, Hash3DAnchored.cpp:131): LibTorch forms
 * 1 - beta and the bias corrections 1 - beta^step in double and only then narrows them to the tensors' float, so (float)(1 - 0.9) =
 * 0.1f -- a float beta would give 1 - 0.9f = 0.100000024 and a second-moment coefficient that is 9e-7 off (ABI v12; up to v11 the
 * betas were floats).  f2n_adam_coefficients returns the nine scalars every Adam kernel of this library is launched with:
 * out[0..8] = -step_size's magnitude lr / (1 - beta1^step), sqrt(1 - beta2^step), beta1, beta2, 1 - beta1, 1 - beta2, eps,
 * weight_decay, grad_scale.  Host function, no device needed (tests pin it against LibTorch's own scalars).
 * ------------------------------------------------------------------------------------------------- */
int f2n_adam_coefficients(int step, float lr, double beta1, double beta2, float eps, float weight_decay, float grad_scale,
                          float* out9 /*host*/);
/* fp32 gradient (grad * grad_scale is the true gradient); optionally refreshes an h16 working copy.
 * grad_round_h16 != 0 reproduces the two binary16 roundings the reference applies to MLP parameter gradients:
 * g = f16(f16(grad) * grad_scale) (tcnn param-precision output while loss-scaled, Field/TCNNWP.cpp:214-215, then
 * autograd's cast of the unscaled gradient to the f16 dtype of the Function input, :111,:242). */
int f2n_adam_step(void* stream, int n, float* param, float* grad, float grad_scale, int grad_round_h16,
                  float* exp_avg, float* exp_avg_sq, int step, float lr, double beta1, double beta2, float eps,
                  float weight_decay, void* param_h_or_null, int zero_grad /* clear grad after use (also when skipped) */,
                  const int32_t* skip_flag /*device, or NULL*/);
/* The small parameter groups of an iteration in ONE launch: finiteness flags (TCNNWP.cpp:234-240; layout of
 * f2n_nonfinite_flags: flags[0] = group 0 has a non-finite gradient, flags[1] = group 1, flags[2] = either; only groups
 * 0 and 1 may ask for the check) followed by f2n_adam_step on every group, predicated on flags[2] and on *skip_flag.
 * `groups` is a HOST array of 1..4 descriptors; element-wise arithmetic identical to f2n_adam_step. */
typedef struct F2nAdamGroup {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  void* param_h;      /* h16 working copy to refresh, or NULL */
  int n;
  float grad_scale;
  float weight_decay;
  int grad_round_h16;
  int check_finite;
} F2nAdamGroup;
int f2n_adam_small_groups(void* stream, int n_groups, const F2nAdamGroup* groups, int step, float lr, double beta1,
                          double beta2, float eps, int zero_grad, int32_t* flags /*device [3] or NULL*/,
                          const int32_t* skip_flag /*device, or NULL*/);
/* Every group of an iteration in one launch: `groups` (HOST array of 0..4 descriptors; check_finite is ignored) as in
 * f2n_adam_small_groups and the h16-gradient table as in f2n_adam_step_h16grad, all predicated on *skip_flag (device, or
 * NULL) -- typically flags + 2 of a preceding f2n_nonfinite_flags.  Element-wise arithmetic identical to the separate
 * steps; no block-wide dependency inside (the flags-then-step single-block launch took 21 us in front of the table pass). */
int f2n_adam_fused(void* stream, int n_groups, const F2nAdamGroup* groups, int n_table, float* table_param, void* table_grad_h,
                   float table_grad_scale, float* table_exp_avg, float* table_exp_avg_sq, void* table_param_h, int step, float lr,
                   double beta1, double beta2, float eps, int zero_grad, const int32_t* skip_flag /*device, or NULL*/);
/* ---------------------------------------------------------------------------------------------------
 * The TAIL of a training step in one call (ABI v13, round 6) -- f2n_field_bwd_dyn + f2n_reduce_deferred + f2n_nonfinite_flags_ex +
 * f2n_adam_fused, re-ordered the way their data allows (ExpRunner.cpp:127-137: backward, finiteness check, optimizer->step(),
 * zero_grad()):
 *   stream:       field-MLP backward | hash_bin (scatter producers) ............... | owners: slice sums -> THE TABLE'S ADAM on that slice
 *   tail_stream:                     | reduce_deferred -> flags -> Adam of the small groups |  (joined in front of the owners)
 * The finiteness flags depend on the two MLPs' parameter gradients only (TCNNWP.cpp:234-240), which are complete once the
 * field-MLP backward kernel has run -- not on the scatter that follows it; and an owner block of the binned scatter holds the
 * finished gradient sums of its 4096-entry table slice in LDS, so it steps that slice's parameters itself (fp16->fp32 widening,
 * /128, Adam, fp32->fp16 refresh, the gradient table left zero: f2n_adam_step_h16grad's arithmetic, element for element)
 * instead of writing the sums to the gradient table for a 267 MB streaming pass behind the step's last kernel.  Parameters,
 * moments, working copies, flags and the (zero) gradient buffers end bit-identical to the four separate calls
 * (tests/test_gpu_parity.py::test_step_tail_equals_separate_launches).
 * tail_stream NULL: everything on `stream` (same results).  A batch that does not take the binned scatter (f2n_hash_bwd's
 * small-batch path), a table whose active prefix is not whole 4096-entry slices, or a bucket hook on this gradient table
 * (f2n_set_scatter_buckets_for: data-parallel runs all-reduce the gradient first) steps the table with the ordinary launch
 * behind the scatter.  `tail->groups`: HOST array as for f2n_adam_fused; every update is predicated on flags[2]. */
typedef struct F2nStepTail {
  int n_flags_a;                 /* f2n_nonfinite_flags_ex: field-MLP gradient ... */
  const float* flags_grad_a;
  int n_flags_b;                 /* ... colour-MLP gradient */
  const float* flags_grad_b;
  int32_t* flags;                /* device int32[3] */
  int32_t* flags_mirror;         /* mapped host words, or NULL */
  int n_groups;                  /* the small fp32 groups (0..4) */
  const F2nAdamGroup* groups;
  int n_table;                   /* halves of the table group's active prefix */
  float* table_param;
  float* table_exp_avg;
  float* table_exp_avg_sq;
  void* table_param_h;
  float table_grad_scale;
  int step;
  float lr;
  double beta1, beta2;
  float eps;
  /* Data-parallel hosts (the gradients travel before anything is stepped):
   * after_reduce != NULL: called on the calling host thread once the deferred reductions have been QUEUED on the tail stream (or on
   *   `stream` when there is none) -- the small gradient buffers are complete behind that point of that stream -- and before the flag
   *   kernel is: the host orders its exchange of the small buffers in between (event on that stream -> communicator stream -> event ->
   *   back), so that it travels beside the scatter's producers instead of behind the step;
   * leave_table_to_caller != 0: everything BUT the table's Adam -- the owners write their sums to the gradient table as
   *   f2n_field_bwd_dyn's do (a bucket hook on the table reports as usual), `stream` is left waiting for the tail chain, and the caller
   *   steps the table itself (f2n_adam_fused with no small groups, skip flag = flags + 2) once ITS exchange is through. */
  void (*after_reduce)(void* user, void* chain_stream);
  void* after_reduce_user;
  int leave_table_to_caller;
} F2nStepTail;
int f2n_field_bwd_step_tail(void* stream, void* tail_stream /* or NULL */, int n_max, const int32_t* n_dev, int n_off, int n_volumes,
                            const int32_t* prim_pool, const int32_t* local_idx, const int32_t* local_size, const float* bias_pool,
                            const float* level_scale, const float* pts_warped, const int32_t* volume_idx, int vol_stride,
                            const void* mlp_params_h, const void* saved_x_h, const float* dfeat, float loss_scale,
                            float* dparams_f32_scaled, void* grad_table_h, int level_entries, const F2nStepTail* tail,
                            int* table_stepped_by_owners /* out, or NULL: 1 when the owners applied the table's Adam */);
/* h16 gradient table produced by f2n_hash_bwd / f2n_field_bwd (true gradient = float(grad_h) * grad_scale,
 * grad_scale = 1/128): fuses the fp16->fp32 cast, the /128 (Hash3DAnchored.cu:232), Adam, the fp32->fp16
 * refresh of the table (Hash3DAnchored.cu:186) and the re-zeroing of the gradient (:222) in one pass. */
int f2n_adam_step_h16grad(void* stream, int n, float* param, void* grad_h, float grad_scale, float* exp_avg,
                          float* exp_avg_sq, int step, float lr, double beta1, double beta2, float eps,
                          float weight_decay, void* param_h, int zero_grad, const int32_t* skip_flag /*device, or NULL*/);
/* skip_flag (both steps): when *skip_flag != 0 on the device the update is dropped -- parameters and moments stay,
 * an h16 gradient table is still cleared when zero_grad is set.  This is the `continue` of ExpRunner.cpp:131-134
 * without a host round trip between the finiteness check and the optimiser. */

/* ---------------------------------------------------------------------------------------------------
 * Training loss -- replaces the ~30 ATen element-wise / reduction ops (and their autograd mirror images) of
 * ExpRunner::Train (ExpRunner.cpp:95-120) by one pass that returns the loss terms AND their gradients:
 *   color = mean sqrt((pred-gt)^2 + 1e-4), disp = mean disparity^2, var = mean sqrt(sampled_var + 1e-2),
 *   tv = mean (edge_feats[:,0,:] - edge_feats[:,1,:])^2, loss = color + var_w*var + disp_w*disp + tv_w*tv.
 * out_losses [8] = {loss, color, var, disp, tv, mse, 0, 0}.  Gradients of `loss`: dcolors [R,3], ddisparity [R],
 * dvar [R], dedge_feats [E,2,feat_dim]; any gradient pointer may be NULL, and disparity / sampled_var /
 * edge_feats may be NULL (term = 0), as on the reference's no-sample early return (Renderer.cpp:83-97).
 * ------------------------------------------------------------------------------------------------- */
int f2n_train_loss(void* stream, int n_rays, const float* pred_colors /*[R,3]*/, const float* gt_colors /*[R,3]*/,
                   const float* disparity /*[R]*/, const float* sampled_var /*[R]*/, int n_edge, int feat_dim,
                   const float* edge_feats /*[E,2,feat_dim]*/, float var_w, float disp_w, float tv_w,
                   float* out_losses /*[8]*/, float* dcolors, float* ddisparity, float* dvar, float* dedge_feats);

/* f2n_composite_fwd -> f2n_train_loss -> f2n_composite_bwd in ONE launch (the streaming training step; round 4): every ray is
 * walked forward, its loss terms and their gradients are formed in registers (they are element-wise per ray: ExpRunner.cpp:95-120),
 * and the same row walks it backward with the totals still at hand.  Outputs as the three calls': colors [R,3], weights [M], drgb
 * [M,3], df0 (stride df0_stride) -- bit-identical -- plus dedge_feats (TV gradient) and out_losses [8] = {loss, colour, var,
 * disparity, tv, mse, 0, 0}: zeroed by the launch and completed by the reduction of its per-block partial sums -- at once
 * (defer_reduce = 0) or by the step's f2n_reduce_deferred (defer_reduce = 1).  The reported loss VALUES sum pre-scaled terms
 * (mean = sum of x / n instead of (sum of x) / n): equal to f2n_train_loss's to rounding, not bit for bit. */
int f2n_composite_train(void* stream, int n_rays, const int32_t* pts_start_end, const float* f0, int f0_stride, const float* dt,
                        const float* t, const float* rgb, const float* bg, const float* gt_colors, float var_w, float disp_w, float tv_w,
                        float grad_scaling_progress, int n_edge, int feat_dim, const float* edge_feats /*[E,2,feat_dim] or NULL*/,
                        float* dedge_feats /*or NULL*/, float* colors, float* weights, float* drgb, float* df0, int df0_stride,
                        float* out_losses /*[8]*/, int defer_reduce);

/* Gradient finiteness check of the two MLPs (Field/TCNNWP.cpp:234-240), device side:
 * flags[0] = a has a non-finite value, flags[1] = b has one, flags[2] = either (the optimiser's skip_flag). */
int f2n_nonfinite_flags(void* stream, int n_a, const float* a, int n_b, const float* b, int32_t* flags /*[3]*/);
/* The same, with the three flags also written to `mirror` (DEVICE pointer to MAPPED HOST memory, or NULL): the host-side
 * reaction of TCNNWP.cpp:236-240 (halve the loss scale, drop the iteration) reads them there behind an event, without a copy
 * launch behind the optimiser. */
int f2n_nonfinite_flags_ex(void* stream, int n_a, const float* a, int n_b, const float* b, int32_t* flags /*[3]*/,
                           int32_t* mirror /*mapped host [3] or NULL*/);

#ifdef __cplusplus
}
#endif
#endif /* F2N_ABI_H */
