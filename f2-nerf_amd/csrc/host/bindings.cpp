// pybind11 surface of the C++/LibTorch host layer (used by bench.py, __graft_entry__.smoke() and the
// end-to-end tests).  Thin: object lifetime + tensor hand-over only.
#include <torch/extension.h>

#include "DataParallel.h"
#include "ExpRunner.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>

using namespace f2n;

namespace {

Hash3DAnchored* FieldOf(ExpRunner& r) { return static_cast<Hash3DAnchored*>(r.renderer_->scene_field_.get()); }
SHShader* ShaderOf(ExpRunner& r) { return static_cast<SHShader*>(r.renderer_->shader_.get()); }
PersSampler* SamplerOf(ExpRunner& r) { return static_cast<PersSampler*>(r.renderer_->pts_sampler_.get()); }

py::dict StatsToDict(const TrainStats& s) {
  py::dict d;
  d["loss"] = s.loss;
  d["mse"] = s.mse;
  d["n_rays"] = s.n_rays;
  d["n_samples"] = s.n_samples;
  d["n_meaningful"] = s.n_meaningful;
  d["skipped_nan"] = s.skipped_nan;
  return d;
}

}  // namespace

// the ABI version this host layer was compiled against (include/f2n_abi.h); a kernel library of another version next to it
// means one of the two was not rebuilt -- calls would pass the wrong argument lists (observed once: a memory fault)
#define F2N_HOST_EXPECTS_ABI 13

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "f2-nerf hot path: C++/LibTorch host layer over libf2n_hip.so";
  TORCH_CHECK(f2n_abi_version() == F2N_HOST_EXPECTS_ABI, "libf2n_hip reports ABI version ", f2n_abi_version(), ", this host extension was built for ",
              F2N_HOST_EXPECTS_ABI, ": rebuild both (python -c 'import __graft_entry__ as g; g.build()')");
  m.attr("abi_version") = F2N_HOST_EXPECTS_ABI;
  auto bounded = [](const BoundedRays& r) { return std::vector<Tensor>{r.origins, r.dirs, r.bounds}; };
  auto ray_data = [](const std::tuple<BoundedRays, Tensor, Tensor>& t) {
    const auto& r = std::get<0>(t);
    return std::vector<Tensor>{r.origins, r.dirs, r.bounds, std::get<1>(t), std::get<2>(t)};  // + gt colours, image index
  };
  // octree / warp construction from cameras: byte blobs in the reference's checkpoint layout
  m.def("build_octree",
        [](const Tensor& c2w, const Tensor& intri, const Tensor& bounds, int max_depth, float bbox_side_len, float split_dist_thres,
           int n_rand_pts) {
          OctreeBuildResult r = BuildPersOctree(c2w, intri, bounds, max_depth, bbox_side_len, split_dist_thres, n_rand_pts);
          auto blob = [](const void* p, size_t bytes) {
            Tensor t = torch::empty({(int64_t) bytes}, CpuU8());
            if (bytes) std::memcpy(t.data_ptr(), p, bytes);
            return t;
          };
          py::dict d;
          d["tree_nodes"] = blob(r.nodes.data(), r.nodes.size() * sizeof(TreeNode));
          d["pers_trans"] = blob(r.trans.data(), r.trans.size() * sizeof(TransInfo));
          d["edge_pool"] = blob(r.edges.data(), r.edges.size() * sizeof(EdgePool));
          d["n_volumes"] = (int) r.trans.size();
          return d;
        },
        py::arg("c2w"), py::arg("intri"), py::arg("bounds"), py::arg("max_depth"), py::arg("bbox_side_len"),
        py::arg("split_dist_thres"), py::arg("n_rand_pts") = 32 * 32 * 32);
  m.def("construct_trans", [](const Tensor& rand_pts, const Tensor& c2w_vis, const Tensor& intri0, const Tensor& center, int first_cam) {
    TransInfo t = ConstructTrans(rand_pts, c2w_vis, intri0, center, first_cam);
    Tensor out = torch::empty({(int64_t) sizeof(TransInfo)}, CpuU8());
    std::memcpy(out.data_ptr(), &t, sizeof(TransInfo));
    return out;
  });
  m.def("construct_edge_pool", [](const Tensor& tree_nodes_bytes) {
    Tensor b = tree_nodes_bytes.to(torch::kCPU).contiguous();
    std::vector<TreeNode> nodes(b.numel() / sizeof(TreeNode));
    std::memcpy((void*) nodes.data(), b.data_ptr(), nodes.size() * sizeof(TreeNode));
    auto e = ConstructEdgePool(nodes);
    Tensor out = torch::empty({(int64_t) (e.size() * sizeof(EdgePool))}, CpuU8());
    if (!e.empty()) std::memcpy(out.data_ptr(), e.data(), e.size() * sizeof(EdgePool));
    return out;
  });
  py::class_<Dataset>(m, "Dataset")
      .def(py::init<const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, int, int, const std::vector<int>&,
                    const std::vector<int>&, const std::vector<int>&>(),
           py::arg("poses"), py::arg("intri"), py::arg("dist_params"), py::arg("bounds"), py::arg("images"), py::arg("height"),
           py::arg("width"), py::arg("train_set"), py::arg("test_set"), py::arg("val_set"))
      .def("img2world_ray_flex", [](Dataset& d, const Tensor& cam, const Tensor& ij) { auto r = d.Img2WorldRayFlex(cam, ij); return std::vector<Tensor>{r.origins, r.dirs}; })
      .def("gather_colors", &Dataset::GatherColors)
      .def("rays_of_camera", [bounded](Dataset& d, int idx) { return bounded(d.RaysOfCamera(idx)); })
      .def("rays_from_pose", [bounded](Dataset& d, const Tensor& pose, int reso) { return bounded(d.RaysFromPose(pose, reso)); },
           py::arg("pose"), py::arg("reso_level") = 1)
      .def("rand_rays_from_pose", [bounded](Dataset& d, int n, const Tensor& pose) { return bounded(d.RandRaysFromPose(n, pose)); })
      .def("rand_rays_whole_space", [bounded](Dataset& d, int n) { return bounded(d.RandRaysWholeSpace(n)); })
      .def("rand_rays_data", [ray_data](Dataset& d, int n, int sets, int64_t seq) { return ray_data(d.RandRaysData(n, sets, seq)); },
           py::arg("batch_size"), py::arg("sets") = DATA_TRAIN_SET, py::arg("seq") = -1)
      .def("rand_rays_data_of_camera", [ray_data](Dataset& d, int idx, int n) { return ray_data(d.RandRaysDataOfCamera(idx, n)); })
      .def_static("pose_interpolate", &PoseInterpolate)
      .def_readonly("last_cam_indices", &Dataset::last_cam_indices_)
      .def_readonly("last_ij", &Dataset::last_ij_)
      .def_readonly("n_images", &Dataset::n_images_)
      .def_readonly("height", &Dataset::height_)
      .def_readonly("width", &Dataset::width_);
  m.def("dp_set_table_buckets", [](int n) { DataParallel::table_buckets = n; });  // table all-reduce buckets of the next attach (A/B)
  // data-parallel replicas draw stream `rank` of every keyed purpose (KeyedDraws.h; the native attach sets it itself)
  m.def("dp_set_replica", [](int rank) { KeyedUniforms::SetReplica(rank); });
  m.def("keyed_draw_key", [](uint64_t seed, uint64_t purpose, int rank) {  // the Philox key of a purpose on a rank (no device needed)
    return KeyedUniforms::KeyOf(seed, purpose, KeyedUniforms::SaltOf(rank));
  });
  m.def("dp_new_unique_id", []() {  // rank 0: the id every rank passes to attach_data_parallel
    auto id = DataParallel::NewUniqueId();
    return py::bytes(reinterpret_cast<const char*>(id.data()), id.size());
  });
  // the schedules of ExpRunner::Train as a pure function (no device needed): tests pin them against the reference's own code
  m.def("schedule_at", [](float init_fineness, int fineness_decay_end, float lr, float lr_alpha, float lr_warm_up_end, int end_iter,
                          float gs_start, float gs_end, float var_w, int var_start, int var_end, int iter) {
    ExpRunner::ScheduleParams p{init_fineness, fineness_decay_end, lr, lr_alpha, lr_warm_up_end, end_iter, gs_start, gs_end, var_w,
                                var_start, var_end};
    auto v = ExpRunner::ScheduleAt(p, iter);
    return std::vector<float>{v.fineness, v.lr, v.gradient_scaling_progress, v.var_loss_weight};
  });
  // the step / exchange interleaving of ExpRunner::TrainStep on its own (no device): tests drive it with recording callbacks
  py::class_<GradSyncPipeline>(m, "GradSyncPipeline")
      .def(py::init<>())
      .def_readwrite("pipelined", &GradSyncPipeline::pipelined)
      .def("set_blocking", [](GradSyncPipeline& p, py::function f) { p.blocking = [f]() { f(); }; })
      .def("set_begin_end", [](GradSyncPipeline& p, py::function b, py::function e) { p.begin = [b]() { b(); }; p.end = [e]() { e(); }; })
      .def("set_apply", [](GradSyncPipeline& p, py::function f) { p.apply = [f](bool a, float lr) { f(a, lr); }; })
      .def("set_defer_flags", [](GradSyncPipeline& p, py::function f) { p.defer_flags = [f]() { f(); }; })
      .def("set_bucket", [](GradSyncPipeline& p, py::function f) { p.bucket = [f](int b, int n) { f(b, n); }; })
      .def("bucket_ready", &GradSyncPipeline::BucketReady)
      .def("set_small_exchange", [](GradSyncPipeline& p, py::function f) { p.small_exchange = [f](void*) { f(); }; })
      .def("small_grads_ready", [](GradSyncPipeline& p) { p.SmallGradsReady(nullptr); })
      .def_readwrite("small_first", &GradSyncPipeline::small_first)
      .def_property_readonly("small_sent", &GradSyncPipeline::small_sent)
      .def_property_readonly("buckets_sent", &GradSyncPipeline::buckets_sent)
      .def("installed", &GradSyncPipeline::Installed)
      .def("pending", &GradSyncPipeline::Pending)
      .def("begin_step", [](GradSyncPipeline& p, bool apply_optimizer, py::function presample) { p.BeginStep(apply_optimizer, [presample]() { presample(); }); })
      .def("gradients_ready", &GradSyncPipeline::GradientsReady)
      .def("finish_pending_step", &GradSyncPipeline::FinishPendingStep);
  py::class_<ExpRunner>(m, "ExpRunner")
      .def(py::init<const std::map<std::string, std::string>&, int>(), py::arg("flat_config"), py::arg("n_images"))
      .def("load_states", &ExpRunner::LoadStates)
      .def("states", &ExpRunner::States)
      .def("aux_states", &ExpRunner::AuxStates)
      .def("load_aux_states", &ExpRunner::LoadAuxStates)
      .def("train_step",
           [](ExpRunner& r, const Tensor& ro, const Tensor& rd, const Tensor& b, const Tensor& gt, const Tensor& emb, bool apply,
              const std::optional<Tensor>& nro, const std::optional<Tensor>& nrd, const std::optional<Tensor>& nb,
              const std::optional<Tensor>& n2ro, const std::optional<Tensor>& n2rd) {
             TrainStats s;
             {
               py::gil_scoped_release no_gil;  // hooks re-acquire the GIL themselves
               s = r.TrainStep(ro, rd, b, gt, emb, apply, nro.value_or(Tensor()), nrd.value_or(Tensor()), nb.value_or(Tensor()),
                               n2ro.value_or(Tensor()), n2rd.value_or(Tensor()));
             }
             return StatsToDict(s);
           },
           py::arg("rays_o"), py::arg("rays_d"), py::arg("bounds"), py::arg("gt_colors"), py::arg("emb_idx"),
           py::arg("apply_optimizer") = true, py::arg("next_rays_o") = py::none(), py::arg("next_rays_d") = py::none(),
           py::arg("next_bounds") = py::none(), py::arg("next2_rays_o") = py::none(), py::arg("next2_rays_d") = py::none())
      .def("train_step_autograd",
           [](ExpRunner& r, const Tensor& ro, const Tensor& rd, const Tensor& b, const Tensor& gt, const Tensor& emb, bool apply) {
             TrainStats s;
             {
               py::gil_scoped_release no_gil;  // the autograd engine must not be entered while holding the GIL
               s = r.TrainStepAutograd(ro, rd, b, gt, emb, apply);
             }
             return StatsToDict(s);
           },
           py::arg("rays_o"), py::arg("rays_d"), py::arg("bounds"), py::arg("gt_colors"), py::arg("emb_idx"),
           py::arg("apply_optimizer") = true)
      .def_readwrite("end_iter", &ExpRunner::end_iter_)
      .def_readwrite("forward_render", &ExpRunner::forward_render_)
      .def_readwrite("render_chunk_rays", &ExpRunner::render_chunk_rays_)
      .def("render_rays", &ExpRunner::RenderRays)
      .def("render_whole_image", &ExpRunner::RenderWholeImage)
      .def("test_image_psnr", &ExpRunner::TestImagePSNR)
      .def("test_images", &ExpRunner::TestImages)
      .def("visualize_image", &ExpRunner::VisualizeImage)
      .def("render_path_frame", &ExpRunner::RenderPathFrame, py::arg("dataset"), py::arg("pose"), py::arg("res_level") = 1)
      .def("render_path",
           [](ExpRunner& r, Dataset& ds, const Tensor& poses, py::function sink, int res_level) {
             r.RenderPath(ds, poses, [sink](int i, const Tensor& img) { sink(i, img); }, res_level);
           },
           py::arg("dataset"), py::arg("render_poses"), py::arg("sink"), py::arg("res_level") = 1)
      .def("save_checkpoint", &ExpRunner::SaveCheckpoint)
      .def("load_checkpoint", &ExpRunner::LoadCheckpoint)
      .def("train",
           [](ExpRunner& r, Dataset& ds, int until_iter, int sets) {
             int n;
             {
               py::gil_scoped_release no_gil;
               n = r.Train(ds, until_iter, sets);
             }
             py::dict d = StatsToDict(r.last_train_stats_);
             d["iterations"] = n;
             d["total_meaningful"] = r.last_train_meaningful_;
             d["total_marched"] = r.last_train_marched_;
             d["total_rays"] = r.last_train_rays_;
             return d;
           },
           py::arg("dataset"), py::arg("until_iter") = -1, py::arg("sets") = DATA_TRAIN_SET)
      .def("render_train",
           [](ExpRunner& r, const Tensor& ro, const Tensor& rd, const Tensor& b, const Tensor& emb) {
             r.global_data_pool_->mode_ = RunningMode::TRAIN;
             auto rr = r.renderer_->Render(ro, rd, b, emb);
             py::dict d;
             d["colors"] = rr.colors; d["first_oct_dis"] = rr.first_oct_dis; d["disparity"] = rr.disparity;
             d["edge_feats"] = rr.edge_feats; d["depth"] = rr.depth; d["weights"] = rr.weights;
             d["idx_start_end"] = rr.idx_start_end;
             return d;
           })
      .def("get_samples",
           [](ExpRunner& r, const Tensor& ro, const Tensor& rd, const Tensor& b) {
             auto s = r.renderer_->pts_sampler_->GetSamples(ro, rd, b);
             py::dict d;
             d["pts"] = s.pts; d["dirs"] = s.dirs; d["dt"] = s.dt; d["t"] = s.t; d["anchors"] = s.anchors;
             d["pts_idx_bounds"] = s.pts_idx_bounds; d["first_oct_dis"] = s.first_oct_dis;
             return d;
           })
      .def("anchored_query", [](ExpRunner& r, const Tensor& pts, const Tensor& anchors) { return r.renderer_->scene_field_->AnchoredQuery(pts, anchors); })
      .def("shader_query", [](ExpRunner& r, const Tensor& feats, const Tensor& dirs) { return r.renderer_->shader_->Query(feats, dirs); })
      .def("zero_grad", [](ExpRunner& r) { r.renderer_->ZeroGrad(); })
      .def("optim_step", [](ExpRunner& r) { r.OptimStep(nullptr); })
      .def("grads",
           [](ExpRunner& r) {
             py::dict d;
             d["feat_pool"] = FieldOf(r)->TableGradUnscaled();
             d["field_mlp"] = FieldOf(r)->mlp_->GradUnscaled();
             d["color_mlp"] = ShaderOf(r)->mlp_->GradUnscaled();
             d["app_emb"] = r.renderer_->app_emb_grad_.clone();
             return d;
           })
      .def("grad_buffers",  // raw buffers for the data-parallel all-reduce (RCCL): hash table (f16 x128), MLPs, app_emb
           [](ExpRunner& r) {
             return std::vector<Tensor>{FieldOf(r)->grad_h_, FieldOf(r)->mlp_->grad_scaled_, ShaderOf(r)->mlp_->grad_scaled_,
                                        r.renderer_->app_emb_grad_};
           })
      .def("flatten_small_grads", &ExpRunner::FlattenSmallGrads)
      .def("occupancy_buffers",
           [](ExpRunner& r) {
             auto& o = *SamplerOf(r)->pers_octree_;
             return std::vector<Tensor>{o.tree_weight_stats_, o.tree_alpha_stats_, o.tree_visit_cnt_};
           })
      .def("set_grad_sync_hook", [](ExpRunner& r, py::function f) { r.sync_.blocking = [f]() { py::gil_scoped_acquire g; f(); }; })
      .def("set_pipelined_grad_sync",
           [](ExpRunner& r, py::function begin, py::function end) {
             r.sync_.begin = [begin]() { py::gil_scoped_acquire g; begin(); };
             r.sync_.end = [end]() { py::gil_scoped_acquire g; end(); };
             r.sync_.pipelined = true;
           })
      .def("flush", [](ExpRunner& r) { py::gil_scoped_release no_gil; r.FinishPending(); })
      .def("attach_data_parallel",  // native RCCL exchanges, issued from C++ inside TrainStep (DataParallel.h); collective
           [](ExpRunner& r, int rank, int world, const py::bytes& unique_id, bool overlap, bool hooks_for_one_rank) {
             std::string s = unique_id;
             auto dp = std::make_shared<DataParallel>();
             {
               py::gil_scoped_release no_gil;
               dp->Attach(&r, rank, world, std::vector<uint8_t>(s.begin(), s.end()), overlap, hooks_for_one_rank);
             }
             r.data_parallel_ = dp;
           },
           py::arg("rank"), py::arg("world"), py::arg("unique_id"), py::arg("overlap") = true, py::arg("hooks_for_one_rank") = false)
      .def("dp_bucket_callbacks",  // table-gradient ranges the scatter reported while it ran (bucketed exchange, DataParallel.h)
           [](ExpRunner& r) { return r.data_parallel_ ? std::static_pointer_cast<DataParallel>(r.data_parallel_)->BucketCallbacks() : (int64_t) 0; })
      .def("dp_small_exchanges_early",  // steps whose small gradient buffers were exchanged beside the scatter (DataParallel::SmallGradsExchange)
           [](ExpRunner& r) { return r.data_parallel_ ? std::static_pointer_cast<DataParallel>(r.data_parallel_)->SmallExchangesEarly() : (int64_t) 0; })
      .def("dp_enable_timing",  // bracket every gradient exchange / every wait for it with timing events (DataParallel::EnableTiming)
           [](ExpRunner& r, bool on) { if (r.data_parallel_) std::static_pointer_cast<DataParallel>(r.data_parallel_)->EnableTiming(on); })
      .def("dp_collect_timing",  // [exchanges, exchange ms total, waits, exposed wait ms total] since the last call (synchronises)
           [](ExpRunner& r) { return r.data_parallel_ ? std::static_pointer_cast<DataParallel>(r.data_parallel_)->CollectTiming() : std::vector<double>{0, 0, 0, 0}; })
      .def("dp_comm_ranks",  // ranks of the native RCCL communicator as RCCL reports them (0: none attached)
           [](ExpRunner& r) { return r.data_parallel_ ? std::static_pointer_cast<DataParallel>(r.data_parallel_)->CommRanks() : 0; })
      .def("set_occupancy_sync_hook",  // all-reduce(MAX) of the per-node votes so that every replica prunes identically
           [](ExpRunner& r, py::function f) {
             SamplerOf(r)->occupancy_sync_hook_ = [f](Tensor occ) {
               py::gil_scoped_acquire g;
               f(occ);
             };
           })
      .def_static("host_profile",  // on=True: start (cleared); on=False: stop and return {region: (entries, seconds)} of the host thread
                  [](bool on) {
                    auto& hp = HostProf::Get();
                    py::dict d;
                    for (auto& kv : hp.acc) d[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
                    hp.acc.clear();
                    hp.on = on;
                    return d;
                  })
#if F2N_DEBUG_BUILD
      .def_static("debug_side_delay",  // spin kernels (us) in front of every period-th speculative begin / completion / step: a race amplifier
                  [](int begin_us, int complete_us, int main_us, int period, unsigned pollute) {
                    Renderer::SetDebugSideDelay(begin_us, complete_us, main_us, period, pollute);
                  },
                  py::arg("begin_us"), py::arg("complete_us"), py::arg("main_us"), py::arg("period"), py::arg("pollute") = 0u)
#endif
      .def_static("enable_kernel_timing", [](const std::vector<std::string>& names) { KernelTimers::Get().Enable(names); })
      .def_static("disable_kernel_timing", []() { KernelTimers::Get().Disable(); })
      .def_static("collect_kernel_timing",
                  []() {
                    py::dict d;
                    for (auto& kv : KernelTimers::Get().Collect()) d[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
                    return d;
                  })
      .def("install_octree", [](ExpRunner& r, const Tensor& n, const Tensor& t, const Tensor& e) { SamplerOf(r)->InstallOctree(n, t, e); })
      .def("set_edge_pool",[](ExpRunner& r, const Tensor& e) { SamplerOf(r)->SetEdgePool(e); })
      .def("set_train_cameras", [](ExpRunner& r, const Tensor& w2c, const Tensor& intri, const Tensor& b) { SamplerOf(r)->SetTrainCameras(w2c, intri, b); })
      .def("set_forced_randoms",
           [](ExpRunner& r, const Tensor& noise, const Tensor& bg, const Tensor& edge_idx, const Tensor& edge_coords) {
             SamplerOf(r)->forced_noise_ = noise;
             r.renderer_->forced_bg_ = bg;
             SamplerOf(r)->forced_edge_idx_ = edge_idx;
             SamplerOf(r)->forced_edge_coords_ = edge_coords;
           })
      .def("clear_forced_randoms",
           [](ExpRunner& r) {
             SamplerOf(r)->forced_noise_ = Tensor(); r.renderer_->forced_bg_ = Tensor();
             SamplerOf(r)->forced_edge_idx_ = Tensor(); SamplerOf(r)->forced_edge_coords_ = Tensor();
           })
      .def("n_nodes", [](ExpRunner& r) { return SamplerOf(r)->pers_octree_->n_nodes_; })
      .def("proc_octree", [](ExpRunner& r, bool compact, bool subdivide, bool brute) { SamplerOf(r)->pers_octree_->ProcOctree(compact, subdivide, brute); })
      .def("tree_nodes", [](ExpRunner& r) { return SamplerOf(r)->pers_octree_->tree_nodes_gpu_; })
      .def("cur_batch_size", &ExpRunner::CurBatchSize)
      .def("batch_size_for", &ExpRunner::BatchSizeFor)  // ray count of the batch with that sequence number (fixed-lag average)
      .def_property("step_seq",  // training steps taken = the sequence number the next step's draws are keyed by (KeyedDraws.h)
                    [](ExpRunner& r) { return r.step_seq_; }, [](ExpRunner& r, int64_t s) { r.ResetStepSequence(s); })
      .def_readwrite("digest_table", &ExpRunner::digest_table_)
      .def_property("digest_taps",  // per-step checksums of the step's intermediate arrays (Renderer::DigestTap; debugging)
                    [](ExpRunner& r) { return r.renderer_->digest_taps_; }, [](ExpRunner& r, bool on) { r.renderer_->digest_taps_ = on; })
      .def("step_taps",  // int64 [steps, 1 + N_TAPS]: seq, then the taps (pts, dt, anchors, f0, survivors, bg, edge, colours, table grad,
           [](ExpRunner& r) {  // small grads, table, field MLP, colour MLP, app_emb), oldest first
             r.FinishPending();
             std::vector<int64_t> seqs;
             for (auto& d : r.renderer_->digest_) if (d.seq >= 0) seqs.push_back(d.seq);
             std::sort(seqs.begin(), seqs.end());
             Tensor out = torch::zeros({(int64_t) seqs.size(), 1 + Renderer::N_TAPS}, torch::kInt64);
             if (!r.renderer_->digest_tap_sums_.defined()) return out;
             Tensor sums = r.renderer_->digest_tap_sums_.cpu();
             for (size_t i = 0; i < seqs.size(); i++) {
               out[i][0] = seqs[i];
               out[i].narrow(0, 1, Renderer::N_TAPS).copy_(sums[seqs[i] % Renderer::kDigestRing]);
             }
             return out;
           })
      .def("step_digest",  // the last steps' (seq, iter, rays, marched, kept[, table checksum]), oldest first (flushes)
           [](ExpRunner& r) {
             r.FinishPending();
             py::list out;
             auto& dg = r.renderer_->digest_;
             Tensor sums = r.digest_table_sums_.defined() ? r.digest_table_sums_.cpu() : Tensor();
             std::vector<Renderer::StepDigest> rows;
             for (auto& d : dg) if (d.seq >= 0) rows.push_back(d);
             std::sort(rows.begin(), rows.end(), [](const Renderer::StepDigest& a, const Renderer::StepDigest& b) { return a.seq < b.seq; });
             for (auto& d : rows) {
               py::list row;
               row.append(d.seq); row.append(d.iter); row.append(d.n_rays); row.append(d.n_marched); row.append(d.n_kept);
               if (sums.defined()) row.append(sums.data_ptr<int64_t>()[d.seq % Renderer::kDigestRing]);
               out.append(row);
             }
             return out;
           })
      .def_readwrite("iter_step", &ExpRunner::iter_step_)
      .def_readwrite("check_nan", &ExpRunner::check_nan_)
      .def_readwrite("async_counts", &ExpRunner::async_counts_)
      .def_property("speculative_sampling",  // 0 / False never, 1 / True always, 2 while no leaf has died lately (default)
                    [](ExpRunner& r) { return r.renderer_->speculative_sampling_; },
                    [](ExpRunner& r, int mode) { r.renderer_->speculative_sampling_ = mode; })
      .def_property("tail_repair",  // speculative batches repaired by list compaction + a march of the tail behind the first dead leaf (A/B knob)
                    [](ExpRunner& r) { return static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->tail_repair_; },
                    [](ExpRunner& r, bool on) { static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->tail_repair_ = on; })
      .def_readwrite("fused_tail", &ExpRunner::fused_tail_)
      .def_readwrite("spec_start_without_event", &ExpRunner::spec_start_without_event_)
      .def_readwrite("draws_off_main", &ExpRunner::draws_off_main_)  // ExpRunner::Train's batch draws on the tail stream (A/B: False = main queue)
      .def_readwrite("exact_flag_order", &ExpRunner::exact_flag_order_)  // the step's tail inside the field backward's call (A/B: False = separate launches)
      .def_property("march_block_waves",  // workgroup size of the persistent march, in waves (1..16)
                    [](ExpRunner& r) { return static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->march_block_waves_; },
                    [](ExpRunner& r, int n) { static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->march_block_waves_ = std::min(16, std::max(1, n)); })
      .def_property("march_blocks",  // > 0: speculative batches marched on that many persistent one-wave blocks (0: one block per 4 rays)
                    [](ExpRunner& r) { return static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->march_blocks_; },
                    [](ExpRunner& r, int n) { static_cast<PersSampler*>(r.renderer_->pts_sampler_.get())->march_blocks_ = std::max(0, n); })
      .def_property("speculation_depth",  // the batch after next is begun two steps ahead: 1 never, 2 once the octree has outgrown the LDS walk, 3 always
                    [](ExpRunner& r) { return r.renderer_->spec_depth_; },
                    [](ExpRunner& r, int d) { r.FinishPending(); r.renderer_->DropPendingSamples(); r.renderer_->spec_depth_ = std::max(1, std::min(3, d)); })
      .def("speculation_counters",  // batches sampled ahead of the stat update / behind it, rays repaired after a leaf died
           [](ExpRunner& r) {
             r.FinishPending();
             py::dict d;
             d["speculative"] = r.renderer_->n_speculative_;
             d["fallback"] = r.renderer_->n_spec_fallback_;
             d["dropped"] = r.renderer_->n_spec_dropped_;
             auto& o = *SamplerOf(r)->pers_octree_;
             d["rays_repaired"] = o.n_repaired_.defined() ? o.n_repaired_.item<int>() : 0;
             d["stat_updates"] = o.epoch_;
             return d;
           })
      .def("counters",  // running totals over training-mode steps; flushes (the last streaming step's count is still in flight)
           [](ExpRunner& r) {
             r.FinishPending();
             py::dict d;
             d["total_meaningful"] = r.renderer_->total_kept_pts_;
             d["total_marched"] = r.renderer_->total_all_pts_;
             return d;
           })
      .def_readonly("cur_lr", &ExpRunner::cur_lr_)
      .def_property("n_edge_pts", [](ExpRunner& r) { return r.renderer_->n_edge_pts_; }, [](ExpRunner& r, int n) { r.renderer_->n_edge_pts_ = n; })
      .def_property_readonly("fineness", [](ExpRunner& r) { return r.global_data_pool_->ray_march_fineness_; })
      .def_property_readonly("oct_per_ray", [](ExpRunner& r) { return r.global_data_pool_->sampled_oct_per_ray_; })
      .def_property_readonly("sampled_per_ray", [](ExpRunner& r) { return r.global_data_pool_->sampled_pts_per_ray_; })
      .def_property_readonly("meaningful_per_ray", [](ExpRunner& r) { return r.global_data_pool_->meaningful_sampled_pts_per_ray_; })
      .def_property_readonly("n_volumes", [](ExpRunner& r) { return r.global_data_pool_->n_volumes_; })
      .def("update_ada_params", &ExpRunner::UpdateAdaParams);
}
