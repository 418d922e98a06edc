"""Ray-data-parallel collectives of the hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL over
xGMI on MI355X, "gloo" on CPU for tests).

The reference is single-GPU (no distributed code).  Rays are independent given replicated model +
octree, so each rank renders its own ray batch and the replicas exchange exactly two things per step:

  1. gradients -- two all-reduce(AVG): the ACTIVE prefix of the fp16 (x128 loss-scaled) hash-gradient table
     (17 * 2^log2_table_size halves: the only entries any level can address) and ONE flat fp32 buffer holding the two MLP
     gradient vectors and app_emb (ExpRunner::FlattenSmallGrads: small collectives are latency-bound, ~30 us each);
     averaging keeps the loss semantics of `mean` over the global batch;
  2. octree occupancy votes -- ONE all-reduce(MAX) of the [4, n_nodes] buffer (weight votes, alpha votes, visited marks,
     visit counts) between MarkVisit and the stats update (PersSampler.cu:555-603), so that every replica prunes /
     subdivides identically.

xGMI note (point-to-point, 7 links x ~153 GB/s per GPU): the table gradient is 17 MiB in its fp16 form -- half of what an
fp32 all-reduce of the same gradient would move, and 47 % less than the full 32 MiB allocation -- so a ring moves
2*(7/8)*17 MiB per GPU per step.
"""
import torch
import torch.distributed as dist


def _avg_(t, group=None):
    """In-place average over ranks.  NCCL/RCCL has a native AVG; gloo needs SUM + divide and has no fp16 arithmetic."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        return t
    if t.dtype == torch.float16:
        f = t.to(torch.float32)
        dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group)
        t.copy_((f / world).to(torch.float16))
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t.div_(world)
    return t


def active_table_halves(log2_table_size, n_levels=16):
    """Halves [0, (n_levels+1) * 2^log2) are the only ones addressed (level-overlap quirk)."""
    return (n_levels + 1) << log2_table_size


def make_grad_sync(grad_buffers, log2_table_size, group=None):
    """grad_buffers = [hash-gradient table (fp16, flat or [P,2]), field MLP, colour MLP, app_emb]."""
    table = grad_buffers[0].view(-1)[:active_table_halves(log2_table_size)]
    rest = list(grad_buffers[1:])

    def sync():
        _avg_(table, group)
        for b in rest:
            _avg_(b, group)
    return sync


def occupancy_sync(occ, group=None):
    """occ: int32 [4, n_nodes] = weight votes, alpha votes, visited marks, visit counts (all max-combinable)."""
    dist.all_reduce(occ, op=dist.ReduceOp.MAX, group=group)


def broadcast_states(runner, group=None):
    """Replicas must start from identical parameters, hash primes / biases and octree.  runtime.make_runner_from_cameras draws
    them from the torch RNG, the very generator that has to DIFFER per rank for distinct ray batches: whatever the ranks were
    constructed with, rank 0's checkpoint vector is loaded everywhere (sizes first -- another seed may have built another
    number of octree nodes)."""
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def bcast(states):
        out = []
        for t in states:
            n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
            dist.broadcast(n, src=0, group=group)
            b = t.to(dev).contiguous()
            if b.numel() != int(n.item()):
                b = torch.empty(int(n.item()), dtype=b.dtype, device=dev)
            if b.numel() > 0:
                dist.broadcast(b, src=0, group=group)
            out.append(b)
        return out
    runner.load_states(bcast(runner.states()))
    # the edge pool (its t_idx_a/b index rank 0's warps) and the training cameras are not part of the checkpoint vector
    if hasattr(runner, "aux_states"):
        runner.load_aux_states(bcast(runner.aux_states()))


def attach(runner, log2_table_size, group=None, overlap=None, native=None, hooks_for_one_rank=False):
    """Wire the exchanges into an ExpRunner.

    native (default: on for the nccl/RCCL backend): the C++ host creates its own RCCL communicator (ncclCommInitRank; the
    unique id travels through this process group once) and issues the collectives itself from inside TrainStep
    (csrc/host/DataParallel.cpp) -- one ncclGroup for the two gradient buffers on the communicator's stream, the occupancy
    MAX + survivor-count SUM on the compute stream; no Python, no GIL, no dispatcher on the step's critical path.  Rank 0's
    state is broadcast to every rank.  native=False keeps the torch.distributed hooks below (what the gloo tests run):

    3 collectives per training step through Python callbacks.

    overlap (default: on for the nccl/RCCL backend): the two gradient all-reduces are launched asynchronously right
    after backward and only awaited in the NEXT train_step, after its ray sampling (which reads neither parameters nor
    gradients) has been issued -- the 17 MiB table reduction over xGMI then runs under ~0.3 ms of sampler kernels
    instead of in front of the optimiser.  Call runner.flush() before reading parameters outside train_step /
    render_rays / states() (those flush themselves)."""
    if native is None:
        native = dist.get_backend(group) == "nccl" and hasattr(runner, "attach_data_parallel")
    if native:
        from . import runtime
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ids = [runtime.host().dp_new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=group, device=torch.device("cuda", torch.cuda.current_device()))
        # overlap: the gradient exchange runs underneath the next step's ray sampling, which therefore moves from "under this
        # step's backward" to the step boundary -- worth it as soon as there is an exchange to hide (world > 1)
        # (hooks_for_one_rank: a one-rank world installs no exchange unless asked to -- tests, overhead measurements)
        runner.attach_data_parallel(rank, world, ids[0], (world > 1) if overlap is None else bool(overlap), bool(hooks_for_one_rank))
        return
    if hasattr(runner, "states") and hasattr(runner, "load_states"):
        broadcast_states(runner, group)
    if hasattr(runner, "attach_data_parallel") and dist.get_world_size(group) > 1:
        from . import runtime
        runtime.host().dp_set_replica(dist.get_rank(group))  # a draw stream per rank (csrc/host/KeyedDraws.h), as the native attach does
    flat = runner.flatten_small_grads()
    table = runner.grad_buffers()[0].view(-1)[:active_table_halves(log2_table_size)]
    if overlap is None:
        overlap = dist.get_backend(group) == "nccl"
    if overlap:
        works = []

        def begin():
            works.append(dist.all_reduce(table, op=dist.ReduceOp.AVG, group=group, async_op=True))
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group, async_op=True))

        def end():
            for w in works:
                w.wait()  # the compute stream waits for RCCL's stream; the host does not block
            works.clear()
        runner.set_pipelined_grad_sync(begin, end)
    else:
        runner.set_grad_sync_hook(make_grad_sync([table, flat], log2_table_size, group))
    runner.set_occupancy_sync_hook(lambda occ: occupancy_sync(occ, group))
