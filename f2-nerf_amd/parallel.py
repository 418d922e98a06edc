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


def attach(runner, log2_table_size, group=None, overlap=None):
    """Wire both collectives into an ExpRunner (csrc/host/ExpRunner.cpp hooks): 3 collectives per training step.

    overlap (default: on for the nccl/RCCL backend): the two gradient all-reduces are launched asynchronously right
    after backward and only awaited in the NEXT train_step, after its ray sampling (which reads neither parameters nor
    gradients) has been issued -- the 17 MiB table reduction over xGMI then runs under ~0.3 ms of sampler kernels
    instead of in front of the optimiser.  Call runner.flush() before reading parameters outside train_step /
    render_rays / states() (those flush themselves)."""
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
