"""Test infrastructure.  One rank of `bench.py --gpus N` on the emulated stack: bench.main() as it stands, with the emulated host module
behind runtime.host(), the CPU as "the device", and torch.distributed's process group on gloo (reported as "nccl" to f2_nerf_amd.parallel,
so that it takes the NATIVE attach -- DataParallel.cpp over the shared-memory <rccl/rccl.h> -- as it does on the GPU box).  The ranks are
started by tests/test_wave_emul_cpu.py with RANK / WORLD_SIZE / MASTER_* in the environment, as torch.distributed.run would."""
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))


def stand_ins():
    """The emulated host module behind runtime.host(), the CPU as "the device", gloo under torch.distributed (reported as "nccl")."""
    import torch
    import torch.distributed as dist
    import wemu_build
    ctypes.CDLL(wemu_build.LIB, mode=ctypes.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location("_f2n_host_emul", wemu_build.build_host())
    host = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(host)
    import f2_nerf_amd  # noqa: F401
    from f2_nerf_amd import runtime
    runtime._host = host
    torch.set_num_threads(1)
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.max_memory_allocated = lambda *a, **k: 0
    torch.cuda.max_memory_reserved = lambda *a, **k: 0
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    to_dev = runtime.to_dev
    runtime.to_dev = lambda *arrays, device="cpu": to_dev(*arrays, device="cpu")
    cpu = torch.device("cpu")
    real_device = torch.device
    for name in ("rand", "randn", "zeros", "ones", "empty", "full", "tensor", "zeros_like"):
        def factory(*a, _orig=getattr(torch, name), **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return _orig(*a, **k)
        setattr(torch, name, factory)
    init = dist.init_process_group
    dist.init_process_group = lambda backend=None, **k: init("gloo", **{kk: v for kk, v in k.items() if kk != "device_id"})
    dist.get_backend = lambda group=None: "nccl"
    bol = dist.broadcast_object_list
    dist.broadcast_object_list = lambda objs, src=0, group=None, device=None: bol(objs, src=src, group=group)


def main():
    stand_ins()
    import bench
    bench.main()


if __name__ == "__main__":
    main()
