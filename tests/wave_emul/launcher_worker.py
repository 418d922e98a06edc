"""Test infrastructure.  `python -m f2_nerf_amd.run ...` on the emulated stack: run.main() as it stands with the emulated host module behind
runtime.host() and the CPU as "the device"; with RANK / WORLD_SIZE / MASTER_* in the environment it is one rank of the launcher's
data-parallel training (torch.distributed on gloo, reported as "nccl" so that f2_nerf_amd.parallel takes the native attach)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_dp_worker  # noqa: E402  (the same stand-ins; its main() would run bench.py)


def main():
    import bench_dp_worker as w
    import types
    # (reuse the patching of bench_dp_worker.main without running bench: a module-level hook)
    real_main = None
    import importlib
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_dp_worker.py")).read()
    src = src.replace("    import bench\n    bench.main()\n", "    from f2_nerf_amd import run\n    sys.exit(run.main(sys.argv[1:]))\n")
    mod = types.ModuleType("launcher_patched")
    mod.__file__ = w.__file__
    exec(compile(src, w.__file__, "exec"), mod.__dict__)
    mod.main()


if __name__ == "__main__":
    main()
