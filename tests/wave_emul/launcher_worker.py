"""Test infrastructure.  `python -m f2_nerf_amd.run ...` on the emulated stack: run.main() as it stands with the emulated host module behind
runtime.host() and the CPU as "the device" (bench_dp_worker.stand_ins); with RANK / WORLD_SIZE / MASTER_* in the environment it is one rank
of the launcher's data-parallel training (torch.distributed on gloo, reported as "nccl" so that f2_nerf_amd.parallel takes the native
attach)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import bench_dp_worker
    bench_dp_worker.stand_ins()
    from f2_nerf_amd import run
    return run.main(sys.argv[1:])


if __name__ == "__main__":
    sys.exit(main())
