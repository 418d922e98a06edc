"""Runs tools/xcd_probe.hip on the GPU box: random gathers / atomics, whole-table vs XCD-partitioned regions."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "xcd_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "xcd_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = "cuda"
for region_mib in (1.5, 3.0, 6.0):
    region_entries = int(region_mib * (1 << 20) / 4)
    table = torch.zeros(8 * region_entries * 2, dtype=torch.float16, device=dev)
    sink = torch.zeros(1, device=dev)
    for do_atomic in (0, 1):
        for mode in (0, 1):
            blocks, iters = 2048, 256
            args = (ctypes.c_void_p(0), ctypes.c_void_p(table.data_ptr()), ctypes.c_uint32(region_entries), 8, mode, do_atomic, iters, blocks, ctypes.c_void_p(sink.data_ptr()))
            lib.probe_launch(*args); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): lib.probe_launch(*args)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            n = blocks * 256 * iters
            print("region %.1f MiB x8  %s  %s : %.3f ms  %.1f G accesses/s" % (region_mib, "atomic" if do_atomic else "gather", "xcd-partitioned" if mode else "whole-table     ", ms, n / ms / 1e6), flush=True)
