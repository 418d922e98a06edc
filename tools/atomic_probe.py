import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "atomic_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-shared", "-fPIC", os.path.join(here, "atomic_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
names = {0: "pk_add_f16", 1: "add_f32", 2: "add_i32", 3: "add_u64", 4: "2x add_f32", 5: "pk_add_f16 4-lane dup"}
for mib in (17, 3):
    span = mib * (1 << 20) // 4
    table = torch.zeros(span, dtype=torch.int32, device="cuda")
    for kind in range(6):
        for part in ((0, 1) if mib == 17 else (0,)):
            blocks, iters = 1024, 64
            args = (ctypes.c_void_p(0), ctypes.c_void_p(table.data_ptr()), ctypes.c_uint32(span), kind, part, iters, blocks)
            lib.atomic_probe_launch(*args); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): lib.atomic_probe_launch(*args)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            n = blocks * 256 * iters
            print("table %2d MiB %-22s %s: %.3f ms  %.1f G lane-atomics/s" % (mib, names[kind], "xcd-part" if part else "whole   ", ms, n / ms / 1e6), flush=True)
