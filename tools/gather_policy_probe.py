import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "gather_policy_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "gather_policy_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
names = ["global_load", "buffer_load", "buffer_load sc0", "buffer_load nt", "buffer_load sc1", "buffer_load sc0 sc1", "buffer_load sc0 nt",
         "global_load nt", "8-byte pair loads (2 entries each)"]
blocks, iters = 2048, 512
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
for log2 in (19, 20):  # entries per XCD slice: 2 MiB (two levels of the bench table = what one L2 serves) and 4 MiB
    region = 1 << log2
    table = torch.randint(0, 1 << 30, (8 * region,), dtype=torch.int32, device="cuda")
    print("slice of 2^%d entries per XCD" % log2)
    for pol, name in enumerate(names):
        args = (ctypes.c_void_p(0), pol, ctypes.c_void_p(table.data_ptr()), region, iters, blocks, ctypes.c_void_p(sink.data_ptr()))
        assert lib.gather_probe_launch(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): lib.gather_probe_launch(*args)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        n = blocks * 256 * iters
        print("  %-36s %.3f ms  %.1f G lane-loads/s%s" % (name, ms, n / ms / 1e6, "  (x2 entries)" if pol == 8 else ""), flush=True)
