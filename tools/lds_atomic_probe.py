import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "lds_atomic_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "lds_atomic_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
names = ["ds_add_f32", "ds_add_u32", "ds_add_u64", "ds_pk_add_f16", "ds_add_f64", "plain racy RMW", "2x ds_add_f32 adjacent", "address generation only"]
blocks, iters = 512, 2000
sink = torch.zeros(blocks, dtype=torch.int32, device="cuda")
for kind, name in enumerate(names):
    args = (ctypes.c_void_p(0), kind, iters, blocks, ctypes.c_void_p(sink.data_ptr()))
    lib.lds_probe_launch(*args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): lib.lds_probe_launch(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = blocks * 256 * iters * (2 if kind == 6 else 1)
    print("%-26s %.3f ms  %.1f G lane-ops/s  (%.2f per clk per CU at 2.4 GHz, 256 CUs)" % (name, ms, n / ms / 1e6, n / ms / 1e6 / 2.4 / 256), flush=True)
