import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "atomic_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-shared", "-fPIC", os.path.join(here, "atomic_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
hip = ctypes.CDLL("libamdhip64.so")
names = {0: "pk_add_f16", 1: "add_f32", 2: "add_i32", 3: "add_u64", 4: "2x add_f32", 5: "pk_add_f16 4-lane dup", 6: "add_f32 wave scope",
         7: "add_f32 returning", 8: "plain racy RMW"}
parts = {0: "whole   ", 1: "xcd-part", 2: "one XCD "}


def run(ptr, span, kind, part, tag):
    blocks, iters = 1024, 64
    args = (ctypes.c_void_p(0), ctypes.c_void_p(ptr), ctypes.c_uint32(span), kind, part, iters, blocks)
    lib.atomic_probe_launch(*args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): lib.atomic_probe_launch(*args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n = blocks * 256 * iters // (8 if part == 2 else 1)
    print("%-12s span %6.2f MiB %-22s %s: %.3f ms  %.1f G lane-atomics/s" % (tag, span * 4 / 2**20, names[kind], parts[part], ms, n / ms / 1e6), flush=True)


torch.zeros(1, device="cuda")
for mib in (17, 0.25):
    span = int(mib * (1 << 20)) // 4
    table = torch.zeros(span, dtype=torch.int32, device="cuda")
    for kind in (0, 1, 6, 7, 8):
        for part in (0, 1, 2):
            run(table.data_ptr(), span, kind, part, "hipMalloc")
# other allocation flavours: fine-grained (0x1) and uncached (0x3) device memory
for flag, tag in ((0x1, "finegrained"), (0x3, "uncached")):
    span = 17 * (1 << 20) // 4
    ptr = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(ptr), ctypes.c_size_t(span * 4), ctypes.c_uint(flag))
    if rc != 0:
        print(tag, "alloc failed", rc); continue
    hip.hipMemset(ptr, 0, ctypes.c_size_t(span * 4))
    for kind in (0, 1):
        for part in (0, 1):
            run(ptr.value, span, kind, part, tag)
    hip.hipFree(ptr)
