"""Dependent-kernel boundary on this box: eager launches (host far ahead of the device) vs the same chain replayed from a hipGraph.
Backs DESIGN.md section 4 ("hipGraph capture was not attempted"): prints, per kernel size, the chain's device time both ways and
the per-boundary cost (chain time - n x single-kernel time) / n."""
import ctypes
import os
import subprocess

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "boundary_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "boundary_probe.hip"), "-o", so])
lib = ctypes.CDLL(so)
N = 32
print("chain of %d dependent kernels, 1024 blocks x 256 threads each; times in us" % N)
print("%-28s %12s %12s %16s %16s" % ("bytes moved per kernel", "eager chain", "graph chain", "eager per kernel", "graph per kernel"))
for mb in (0, 1, 8, 32, 128):
    out = (ctypes.c_float * 4)()
    rc = lib.boundary_probe(ctypes.c_size_t(mb << 20), N, 1024, out)
    if rc != 0:
        print("probe failed", rc)
        continue
    e, g, k1 = out[0] * 1e3, out[1] * 1e3, out[2] * 1e3
    print("%-28s %12.1f %12.1f %16.2f %16.2f" % ("%d MiB" % mb, e, g, e / N, g / N), flush=True)
