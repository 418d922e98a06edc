#!/usr/bin/env python3
"""Do reads of data the PREVIOUS kernel of the same stream wrote ever return the buffer's previous content when the GPU is heavily shared?
Pure ATen, none of this repository's kernels: on the main stream a 17 MiB f16 buffer (the size of the hash table's active prefix) is
rewritten with iteration-dependent values by one kernel (mul / add of a small integer pattern: exact in f16) and immediately summed by
another (int64 sum of the bit patterns: exact); the sum is compared, on the device, with the value computed from the iteration number.
Two side streams keep other kernels running (matrix products, fills) as the sampler's side streams do.  Started N times side by side
(tools/stale_read_probe.sh) the processes oversubscribe the hardware queues.  A mismatch is a read that did not see the preceding
kernel's write.  Round 5: profiles/r05_determinism.txt, section C."""
import argparse, os, sys, time
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=40.0)
ap.add_argument("--tag", default="")
args = ap.parse_args()
dev = "cuda"
n = 17 * (1 << 19)  # halves in the active prefix of the 2^19 x 16 table
buf = torch.zeros(n, dtype=torch.float16, device=dev)
pattern = (torch.arange(n, device=dev) % 7).to(torch.float16)  # 0..6: every value and every sum below is exact in f16
bad = torch.zeros((), dtype=torch.int64, device=dev)
worst = torch.zeros((), dtype=torch.int64, device=dev)
side = [torch.cuda.Stream(), torch.cuda.Stream()]
A = torch.randn(2048, 2048, device=dev, dtype=torch.float16)
junk = torch.zeros(8 << 20, device=dev)
base = int(pattern.to(torch.int64).sum().item())
torch.cuda.synchronize()
t0, it = time.time(), 0
while time.time() - t0 < args.seconds:
    for _ in range(50):
        k = it % 97 + 1
        torch.add(pattern, float(k), out=buf)                  # kernel 1 writes every element: pattern + k  (<= 103: exact)
        got = buf.to(torch.int64).sum()                        # kernel 2 (+ cast) reads it back
        want = base + k * n
        d = (got - want).abs()
        bad += (d != 0).to(torch.int64)
        worst.copy_(torch.maximum(worst, d))
        with torch.cuda.stream(side[it & 1]):                  # co-running work on the side streams
            A = (A @ A).clamp_(-1.0, 1.0)
            junk.fill_(float(it & 255))
        it += 1
    torch.cuda.synchronize()
print("stale_read_probe%s: %d write->read pairs of %d MiB in %.1f s | reads that did not see the preceding kernel's write: %d (largest |sum error| %d)"
      % ((" " + args.tag) if args.tag else "", it, n * 2 >> 20, time.time() - t0, int(bad), int(worst)), flush=True)
