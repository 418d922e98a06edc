"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the f2-nerf hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package; the product (f2-nerf_amd/) never does.

  oracle.capi      ctypes binding of oracle/libf2n_oracle.so  (hand-written C restatement)
  oracle.ref       ctypes binding of oracle/_ref/libf2n_ref.so (the reference's own kernels, CPU build)
  oracle.pipeline  numpy restatement of the host-side pipeline (Renderer::Render, losses, Adam)
  oracle.octree_construct  restatement of PersOctree construction (fixture generator)
"""
