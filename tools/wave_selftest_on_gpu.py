"""Holds an MI355X to the known answers the wavefront emulation is held to (tests/wave_emul/known_answers.py): compiles
tests/wave_emul/selftest.hip with hipcc for gfx950 and runs its kernels on cuda:0.  A pass says the emulation's reading of the ISA
-- DPP controls and masks, bound_ctrl, ds_bpermute from inactive lanes, EXEC masks under divergence, reconvergence at the end of a
loop body, the MFMA fragment layouts -- is the hardware's.  Needs a GPU (`gpurun -- python tools/wave_selftest_on_gpu.py`); since round 6 part
of the `-m gpu` suite (tests/test_gpu_wave_selftest.py)."""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))


def main():
    import torch
    import known_answers
    if not torch.cuda.is_available():
        sys.exit("needs an MI355X")
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, "libwave_selftest_gpu.so")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "wave_emul", "selftest.hip"),
                        "-o", lib], check=True)
        L = ctypes.CDLL(lib)

        def run(name, *args):
            dev = [torch.from_numpy(x).cuda() if isinstance(x, np.ndarray) else x for x in args]
            rc = getattr(L, name)(*[ctypes.c_void_p(t.data_ptr()) if isinstance(t, torch.Tensor) else t for t in dev])
            torch.cuda.synchronize()
            assert rc == 0, (name, rc)
            for x, t in zip(args, dev):
                if isinstance(x, np.ndarray):
                    x[...] = t.cpu().numpy()
        failed = 0
        for case in known_answers.CASES:
            try:
                getattr(known_answers, "check_" + case)(run)
                print("%-12s ok" % case)
            except AssertionError as e:
                failed += 1
                print("%-12s DIFFERS from the emulation's expectation: %s" % (case, str(e)[:400]))
        return failed


if __name__ == "__main__":
    sys.exit(main())
