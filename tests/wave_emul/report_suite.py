"""Runs tests/test_wave_emul_cpu.py (the GPU parity tests' bodies on the emulated wavefront: tests/wave_emul/) and writes one line
per case: outcome, time, launches, work-items, cross-lane operations, how many of those found their wave in more than one group
(divergent: the scheduler had to decide who is behind), barriers.  `python tests/wave_emul/report_suite.py > profiles/r05_emulated_suite.txt`
(no GPU needed; WEMU_SCHEDULE=1 / 2 in the environment runs the suite under the other workgroup / wave orders)."""
import ctypes
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emul"))


class Plugin:
    def __init__(self):
        self.rows = []
        self.lib = None

    def counters(self):
        if self.lib is None:
            import wemu_build
            p = wemu_build.LIB
            if not os.path.exists(p):
                return [0] * 5
            self.lib = ctypes.CDLL(p)
            self.lib.wemu_counter.restype = ctypes.c_long
        return [self.lib.wemu_counter(i) for i in range(5)]

    @pytest.hookimpl(hookwrapper=True)
    def pytest_runtest_call(self, item):
        c0, t0 = self.counters(), time.time()
        outcome = yield
        c1 = self.counters()
        self.rows.append((item.name, "ok" if outcome.excinfo is None else "FAILED", time.time() - t0, [b - a for a, b in zip(c0, c1)]))


def main():
    pl = Plugin()
    rc = pytest.main(["-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_wave_emul_cpu.py")] + sys.argv[1:], plugins=[pl])
    print("# tests/test_wave_emul_cpu.py on the emulated wavefront (tests/wave_emul/), schedule %s; exit code %d" % (os.environ.get("WEMU_SCHEDULE", "0"), rc))
    print("# %-92s %-6s %7s %9s %11s %12s %10s %9s" % ("case", "", "seconds", "launches", "work-items", "cross-lane", "divergent", "barriers"))
    tot = [0] * 5
    for name, ok, dt, c in pl.rows:
        # (counters of the default library only: the mutant / refnum / selftest libraries count for themselves)
        print("%-94s %-6s %7.2f %9d %11d %12d %10d %9d" % (name, ok, dt, c[0], c[4], c[1], c[2], c[3]))
        tot = [a + b for a, b in zip(tot, c)]
    print("# %d cases, %d failed; %.0f s; %d launches, %d work-items, %d cross-lane operations (%d divergent), %d barriers" % (
        len(pl.rows), sum(r[1] != "ok" for r in pl.rows), sum(r[2] for r in pl.rows), tot[0], tot[4], tot[1], tot[2], tot[3]))
    return rc


if __name__ == "__main__":
    sys.exit(main())
