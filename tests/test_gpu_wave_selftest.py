"""The wavefront emulation's reading of the ISA, held against the hardware (round-5 verdict, next 8): the known-answer kernels of
tests/wave_emul/selftest.hip -- DPP controls and masks, bound_ctrl, ds_bpermute from inactive lanes, EXEC masks under divergence,
reconvergence at the end of a loop body, the two MFMA fragment layouts -- compiled by hipcc for gfx950 and run on the MI355X must
give what tests/wave_emul/known_answers.py expects of the emulator (tools/wave_selftest_on_gpu.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_hardware_gives_the_emulations_known_answers():
    spec = importlib.util.spec_from_file_location("wave_selftest_on_gpu", os.path.join(ROOT, "tools", "wave_selftest_on_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
