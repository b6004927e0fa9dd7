"""The four randomised parity harnesses (tests/fuzz_*.py, written to be run by hand with
hundreds of cases) at a fixed seed and a small number of cases each, so that every run of the
GPU suite exercises random shapes, ranks, chunk lengths, scales and dtypes through the half-sweep
(src/als.py:218-250), the entry points around it (src/als.py:137-140, 161-165, 275-279, 347-364),
whole fits (src/als.py:109-171) and the IMC evaluations (src/imc.py:43-96) against the oracle.
A harness exits with status 1 when a case fails; its report is printed."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('script,cases,seed', [('fuzz_half_sweep.py', 100, 11),
                                               ('fuzz_aux.py', 100, 12),
                                               ('fuzz_fit.py', 40, 13),
                                               ('fuzz_imc.py', 100, 14)])
def test_fuzz_harness(script, cases, seed, capsys):
    from alsmi import device as dev
    dev.require_gpu()
    argv = sys.argv
    sys.argv = [script, str(cases), str(seed)]
    try:
        with pytest.raises(SystemExit) as stop:
            runpy.run_path(os.path.join(HERE, script), run_name='__main__')
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert stop.value.code == 0, out[-4000:]
    assert '%d cases, 0 failed' % cases in out
