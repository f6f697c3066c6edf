"""Thousands of evaluations back to back, each checked bit for bit against the first evaluation of its X buffer (tools/soak.py):
the kernels are deterministic, so any difference would be a race in the fused finalize (arrival counters, polled partial slots)
or in the reuse of the workspace between launches.  Seven launch forms of the batch path, two uses of streams (one batch on two streams in turn; two batches in flight together) and three of the SNOPT callback (F and G poisoned before every call), a second each here (10 s each in the round's record,
profiles/r04/soak.txt: a million evaluations and calls, none differing)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_repeated_evaluations_are_bitwise_stable():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "1.0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if "evaluations in" in ln or "calls in" in ln]
    assert len(lines) == 12 and all(" 0 differ" in ln for ln in lines), res.stdout
