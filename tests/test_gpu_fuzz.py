"""A fixed slice of the randomised differential test (tests/fuzz/fuzz_parity.py): random meshes, settings, tuning switches, host models (one solver,
hosted ranks, sb_group_*) and mid-run host actions, every result bitwise against the CPU oracle. The seeds are fixed, so the test is the same
run every time; the open-ended form is `python tests/fuzz/fuzz_parity.py --seconds N --seed S` (profiles/r04x_fuzz_parity.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fixed_slice_of_the_fuzzer_is_bitwise():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_parity.py"), "--seed", "0", "--max", "160", "--seconds", "400"],
                       capture_output=True, text=True, timeout=600)
    lines = r.stdout.splitlines()
    summary = [l for l in lines if l.startswith("SUMMARY")]
    bad = [l for l in lines if l.split(" ", 1)[0] in ("MISMATCH", "ERROR", "CRASH")]
    assert summary and not bad and r.returncode == 0, "\n".join(bad[:5] + summary + [r.stderr[-800:]])
    assert "160 scenarios" in summary[0], summary[0]
    # the slice holds what once failed -- a cloth or a tet blob through a group under the block partition (planned whole since) -- and the odd meshes
    ok = [l for l in lines if l.startswith("OK")]
    assert any(("kind=tets" in l or "kind=cloth" in l) and "host=group" in l and "partition=blocks" in l for l in ok)
    assert sum("kind=odd" in l for l in ok) >= 10

@pytest.mark.gpu
@pytest.mark.parametrize("seed", [2005569])
def test_seeds_that_once_failed(seed):
    """2005569: particles on a line through sb_group_* on 4 ranks under the block partition -- the windows' pair hashes agree, yet ONE window holds
    a leftover (T2) layer and the others none: tick programs of different shape (the mailbox layout follows the number of halo slots). The
    agreement now carries the shape; the group falls back to the whole mesh."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_parity.py"), "--only", str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SUMMARY 1 scenarios: OK 1" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.gpu
def test_fixed_slice_of_the_schedule_fuzzer_is_bitwise():
    """tests/fuzz/fuzz_schedules.py: one rank with a self-exchange, every drawn (transport, schedule) pair against the serialised eager schedule."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_schedules.py"), "--seed", "0", "--max", "10", "--seconds", "300"],
                       capture_output=True, text=True, timeout=600)
    lines = r.stdout.splitlines()
    summary = [l for l in lines if l.startswith("SUMMARY")]
    bad = [l for l in lines if l.split(" ", 1)[0] in ("MISMATCH", "ERROR", "CRASH")]
    assert summary and not bad and r.returncode == 0, "\n".join(bad[:5] + summary + [r.stderr[-800:]])
    assert "10 scenarios" in summary[0], summary[0]
