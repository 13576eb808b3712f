"""The driver's build() entry point must keep working (round 1 shipped one that raised)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_build_entry_point_runs():
    import __graft_entry__ as g
    g.build()       # make is a no-op on an up-to-date tree; raises on ABI mismatch or a failed compile
