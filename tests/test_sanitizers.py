"""AddressSanitizer + UBSan over the host-side native code (the planner and the C oracle), and ThreadSanitizer over the
planner's host threads. GPU sanitizers are not available on the pool, so this is the CPU build only:
tests/sanitize/plan_san.cpp drives plan.cpp directly."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_planner_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_san")
    csrc = os.path.join(ROOT, "softbodyunity_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-pthread", *SAN, "-I", csrc, os.path.join(ROOT, "tests", "sanitize", "plan_san.cpp"),
                           os.path.join(csrc, "plan.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", SB_PLAN_THREADS="4")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZE OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_planner_threads_under_tsan(tmp_path):
    # the planner splits its phases over host threads (plan.cpp parallel_chunks): no data race, and the same plan for any
    # thread count (checked bit for bit by tests/test_plan.py::test_plan_is_independent_of_the_thread_count)
    exe = str(tmp_path / "plan_tsan")
    csrc = os.path.join(ROOT, "softbodyunity_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-pthread", "-fsanitize=thread", "-g", "-O1", "-I", csrc,
                           os.path.join(ROOT, "tests", "sanitize", "plan_san.cpp"), os.path.join(csrc, "plan.cpp"), "-o", exe])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", SB_PLAN_THREADS="4")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "SANITIZE OK" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, out.stdout[-2000:] + out.stderr[-4000:]


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", *SAN, "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "sanitize", "oracle_san.c"), "-lm", "-o", exe])   # includes oracle.c
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2")
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SANITIZE OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
