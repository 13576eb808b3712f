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


def test_planner_on_the_fuzz_corpus_under_asan_ubsan(tmp_path):
    """The degenerate meshes of tests/fuzz/fuzz_plan.py (a point, a line, no constraints, complete graphs, NaN ...) through plan.cpp under ASan + UBSan."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    import fuzz_plan
    corpus = tmp_path / "corpus.bin"
    with open(corpus, "wb") as f:
        for seed in range(250):
            sc = fuzz_plan.make_scenario(seed)
            m = sc["_mesh"]
            np.array([m.n, len(m.dist_rest), len(m.vol_rest), len(m.bend_rest), sc["world"], sc["tile"], sc["partition"], -1, 0, 0, 0, 0], np.int32).tofile(f)
            np.ascontiguousarray(m.rest_pos, np.float32).tofile(f)
            for a in (m.dist_ij, m.vol_ijkl, m.bend_ijkl):
                np.ascontiguousarray(a, np.int32).tofile(f)
        # ranks' windows (sharded authoring): domain + whole-mesh ids, cut as the group host cuts them
        import fuzz_windows
        from softbodyunity_amd import native
        n_windows = 0
        for seed in range(40):
            sc = fuzz_windows.make_scenario(seed)
            rest, ij, W = sc["_rest"], sc["_ij"], sc["world"]
            dom = native.domain_from_mesh(rest, ij)
            for r in range(W):
                lo, hi = native.domain_window(dom, r, W, sc["dims"], sc["tile"])
                gid = np.nonzero(np.all((rest >= np.array(lo)) & (rest < np.array(hi)), axis=1))[0].astype(np.int32)
                if len(gid) == 0:
                    continue
                new = -np.ones(len(rest), np.int64); new[gid] = np.arange(len(gid))
                wij = new[ij[np.all(new[ij] >= 0, axis=1)]].astype(np.int32)
                np.array([len(gid), len(wij), 0, 0, W, sc["tile"], 1, r, *sc["dims"], 0], np.int32).tofile(f)
                np.ascontiguousarray(rest[gid], np.float32).tofile(f)
                wij.tofile(f); np.zeros(0, np.int32).tofile(f); np.zeros(0, np.int32).tofile(f)
                np.array([dom.n_global, *dom.lo, *dom.hi, dom.spacing, dom.fill], np.float64).tofile(f)
                gid.tofile(f)
                n_windows += 1
    exe = str(tmp_path / "plan_corpus_san")
    csrc = os.path.join(ROOT, "softbodyunity_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-pthread", *SAN, "-I", csrc, os.path.join(ROOT, "tests", "sanitize", "plan_corpus_san.cpp"),
                           os.path.join(csrc, "plan.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", SB_PLAN_THREADS="4")
    out = subprocess.run([exe, str(corpus)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and f"SANITIZE OK entries {250 + n_windows}" in out.stdout and n_windows > 100, out.stdout[-2000:] + out.stderr[-4000:]
