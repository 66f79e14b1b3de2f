"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle (CPU build only: GPU
sanitizers are not available on this pool).  The reference has no sanitizer or race-detection
story; the checker that every parity claim rests on gets one."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "selfcheck_san")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("selfcheck.c", "sparse_oracle.c", "lu_oracle.c", "cpu_fair.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fopenmp", "-std=c11", "-o", exe] + srcs + ["-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "selfcheck OK" in r.stdout, r.stdout + r.stderr
