"""ASan + UBSan and ThreadSanitizer builds of the CPU-side native code (GPU sanitizers are not available on the
pool): the host front end tests and the oracle KAT tests re-run in a child process with the instrumented
libraries preloaded."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _san_libs():
    out = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
        if not os.path.isabs(p) or not os.path.exists(p):
            return None
        out.append(p)
    return out


@pytest.mark.skipif(_san_libs() is None, reason="sanitizer runtimes not installed")
def test_host_front_end_under_asan_ubsan():
    subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "inquistr_amd", "host"), "../lib/libinquistr_host_asan.so"],
                          stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["LD_PRELOAD"] = " ".join(_san_libs())
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["INQ_FUZZ_TRIALS"] = "12"
    env["INQ_HOST_LIB"] = os.path.join(ROOT, "inquistr_amd", "lib", "libinquistr_host_asan.so")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_host_frontend.py", "tests/test_host_spans.py", "tests/test_outlier_oracle.py", "tests/test_csi_index.py",
                        "tests/test_error_class.py", "tests/test_corrupt_inputs.py", "tests/test_multi_device.py", "-x", "-q", "-k", "not cli",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout


def _tsan_lib():
    p = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_tsan_lib() is None, reason="ThreadSanitizer runtime not installed")
def test_host_threads_under_tsan():
    """The sweep front end's worker pool, the span loader's parallel reads, the lazily built index anchors (shared by the
    span planner's helper thread and the loader), the device parts of the multi-device entry and the bounded waits for a context
    thread that never finishes under ThreadSanitizer.  The CLI test is left out: it starts another program."""
    subprocess.check_call(["make", "-j8", "-C", os.path.join(ROOT, "inquistr_amd", "host"), "../lib/libinquistr_host_tsan.so"],
                          stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["LD_PRELOAD"] = _tsan_lib()
    env["TSAN_OPTIONS"] = "halt_on_error=1:report_signal_unsafe=0"
    env["INQ_HOST_LIB"] = os.path.join(ROOT, "inquistr_amd", "lib", "libinquistr_host_tsan.so")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_host_frontend.py", "tests/test_host_spans.py", "tests/test_csi_index.py",
                        "tests/test_multi_device.py",  # one thread per device part; waiters that give a dead context thread up
                        "-x", "-q", "-k", "not cli", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stdout + r.stderr, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout
