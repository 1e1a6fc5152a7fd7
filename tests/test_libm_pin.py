"""CPU: the product's atan2f / asinf / atanf restatements are bit-identical to this image's glibc libm
(quick stride here; the exhaustive 2^32 run is recorded in oracle/libm_pin_full.log)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_libm_pin_quick():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libm_pin"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_build", "libm_pin"), "quick"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert out.stdout.count("mismatches: 0") == 3, out.stdout


def test_full_pin_log_is_clean():
    log = open(os.path.join(ROOT, "oracle", "libm_pin_full.log")).read()
    assert "asinf  inputs 4294967296 mismatches: 0" in log
    assert "atanf  inputs 4294967296 mismatches: 0" in log
    assert "atan2f pairs  2147483648" in log and log.count("mismatches: 0") == 3
