"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's `race detection / memory
checking` slot).  The oracle is the checker of every parity test; a checker that reads out of bounds or relies on undefined
behaviour (signed overflow in the hash, a shift of a negative key, an unaligned float load) checks nothing.  dsrg_oracle.c is
rebuilt with -fsanitize=address,undefined -fno-sanitize-recover=all and the golden-vector, cross-check and layer tests are run
against that build in a child interpreter (the sanitizer runtime has to be loaded before libc's allocator is used, hence
LD_PRELOAD in a fresh process); any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=%s" % name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if asan is None or ubsan is None:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    so = str(tmp_path / "liboracle_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fPIC", "-std=c11", "-shared",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", so, os.path.join(ROOT, "oracle", "dsrg_oracle.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD="%s:%s" % (asan, ubsan), DSRG_ORACLE_LIB=so, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",      # (CPython itself is not leak-clean)
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_layers.py"),
                        os.path.join(ROOT, "tests", "test_seed_rule.py")],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    # the child really ran on the sanitized build
    probe = subprocess.run([sys.executable, "-c", "from oracle import oracle as O; O.lib(); print(open('/proc/self/maps').read())"],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert "liboracle_san.so" in probe.stdout and "oracle/liboracle.so" not in probe.stdout, probe.stderr[-1000:]
