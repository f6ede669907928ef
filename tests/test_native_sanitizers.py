"""ASan + UBSan over the native CPU code: the oracle's C restatement and the host mirror's pure
C++ (parsers, show, buildMatrix, optimum).  GPU sanitizers are not available on this pool, so this
is the sanitizer coverage there is."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g",
       "-O1"]


def _run(exe):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    return subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_mirror_cpu_code_under_asan_ubsan(tmp_path):
    host = os.path.join(ROOT, "floydwarshall_amd", "csrc", "host")
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(host, f) for f in ("algorithms.cpp", "parsers.cpp", "show.cpp")]
    srcs.append(os.path.join(ROOT, "tests", "native", "host_sanitize_main.cpp"))
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra"] + SAN +
                   ["-I" + host, "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", exe],
                   check=True)
    r = _run(exe)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host sanitize ok" in r.stdout


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_c_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-ffp-contract=off", "-pthread"] + SAN +
                   [os.path.join(ROOT, "oracle", "fw_oracle.c"), os.path.join(ROOT, "oracle", "fw_oracle_fast.c"),
                    os.path.join(ROOT, "tests", "native", "oracle_sanitize_main.c"),
                    "-o", exe, "-lpthread"], check=True)
    r = _run(exe)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle sanitize ok" in r.stdout
