"""A plain C99 program against include/fwx.h + include/fwx_host.h and libfwx.so: the boundary is
usable from a foreign host language with nothing but a C compiler."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "c_abi_consumer.c")
PKG = os.path.join(ROOT, "floydwarshall_amd")


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_consumer")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           SRC, "-o", exe, "-L" + PKG, "-lfwx", "-Wl,-rpath," + PKG,
           "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_headers_are_valid_c99_and_the_library_links_from_c(tmp_path):
    """No GPU needed: the program builds warning-free as strict C99, links, starts, and -- there
    being no HIP device here -- stops at its first check instead of falling back to anything."""
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if r.returncode == 0:
        pytest.skip("a HIP device is present: covered by the gpu test")
    assert r.returncode == 2 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_program_solves_the_reference_graph_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "c_abi_consumer: OK" in r.stdout
