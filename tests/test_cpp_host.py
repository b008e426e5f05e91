"""The compiled-language host layer (include/qexhip.hpp) and its C++ parity program.

CPU: the program must compile and link against libqexhip.so (+ the oracle as the checker).
GPU: it must run and pass (tests/cpp/test_stag_prop.cpp mirrors tests/examples/testStagProp.nim,
tests/reprod/trandgauge.nim and the self-test of src/gauge/wflow.nim).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_stag_prop")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libqexoracle.so"])
    src = os.path.join(ROOT, "tests", "cpp", "test_stag_prop.cpp")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "qexhip.hpp")),
            os.path.getmtime(os.path.join(ROOT, "qex_amd", "libqexhip.so"))):
        return
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           src, "-o", EXE,
           "-L" + os.path.join(ROOT, "qex_amd"), "-lqexhip", "-L" + os.path.join(ROOT, "oracle"), "-lqexoracle",
           "-Wl,-rpath," + os.path.join(ROOT, "qex_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-fopenmp"]
    subprocess.check_call(cmd)


def test_cpp_host_layer_compiles_and_links():
    build()
    out = subprocess.check_output(["ldd", EXE], text=True)
    assert "libqexhip.so" in out and "not found" not in out


@pytest.mark.gpu
def test_cpp_host_parity_program():
    build()
    env = dict(os.environ, OMP_NUM_THREADS="8")
    p = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    print(p.stdout)
    assert p.returncode == 0 and "Passed" in p.stdout, p.stdout


def test_closed_form_exp_against_the_reference_algorithm_and_long_double(tmp_path):
    """csrc/su3.h is host-callable: the flow's closed-form exp (m3_exp_tah) against the reference-algorithm exp (m3_exp:
    order-4 Taylor at v/2^20 + 20 squarings, matexp.nim) and a long-double series, over 4400 random, diagonal and
    nearly degenerate traceless anti-Hermitian matrices of norm 1e-9 ... 20 (tests/cpp/test_exp_ch.cpp states the bounds:
    never further from the exact result than 1.5 x the reference algorithm or 2e-15)."""
    exe = str(tmp_path / "test_exp_ch")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "qex_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_exp_ch.cpp"), "-o", exe])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    print(p.stdout)
    assert p.returncode == 0 and "0 scale(s) out of bounds" in p.stdout, p.stdout
