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
