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
SHIM = os.path.join(ROOT, "tests", "cpp", "test_shim_sequence")


def build(exe=EXE):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libqexoracle.so"])
    src = exe + ".cpp"
    if os.path.exists(exe) and os.path.getmtime(exe) > max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "qexhip.hpp")),
            os.path.getmtime(os.path.join(ROOT, "include", "qexhip.h")),
            os.path.getmtime(os.path.join(ROOT, "qex_amd", "libqexhip.so"))):
        return
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           src, "-o", exe,
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


def test_nim_shim_call_sequence_compiles_and_binds_what_the_shim_declares():
    """tests/cpp/test_shim_sequence.cpp performs the C calls of every proc of qex_amd/nim/qexhip.nim (no Nim compiler in the
    image); here: it builds, and every `importc` line of the shim names an entry point include/qexhip.h declares."""
    import re

    build(SHIM)
    out = subprocess.check_output(["ldd", SHIM], text=True)
    assert "libqexhip.so" in out and "not found" not in out
    nim = open(os.path.join(ROOT, "qex_amd", "nim", "qexhip.nim")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qexhip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(qexhip_[A-Za-z0-9_]+)\s*\(", hdr))
    imported = set(re.findall(r"^proc (qexhip_[A-Za-z0-9_]+)\(", nim, flags=re.M))
    assert len(imported) >= 45 and imported <= declared, sorted(imported - declared)
    # every entry point a wrapper proc calls is one the C++ replay calls too
    called = set(re.findall(r"chk (qexhip_[A-Za-z0-9_]+)\(", nim)) | set(re.findall(r"chk (qexhip_[A-Za-z0-9_]+) *\(", nim))
    cpp = open(SHIM + ".cpp").read()
    missing = sorted(n for n in called if n + "(" not in cpp and n not in ("qexhip_comm_unique_id", "qexhip_comm_init", "qexhip_comm_info"))
    assert not missing, missing


@pytest.mark.gpu
def test_nim_shim_call_sequence_against_the_oracle():
    build(SHIM)
    env = dict(os.environ, OMP_NUM_THREADS="8")
    p = subprocess.run([SHIM], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    print(p.stdout)
    assert p.returncode == 0 and "shim sequence: Passed" in p.stdout, p.stdout


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


def test_peer_transport_host_rendezvous(tmp_path):
    """qex_amd/csrc/peer_shm.cpp, the out-of-band channel of the peer-memory transport (what QMP's init / barrier / max give
    QEX, src/comms/commsQmp.nim:14-33,127-140), CPU only: 1-8 forked ranks meet in the shm segment, pass 1000 barriers, agree
    bit for bit on max / min / rank-ordered sum; a rank that never arrives or reports a failure is an ERROR on the others
    within the timeout, never a hang; a slot is taken once."""
    exe = str(tmp_path / "test_peer_shm")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_peer_shm.cpp"), os.path.join(ROOT, "qex_amd", "csrc", "peer_shm.cpp"),
                           "-o", exe, "-lrt", "-lpthread"])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    print(p.stdout)
    assert p.returncode == 0 and "peer rendezvous: Passed" in p.stdout, p.stdout
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("qexhip_")], "a rendezvous segment was left behind"
