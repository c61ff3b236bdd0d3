"""The reference crate's own tests as a C++ program against the C ABI and the C++
host mirror — tests/native/reference_tests.cpp.  CPU part here, GPU part under -m gpu."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

NATIVE = os.path.join(ROOT, "tests", "native")
EXE = os.path.join(NATIVE, "reference_tests")


@pytest.fixture(scope="module")
def exe(pkg):  # pkg: makes sure libcsvsimd_hip.so exists before linking against it
    if not (os.path.exists(EXE) and os.path.exists(os.path.join(NATIVE, "libreference_tests.so"))):
        subprocess.run(["make", "-C", NATIVE, "-s"], check=True)   # normally done by __graft_entry__.build()
    return EXE


def run(exe, mode):
    p = subprocess.run([exe, mode, GOLDEN], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ok: all checks passed" in p.stdout
    return p.stdout


def test_reference_tests_cpu_part(exe):
    out = run(exe, "cpu")
    assert "doc_test_boundaries" in out and "header_new" in out


@pytest.mark.gpu
def test_reference_tests_gpu_part(exe, capfd):
    # in-process (ctypes): this runner already holds the GPU and must not exec another program
    import ctypes
    lib = ctypes.CDLL(os.path.join(NATIVE, "libreference_tests.so"))
    lib.run_reference_tests.restype = ctypes.c_int
    lib.run_reference_tests.argtypes = [ctypes.c_int, ctypes.c_char_p]
    rc = lib.run_reference_tests(1, GOLDEN.encode())
    out = capfd.readouterr().out
    assert rc == 0, out
    assert "reader::tests::mk_index" in out and "create(sample_rx.csv)" in out and "ok: all checks passed" in out


def test_ingest_pool_under_thread_sanitizer():
    """csv-simd_amd/host/ingest_pool.hpp (CopyPool, TaskThread: what the ingest pipelines run their host side on) built with
    -fsanitize=thread and driven like a pipeline drives it — a stager and an expander thread and the caller slicing work over
    one pool, hot and cold starts, chunk 0 by the caller while the stager is on chunk 1: every byte in place, no data race
    reported.  No GPU involved (GPU sanitizers are not available on this pool; the host side is where the threads are)."""
    subprocess.run(["make", "-C", NATIVE, "-s", "pool_stress"], check=True)
    p = subprocess.run([os.path.join(NATIVE, "pool_stress")], capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "pool_stress ok" in out, out[-3000:]
    assert "ThreadSanitizer" not in out, out[-3000:]
