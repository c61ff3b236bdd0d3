"""The C-ABI library loads and exports every symbol include/csvsimd.h declares (CPU only: no
compute call is made here).  Without a GPU every compute entry point must fail loudly."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "csvsimd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csvsimd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(pkg):
    so = pkg.LIB_PATH
    nm = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (csvsimd_[a-z0-9_]+)", nm))
    declared = header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if s not in exported]
    assert not missing, missing
    # nothing is exported behind the header's back
    assert sorted(exported) == declared, sorted(exported - set(declared))
    # and the Python binding knows each of them
    assert sorted(pkg.EXPORTED_SYMBOLS) == declared
    pkg.lib()  # resolves all prototypes


def test_abi_version_and_strerror(pkg):
    L = pkg.lib()
    assert L.csvsimd_abi_version() == pkg.ABI_VERSION == 5
    assert L.csvsimd_strerror(-4).decode().startswith("Unsupported csv structure")  # src/error.rs:19
    assert L.csvsimd_strerror(-3).decode() == "Invalid state"                          # src/error.rs:17
    assert L.csvsimd_strerror(-2).decode() == "Missing a value"                        # src/error.rs:15


def test_product_library_has_no_probe_hooks(pkg):
    # VERDICT r1 #8 / ADVICE: the function behind bench.py's roofline.frac must not be steerable from the
    # environment, and the ablation instantiations (DBG != 0) must not ship
    so = pkg.LIB_PATH
    assert os.path.basename(so) == "libcsvsimd_hip.so"
    assert not pkg.build_has_probes()
    nm = subprocess.run(["nm", "-C", so], check=True, capture_output=True, text=True).stdout
    inst = sorted(set(re.findall(r"__device_stub__stage1_kernel<(\w+), (\d+), (\d+), (\w+), (\w+)>", nm)))
    # emit / count-only x four dialect classifications, DBG always 0; + the batched launch; + the dense-emit instantiations
    # (emit only: reference dialect — one buffer or a batch — and another delimiter / quote byte)
    assert inst == sorted([(e, "0", d, "false", "false") for e in ("false", "true") for d in ("0", "1", "2", "3")]
                          + [("true", "0", "0", "true", "false"), ("true", "0", "0", "false", "true"),
                             ("true", "0", "0", "true", "true"), ("true", "0", "1", "false", "true")]), inst
    raw = open(so, "rb").read()
    assert b"CSVSIMD_PROBE" not in raw and b"zero_kernel" not in raw and b"finalize_kernel" not in raw
    # what bench.py reports as the timed kernel comes from the library, and is the default instantiation
    assert pkg.stage1_kernel_name(True) == "void csvsimd::stage1_kernel<true, 0, 0, false, false>(csvsimd::KernelArgs)"
    # an escape dialect runs the hashed classification (<..., 3>) when its special bytes hash without a collision,
    # the direct compares (<..., 2>) otherwise: the library says which
    assert pkg.stage1_kernel_name(False, pkg.Dialect(";", "'", "\\")).endswith("<false, 0, 3, false, false>(csvsimd::KernelArgs)")
    names = {pkg.stage1_kernel_name(True, pkg.Dialect(d, q, e)) for d in ";|\t:A" for q in "'`\"" for e in "\\^~%/"}
    assert {n.split(", ")[-3] for n in names} == {"2", "3"}


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be observed")
    assert pkg.device_count() == 0
    with pytest.raises(pkg.StructureError) as e:
        pkg.Context(0)
    assert e.value.code == pkg.ERR_NO_DEVICE
    with pytest.raises(pkg.StructureError) as e:
        pkg.selftest(0)
    assert e.value.code == pkg.ERR_NO_DEVICE


def test_product_never_touches_oracle():
    # the product path must not import, link or call anything under oracle/
    pkg_dir = os.path.join(ROOT, "csv-simd_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    ldd = subprocess.run(["ldd", os.path.join(pkg_dir, "csrc", "libcsvsimd_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
