"""CPU checks of the drop-in boundary: libtodhip.so loads and exports every function that
include/todhip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from tod_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "todhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(todhip_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    L = capi.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libtodhip.so does not export " + n
    assert sorted(capi.EXPORTS) == names
    assert L.todhip_version() == 1


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Pose) == 4 + 36 + 12 + 8
    assert capi.DMATCH_DTYPE.itemsize == 16          # == sizeof(cv::DMatch)
    assert ctypes.sizeof(capi.Rng) == 144               # 33 x u32, 4 pad, u64
    assert ctypes.sizeof(capi.VerifyParams) == 12


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.TodError):
        capi.Context(0)


def test_product_does_not_reference_oracle():
    """The product path must never route through the CPU oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tod_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in src.lower(), f


def test_header_is_plain_c():
    """The boundary is a C ABI: include/todhip.h must compile as C99 (no C++ types in the signatures)."""
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "todhip.h"\nint main(void) { todhip_rng r; todhip_rng_seed(&r, 1); return 0; }\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-c", src, "-o", os.path.join(d, "t.o")], check=True)
