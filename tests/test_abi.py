"""CPU: libbzhip.so builds for gfx950, loads, exports every symbol include/bzhip.h declares, and
fails loudly (no CPU fallback) when no GPU is usable.  No compute calls here."""
import ctypes
import io
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bzhip.h")).read()
    return sorted(set(re.findall(r"BZH_API[^;(]*?\b(bzh_\w+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for need in ("bzh_create", "bzh_destroy", "bzh_encode", "bzh_encode_device", "bzh_plan_device",
                 "bzh_encode_range_device", "bzh_assemble_device", "bzh_rle1_split", "bzh_bwt", "bzh_mtf",
                 "bzh_huffman", "bzh_crc32"):
        assert need in syms


def test_library_exports_every_declared_symbol(native):
    L = ctypes.CDLL(native.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), f"{name} declared in include/bzhip.h but not exported"
    assert native.MISSING == []
    assert sorted(native.SIGNATURES) == header_symbols()


def test_strerror(native):
    L = native.lib()
    assert L.bzh_strerror(0) == b"ok"
    assert b"HIP" in L.bzh_strerror(-3)
    assert L.bzh_strerror(-99) == b"unknown status"


def test_create_rejects_bad_arguments(native):
    L = native.lib()
    h = ctypes.c_void_p()
    assert L.bzh_create(ctypes.byref(h), 0, 0, 0) == -1    # level 0 (reference asserts 1..=9, lib/lib.rs:89)
    assert L.bzh_create(ctypes.byref(h), 0, 10, 0) == -1
    assert L.bzh_create(None, 0, 9, 0) == -1


def test_arch_gate(native):
    """bzh_create refuses any device whose gcnArchName is not gfx950 (BZH_E_HIP): the gate it applies is
    exported, so the wrong-arch path can be checked without such a device"""
    L = native.lib()
    assert L.bzh_arch_supported(b"gfx950") == 1
    assert L.bzh_arch_supported(b"gfx950:sramecc+:xnack-") == 1
    for bad in (b"gfx942", b"gfx942:sramecc+:xnack-", b"gfx90a", b"gfx9500", b"gfx95", b"", b"sm_90"):
        assert L.bzh_arch_supported(bad) == 0, bad
    assert L.bzh_arch_supported(None) == 0


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_fails_loudly(native):
    import banzai_amd
    with pytest.raises(native.BzhError) as e:
        native.Context(0, 9, 4)
    assert e.value.status == -3
    with pytest.raises(native.BzhError):
        banzai_amd.encode(io.BytesIO(b"abc"), io.BytesIO(), 9)


def test_encode_level_validation():
    import banzai_amd
    for bad in (0, 10, -1, "9", 9.0):
        with pytest.raises(ValueError):
            banzai_amd.encode(io.BytesIO(b""), io.BytesIO(), bad)


def test_product_never_imports_oracle():
    """the product package must not reference oracle/ (SURVEY: oracle is test infrastructure)"""
    pkg = os.path.join(ROOT, "banzai_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in src and "banzai_oracle" not in src and "from oracle" not in src, f


def test_encode_hands_views_only_to_copying_sinks(tmp_path):
    """banzai_amd.encode passes a view of its reusable output buffer to BytesIO / real files only; any other writer may
    retain the object and gets bytes (tests/test_gpu_parity.py::test_public_api_retaining_writer runs it on the GPU)"""
    import io
    import banzai_amd
    assert banzai_amd._copying_sink(io.BytesIO())
    with open(tmp_path / "f", "wb") as f:
        assert banzai_amd._copying_sink(f)
    with open(tmp_path / "g", "wb", buffering=0) as f:
        assert banzai_amd._copying_sink(f)

    class Keeper:
        def write(self, b):
            return len(b)
    assert not banzai_amd._copying_sink(Keeper())
    assert not banzai_amd._copying_sink([].append)


def test_no_kernel_uses_scratch_memory():
    """The compiler's own resource report (hipcc -Rpass-analysis=kernel-resource-usage, scripts/resource_usage.py) for the
    five translation units of libbzhip.so: no kernel may spill a vector register or use scratch memory -- round 4's
    chunk_finish ran 23 % of the step through 272 bytes of scratch per lane; round 5 holds it at 128 VGPRs without."""
    import concurrent.futures
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import resource_usage
    srcs = [os.path.join(ROOT, "banzai_amd", "csrc", f"{n}.hip") for n in ("api", "bwt", "mtf", "huffman", "rle1")]
    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        reports = list(ex.map(resource_usage.report, srcs))
    kernels = [r for rep in reports for r in rep]
    assert len(kernels) >= 50
    bad = [(r["name"], r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]")) for r in kernels
           if r.get("ScratchSize [bytes/lane]", "0") != "0" or r.get("VGPRs Spill", "0") != "0"]
    assert not bad, bad
