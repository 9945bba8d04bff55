"""CPU-side checks of the drop-in boundary: the C-ABI shared library builds for gfx950,
loads, exports every symbol include/rmqtt_gpu_router.h declares, and refuses to run
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from rmqtt_amd import build, capi
from tests.conftest import ROOT, has_gpu


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rmqtt_gpu_router.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rgr_[a-z_0-9]+)\s*\(", src)))


def test_header_is_plain_c99():
    hdr = os.path.join(ROOT, "include", "rmqtt_gpu_router.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    declared = header_symbols()
    assert declared == sorted(capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rgr_version()


def test_library_contains_gfx950_code_object():
    so = build.build_gpu()
    data = open(so, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    for kern in (b"walk_kernel", b"expand_kernel", b"compact_kernel"):
        assert kern in data


def test_struct_layouts_match_header(tmp_path):
    """sizeof of every struct of the header, from a C compiler, against the ctypes / numpy mirrors the
    tests and bench use (the Rust side mirrors the same numbers)."""
    import subprocess
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rmqtt_gpu_router.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(rgr_config),sizeof(rgr_tuple),sizeof(rgr_window),sizeof(rgr_stats),sizeof(rgr_result),"
                   "sizeof(rgr_filters_result),sizeof(rgr_retain_result),sizeof(rgr_publish_attr));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(capi.Config), capi.TUPLE_DTYPE.itemsize, C.sizeof(capi.Window), C.sizeof(capi.Stats), C.sizeof(capi.Result),
                     C.sizeof(capi.FiltersResult), C.sizeof(capi.RetainResult), capi.PUBLISH_ATTR_DTYPE.itemsize]
    assert sizes[:3] == [40, 12, 128]          # rgr_window grew by d_hits8 (RGR_FORMAT_DELIVER8) and d_topic_order (rgr_batch_set_order), r6


@pytest.mark.skipif(has_gpu(), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(capi.RgrError) as ei:
        capi.Router()
    assert ei.value.code == capi.RGR_EDEVICE
    assert "no CPU fallback" in str(ei.value)
