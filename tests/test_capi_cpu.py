"""The C-ABI library on a machine without a GPU: it loads, exports exactly the symbols
include/rtx_capi.h declares, mirrors the header's struct layouts, and fails loudly
(no CPU fallback) when there is no device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "rtx_capi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rtx_lib):
    declared = header_symbols()
    assert declared == sorted(capi.SYMBOLS), "capi.SYMBOLS and include/rtx_capi.h disagree"
    for name in declared:
        assert hasattr(rtx_lib, name), f"librtx_hip.so does not export {name}"


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof of every struct, taken from the header itself with gcc."""
    src = tmp_path / "probe.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "rtx_capi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rtx_material), sizeof(rtx_sphere), sizeof(rtx_config),
         sizeof(rtx_scene), sizeof(rtx_stats), sizeof(rtx_scene_info));
  printf("%zu %zu %zu %zu\\n", offsetof(rtx_config, pool_paths), offsetof(rtx_config, result_buffer_mib),
         offsetof(rtx_stats, ms_generate), offsetof(rtx_scene_info, build_ms));
  return 0;
}
''')
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[:6] == [C.sizeof(capi.Material), C.sizeof(capi.Sphere), C.sizeof(capi.CConfig),
                         C.sizeof(capi.CScene), C.sizeof(capi.Stats), C.sizeof(capi.SceneInfo)]
    assert sizes[6:] == [capi.CConfig.pool_paths.offset, capi.CConfig.result_buffer_mib.offset,
                         capi.Stats.ms_generate.offset, capi.SceneInfo.build_ms.offset]
    assert C.sizeof(capi.Material) == 20      # Material.h:11-15
    assert capi.MATERIAL_DTYPE.itemsize == 20 and capi.SPHERE_DTYPE.itemsize == 36


def test_no_cpu_fallback_and_argument_errors(rtx_lib):
    ctx = C.c_void_p()
    cfg = capi.Config(width=8, height=8, pool_paths=256).to_c()
    rc = rtx_lib.rtx_create(C.byref(cfg), C.byref(ctx))
    if rc == 0:
        rtx_lib.rtx_destroy(ctx)
        pytest.skip("a GPU is present: the no-device path cannot be exercised here")
    assert rc == -2, rc  # RTX_ERR_NO_DEVICE
    assert b"no HIP device" in rtx_lib.rtx_last_error() or b"gfx950" in rtx_lib.rtx_last_error()
    assert not ctx.value
    assert rtx_lib.rtx_create(None, C.byref(ctx)) == -1
    assert b"null" in rtx_lib.rtx_last_error()
    assert rtx_lib.rtx_render(None, None, 1, 0, 0, None, 0, 0, None) == -1
    assert rtx_lib.rtx_set_scene(None, None) == -1
    with pytest.raises(capi.RtxError):
        capi.Renderer(capi.Config(width=8, height=8))


def test_missing_library_is_an_import_error(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "librtx_hip.so"))
