"""Record layouts (core/raydata.h:40-94, core/CentroidBVH.h:13-59, core/MtlLib.h:7-15) and the
C-ABI surface: the library loads and exports every function include/rt_amd.h declares."""
import ctypes
import os
import re
import subprocess

from conftest import ROOT


def test_layout_header_compiles_as_c_and_cpp(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "rt_amd.h"\nint main(void){return (int)sizeof(rt_ray_isect)-64;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / ("a_" + cc)
        subprocess.check_call([cc, std, "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"),
                               str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_every_declared_symbol_is_exported(rt):
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(rt_[a-z0-9_]+|ray_scene_intersection)\s*\(", hdr))
    names.discard("rt_status")
    assert "ray_scene_intersection" in names and len(names) >= 35
    lib = ctypes.CDLL(rt.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes binding covers the same set
    assert set(rt.EXPORTS) == names


def test_error_reporting_without_gpu_calls(rt):
    h = ctypes.c_void_p()
    rc = rt.lib.rt_mesh_load_obj(b"/nonexistent/file.obj", 1, ctypes.byref(h))
    assert rc == -2 and b"cannot open" in rt.lib.rt_last_error()
    rc = rt.lib.rt_mesh_load_obj(None, 1, ctypes.byref(h))
    assert rc == -1
    rc = rt.lib.ray_scene_intersection(None, 4, None, None)
    assert rc == -1 and b"null scene" in rt.lib.rt_last_error()
