"""The C-ABI library loads without a GPU and exports every entry point include/pt_amd.h declares."""
import ctypes as C
import os
import re

import pytest

import path_tracing_amd as pt
from path_tracing_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^(?:const char\*|int|void|float)\s+(pt_\w+)\s*\(", text, flags=re.M)
    assert len(names) >= 15
    return names


def test_library_exports_every_declared_symbol():
    lib = pt.lib()
    for name in declared_functions():
        assert hasattr(lib, name), "libptamd.so does not export %s" % name


def test_version_and_error_string():
    lib = pt.lib()
    assert lib.pt_version().decode().startswith("ptamd")
    assert isinstance(lib.pt_last_error(), bytes)
    assert pt.device_count() >= 0


def test_struct_sizes_match_the_header():
    # sizes as compiled from include/pt_amd.h by the C compiler
    import subprocess
    import tempfile
    names = list(abi.EXPECTED_SIZES)
    src = '#include "pt_amd.h"\n#include <stdio.h>\nint main(void){' + "".join(
        'printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as tmp:
        c = os.path.join(tmp, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(tmp, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().split("\n"))
    for n in names:
        assert int(sizes[n]) == abi.EXPECTED_SIZES[n] == C.sizeof(getattr(abi, n)), n


def test_null_and_bad_arguments_are_errors_not_crashes():
    lib = pt.lib()
    assert lib.pt_scene_desc_validate(None) < 0
    assert b"null" in lib.pt_last_error()
    out = C.c_void_p()
    assert lib.pt_scene_create(None, 0, C.byref(out)) < 0
    d = abi.pt_scene_desc()
    d.abi_version = 99
    assert lib.pt_scene_desc_validate(C.byref(d)) < 0
    ptr = C.POINTER(abi.pt_scene_desc)()
    assert lib.pt_scene_desc_load(b"/nonexistent/file.ptscene", C.byref(ptr)) == -4  # PT_ERR_IO
    assert lib.pt_last_kernel_ms(None) < 0


def test_render_without_gpu_fails_loudly(golden_dir):
    if pt.device_count() > 0:
        pytest.skip("a GPU is present")
    scene = pt.LoadedScene(os.path.join(golden_dir, "cornell.ptscene"))
    with pytest.raises(pt.PtError) as e:
        pt.DeviceScene(scene)
    assert e.value.code == -3  # PT_ERR_DEVICE: there is no CPU fallback
