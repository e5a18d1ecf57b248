"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import subprocess

from prifit_amd import _lib


def test_header_symbols_exported(hiplib):
    names = _lib.declared_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(hiplib, n), n


def test_version_and_arch(hiplib):
    arch = ctypes.c_char_p()
    v = hiplib.prifit_version(ctypes.byref(arch))
    assert v >= 100 and arch.value == b"gfx950"


def test_code_object_is_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], capture_output=True,
                         text=True)
    # the fat binary embeds an amdgcn-amd-amdhsa--gfx950 code object
    with open(_lib.LIB_PATH, "rb") as f:
        blob = f.read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob


def test_header_is_plain_c():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = "#include \"prifit_hip.h\"\nint main(void){return 0;}\n"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-x", "c", "-",
                        "-o", "/dev/null"], input=src, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr
