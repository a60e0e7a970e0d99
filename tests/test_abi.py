"""The C-ABI shared library loads without a GPU and exports every entry point include/grafthip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "grafthip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.findall(r"GRAFT_API\s+[\w\s\*]+?\b(graft\w+)\s*\(", text)


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("graftInit", "graftShutdown", "graftLastError", "graftJobCreate", "graftJobTrace",
                 "graftJobDestroy", "graftRenderRows", "graftLoopOffsets", "graftScreenCoord", "graftCamInitialize"):
        assert must in names
    assert len(names) == len(set(names))


def test_library_exports_every_declared_symbol():
    from raytracing_amd._lib import LIB_PATH
    lib = ctypes.CDLL(LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), "libgrafthip.so does not export %s" % name


def test_struct_layouts_match_the_header():
    """ctypes mirrors must have the sizes the C compiler gives the header's structs."""
    import subprocess
    import tempfile
    from raytracing_amd import _lib
    src = '#include <stdio.h>\n#include "grafthip.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(GraftShape),' \
          ' sizeof(GraftLight), sizeof(GraftScene), sizeof(GraftView), sizeof(GraftCounters));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(_lib.GraftShape), ctypes.sizeof(_lib.GraftLight), ctypes.sizeof(_lib.GraftScene),
                     ctypes.sizeof(_lib.GraftView), ctypes.sizeof(_lib.GraftCounters)]


def test_compute_entry_points_fail_loudly_without_init():
    """No CPU fallback: before graftInit (or with no GPU) the compute entry points return an error code."""
    import raytracing_amd
    from raytracing_amd._lib import GRAFT_OK, lib
    from raytracing_amd.tracer import _bound_device
    if _bound_device is not None:
        return  # a GPU session already bound the library in this process
    rt = raytracing_amd.RayTracer(64, 64)
    rt.sceneInitialize([0, 0, 0], 256, 1000)
    scene, view, keep = rt.pack()
    job = ctypes.c_void_p()
    rc = lib.graftJobCreate(ctypes.byref(scene), ctypes.byref(view), ctypes.byref(job))
    assert rc != GRAFT_OK and not job.value
    assert b"graftInit" in lib.graftLastError()


def test_product_neither_imports_nor_links_the_oracle():
    """oracle/ is test infrastructure: no file of the package, the C sources or the headers may name it, and the
    shared library may not depend on it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for top in ("raytracing_amd", "include"):
        for folder, _, files in os.walk(os.path.join(root, top)):
            for name in files:
                if name.endswith((".py", ".h", ".hip", ".c", ".cpp", "Makefile")):
                    with open(os.path.join(folder, name), errors="replace") as f:
                        assert "oracle" not in f.read().lower(), "%s/%s names the oracle" % (folder, name)
    needed = subprocess.run(["readelf", "-d", os.path.join(root, "raytracing_amd", "csrc", "libgrafthip.so")],
                            capture_output=True, text=True, check=True).stdout
    assert "NEEDED" in needed and "oracle" not in needed.lower()


def test_integration_table_names_and_arities_match_the_header():
    """Every `graft*(...)` signature quoted in INTEGRATION.md names a function the header declares, with as many
    parameters as the header gives it (a reader who binds from the table must not be led astray)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "grafthip.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    declared = {}
    for m in re.finditer(r"GRAFT_API\s+[\w\s\*]+?\b(graft\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.S):
        params = m.group(2).strip()
        declared[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    assert len(declared) >= 25
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    quoted = re.findall(r"`(graft\w+)\(([^`]*)\)`", text)
    assert len(quoted) >= 20
    for name, args in quoted:
        assert name in declared, "INTEGRATION.md quotes %s, which include/grafthip.h does not declare" % name
        if args.strip() in ("...", "…"):
            continue
        arity = 0 if not args.strip() else args.count(",") + 1
        assert arity == declared[name], "%s: %d arguments in INTEGRATION.md, %d in the header" % (name, arity, declared[name])
    # and nothing the header exports is missing from the document
    for name in declared:
        assert name in text, "%s is not mentioned in INTEGRATION.md" % name
