"""TEST INFRASTRUCTURE: build the reference's UNMODIFIED Cython binding (src/edt.pyx) against this repo's
drop-in headers (cpp/edt.hpp + cpp/edt_voxel_graph.hpp over the C ABI of include/edt_hip.h).

    python tests/cython_dropin.py            -> tests/_build/cydrop/edt.<EXT_SUFFIX>

The generated C++ lands in tests/_build/cydrop (NOT next to edt.pyx), so `#include "edt.hpp"` cannot pick
up the reference's header from src/: the only edt.hpp on the include path is ours.  Needs /root/reference
(absent on the GPU box: the built module travels with the tree, like every other in-tree .so).
"""
import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "euclidean-distance-transform-3d_amd")
OUT = os.path.join(ROOT, "tests", "_build", "cydrop")
PYX = "/root/reference/src/edt.pyx"


def module_path():
    return os.path.join(OUT, "edt" + sysconfig.get_config_var("EXT_SUFFIX"))


def up_to_date():
    so = module_path()
    if not os.path.exists(so):
        return False
    deps = [os.path.join(PKG, "cpp", "edt.hpp"), os.path.join(PKG, "cpp", "edt_voxel_graph.hpp"),
            os.path.join(ROOT, "include", "edt_hip.h")]
    if os.path.exists(PYX):
        deps.append(PYX)
    return os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps)


def build(force=False):
    """Returns the module path; raises FileNotFoundError when the reference tree is not there."""
    if up_to_date() and not force:
        return module_path()
    if not os.path.exists(PYX):
        if os.path.exists(module_path()):
            # no reference tree here (the GPU box): the module that travelled with the tree is the one to test,
            # even if a header was touched after it was built
            return module_path()
        raise FileNotFoundError(PYX)
    import numpy
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "edt_cydrop.cpp")
    subprocess.run(["cython", "-3", "--fast-fail", "--cplus", PYX, "-o", cpp], check=True, capture_output=True)
    # -DEDT_HIP_PYTHON_ERRORS (cpp/edt.hpp): the binding declares the transforms without `except +` and calls them nogil --
    # a failing call (no device, out of memory) then raises in Python instead of ending the interpreter in std::terminate
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w", "-DEDT_HIP_PYTHON_ERRORS",
           "-I" + os.path.join(PKG, "cpp"), "-I" + os.path.join(ROOT, "include"),
           "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(), cpp,
           "-L" + os.path.join(PKG, "lib"), "-ledt_hip", "-ldl",
           # found relative to the module itself, wherever the tree is checked out
           "-Wl,-rpath,$ORIGIN/../../../euclidean-distance-transform-3d_amd/lib",
           "-o", module_path()]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building the reference's edt.pyx against the drop-in headers failed:\n" + res.stderr[-4000:])
    os.remove(cpp)
    return module_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
