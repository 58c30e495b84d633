#!/usr/bin/env python
"""Build the reference's own native mesh-extraction libraries into oracle/_ref/ (checker only, never shipped).

SURVEY section 8c: the ONet-Mesh path (row N3) uses two Cython/C++ extensions of the reference,
``im2mesh/utils/libmise/mise.pyx`` (multi-resolution iso-surface extraction) and ``im2mesh/utils/libmcubes``
(marching cubes).  They compile from their own few source files with Cython + g++ (no reference build system, no
stand-ins): the sources are compiled WHERE THEY LIE under /root/reference, only the build products go to
``oracle/_ref/`` (git-ignored; they travel to the GPU box like our own .so files).  Used by the tests to validate
``oracle/mesh_oracle.py`` (the Python around them) and the HIP mesh path; nothing in the product imports them.

    python oracle/build_ref.py        (needs /root/reference; a no-op with a message otherwise)
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = "/root/reference/ONet/im2mesh/utils"


def main() -> int:
    if not os.path.isdir(REF):
        print("oracle/build_ref.py: %s not present - using the prebuilt oracle/_ref (if any)" % REF)
        return 0
    import numpy
    from Cython.Build import cythonize
    from setuptools import Extension, setup
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ifd_ref_build_")
    # NumPy 2 removed the PyArray_DOUBLE / PyArray_ULONG aliases the 2013-era wrapper uses
    macros = [("PyArray_DOUBLE", "NPY_DOUBLE"), ("PyArray_ULONG", "NPY_ULONG"), ("NPY_NO_DEPRECATED_API", "0")]
    exts = [
        Extension("ref_mise", [os.path.join(REF, "libmise", "mise.pyx")], language="c++"),
        Extension("ref_mcubes", [os.path.join(REF, "libmcubes", f) for f in ("mcubes.pyx", "pywrapper.cpp", "marchingcubes.cpp")],
                  language="c++", include_dirs=[numpy.get_include(), os.path.join(REF, "libmcubes")], define_macros=macros),
    ]
    # module names must match the .pyx basenames for the init symbol: build under the original names, rename the files
    exts[0].name, exts[1].name = "mise", "mcubes"
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        setup(script_args=["build_ext", "--build-lib", tmp, "--build-temp", os.path.join(tmp, "t")],
              ext_modules=cythonize(exts, build_dir=os.path.join(tmp, "cy"), quiet=True,
                                    compiler_directives={"language_level": "3"}))
    finally:
        os.chdir(cwd)
    for f in os.listdir(tmp):
        if f.endswith(".so"):
            shutil.copy(os.path.join(tmp, f), os.path.join(OUT, f))
            print("built oracle/_ref/" + f)
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
