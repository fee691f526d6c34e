"""The plain-C host (same call sequence and AoS strides as gadget-2.0.7-ngravs_amd/host/gadget_glue.c)."""
import os
import subprocess

import pytest


def _build(pkg):
    host = os.path.join(os.path.dirname(pkg.__file__), "host")
    exe = os.path.join(host, "host_shim_test")
    inc = os.path.join(os.path.dirname(pkg.__file__), "..", "include")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", os.path.join(host, "host_shim_test.c"), "-I" + inc, "-L" + libdir,
                           "-lngravs_hip", "-lm", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_host_links_against_the_abi(pkg, have_lib):
    """CPU: the C host compiles against include/ngravs_hip.h and links against the library"""
    exe = _build(pkg)
    assert os.path.exists(exe)


@pytest.mark.parametrize("flags", [[], ["-DPERIODIC", "-DPMGRID=64"], ["-DPERIODIC", "-DPMGRID=64", "-DFORCETEST=0.1"],
                                   ["-DFORCETEST=0.1", "-DN_GRAVS=3"]])
def test_glue_compiles_against_the_reference_interface(pkg, flags):
    """gadget_glue.c is what a maintainer drops into the reference tree.  The reference cannot be built here (GSL, FFTW-2),
    so the glue is compiled -fsyntax-only -Wall -Wextra -Werror against tests/glue_stub/: declarations of exactly the
    globals / prototypes it touches (names and layouts from SURVEY.md 8(a'), 8(b)), for the tree-only, TreePM and FORCETEST
    variants of the reference's Makefile options."""
    root = os.path.join(os.path.dirname(pkg.__file__), "..")
    glue = os.path.join(os.path.dirname(pkg.__file__), "host", "gadget_glue.c")
    cmd = ["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-DNGRAVS_BUILD_INSIDE_REFERENCE", "-DDOUBLEPRECISION",
           "-DUNEQUALSOFTENINGS", "-I" + os.path.join(root, "tests", "glue_stub"), "-I" + os.path.join(root, "include")] + flags + [glue]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # single precision P[] must be refused at compile time, not read as garbage
    cmd.remove("-DDOUBLEPRECISION")
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode != 0 and "DOUBLEPRECISION" in out.stderr


def test_glue_source_mentions_every_entry_point(pkg):
    src = open(os.path.join(os.path.dirname(pkg.__file__), "host", "gadget_glue.c")).read()
    for name in ("void domain_Decomposition(void)", "int force_treebuild(int npart)", "void gravity_tree(void)",
                 "void pmforce_periodic(void)", "void force_treeallocate(int maxnodes, int maxpart)",
                 "void force_treefree(void)", "peanokey peano_hilbert_key(int x, int y, int z, int bits)",
                 "void peano_hilbert_order(void)", "void pm_init_periodic(void)"):
        assert name in src, name


@pytest.mark.gpu
def test_c_host_runs_on_gpu(pkg, have_lib):
    exe = _build(pkg)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
