"""The plain-C host (same call sequence and AoS strides as gadget-2.0.7-ngravs_amd/host/gadget_glue.c)."""
import os
import subprocess

import pytest


def _build(pkg):
    host = os.path.join(os.path.dirname(pkg.__file__), "host")
    exe = os.path.join(host, "host_shim_test")
    inc = os.path.join(os.path.dirname(pkg.__file__), "..", "include")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-DWITH_RCCL", os.path.join(host, "host_shim_test.c"), "-I" + inc, "-L" + libdir,
                           "-lngravs_hip", "-lngravs_rccl", "-lm", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_host_links_against_the_abi(pkg, have_lib):
    """CPU: the C host compiles against include/ngravs_hip.h and links against the library"""
    exe = _build(pkg)
    assert os.path.exists(exe)


@pytest.mark.parametrize("flags", [[], ["-DPERIODIC", "-DPMGRID=64"], ["-DPERIODIC", "-DPMGRID=64", "-DFORCETEST=0.1"],
                                   ["-DFORCETEST=0.1", "-DN_GRAVS=3"], ["-DPERIODIC"], ["-DPERIODIC", "-DPMGRID=64", "-DNGRAVS_WITH_RCCL"]])
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


OPTION_SETS = [[], ["-DPERIODIC", "-DPMGRID=64"], ["-DPERIODIC", "-DPMGRID=64", "-DFORCETEST=0.1"], ["-DFORCETEST=0.1", "-DN_GRAVS=3"],
               ["-DPERIODIC"], ["-DPERIODIC", "-DPMGRID=64", "-DNGRAVS_WITH_RCCL"]]


@pytest.mark.parametrize("flags", OPTION_SETS)
def test_glue_defines_every_symbol_the_link_recipe_needs(pkg, flags, tmp_path):
    """INTEGRATION.md's recipe drops gravtree.o forcetree.o pm_periodic.o domain.o peano.o gravtree_forcetest.o from the
    reference's OBJS and adds gadget_glue.o.  tests/golden/glue_required_symbols.json (written by tools/glue_required_symbols.py
    in the build container: NAMES of the non-static functions those six units define and a kept unit calls, with the #if guards
    of their definitions) says what the kept objects will look for; the glue, compiled to an object against the interface
    stubs, must define every one of them under the same options -- checked with nm, so a forgotten symbol is a test failure
    here instead of a link error in the maintainer's tree."""
    import json
    root = os.path.join(os.path.dirname(pkg.__file__), "..")
    glue = os.path.join(os.path.dirname(pkg.__file__), "host", "gadget_glue.c")
    obj = str(tmp_path / "gadget_glue.o")
    cmd = ["gcc", "-c", "-O0", "-Wall", "-Wextra", "-Werror", "-DNGRAVS_BUILD_INSIDE_REFERENCE", "-DDOUBLEPRECISION",
           "-DUNEQUALSOFTENINGS", "-I" + os.path.join(root, "tests", "glue_stub"), "-I" + os.path.join(root, "include")] + flags + \
          [glue, "-o", obj]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    nm = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtDdBb"}
    macros = {f[2:].split("=")[0] for f in flags}
    req = json.load(open(os.path.join(root, "tests", "golden", "glue_required_symbols.json")))["required"]
    assert len(req) >= 16
    missing = []
    for name, rec in req.items():
        on = True
        for g in rec["guards"]:
            m = g.replace("#ifdef", "").strip()
            assert g.startswith("#ifdef"), g      # the only guard form on these definitions
            on = on and m in macros
        if on and name not in defined:
            missing.append("%s (%s, called from %s)" % (name, rec["defined"], rec["used_by"][0]))
    assert not missing, "gadget_glue.o does not define: " + ", ".join(missing)
    # and it must not define what a KEPT unit defines (duplicate symbols at link time)
    undefined = {ln.split()[-1] for ln in subprocess.run(["nm", "-u", obj], capture_output=True, text=True).stdout.splitlines()}
    for kept_symbol in ("endrun", "second", "timediff", "do_box_wrapping", "get_random_number"):
        assert kept_symbol not in defined
    assert {"endrun", "ngravs_create", "ngravs_gravity_tree"} <= undefined
    if "-DNGRAVS_WITH_RCCL" in flags:      # the exchanges go through the C RCCL communicator, MPI only broadcasts the id
        assert {"ngravs_rccl_create", "ngravs_rccl_fill", "MPI_Bcast"} <= undefined and "MPI_Isend" not in undefined


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
