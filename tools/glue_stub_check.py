#!/usr/bin/env python3
"""Do the hand-written declarations the glue is test-compiled against (tests/glue_stub/{allvars,proto,ngravs}.h) still say what
the reference's headers say?

gadget_glue.c is compiled in the CPU suite against TEST-ONLY stubs of the reference's interface (the reference's own headers need
GSL and FFTW-2, which this image lacks).  Nothing tied the stubs to the headers they stand for: a field with another type, two
fields in another order (struct particle_data travels as raw bytes in the glue's particle exchange) or a prototype with another
signature would compile and test green.  This script reads the reference's headers ONLY FOR NAMES AND TYPES and writes, for
every struct field, global and prototype the stubs declare, what the reference declares under that name:

  python tools/glue_stub_check.py [/root/reference] > tests/golden/glue_stub_check.json

It runs in the build container (/root/reference is not on the GPU box); the output is data (names, types, array extents, the
order of struct fields, reference line numbers).  tests/test_host_glue.py::test_stubs_declare_what_the_reference_declares
parses the stubs again and compares them with the committed JSON: same type and extent for every field, stub fields in the
reference's relative order, same return and parameter types for every prototype.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "glue_stub")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def norm_type(t):
    t = re.sub(r"\bextern\b", "", t)
    return " ".join(t.replace("*", " * ").split())


def split_decl(stmt):
    """'long long a, *b[3]' -> ('long long', [('a', '', ''), ('b', '*', '[3]')]) or None"""
    parts = [x.strip() for x in stmt.split(",")]
    m = re.match(r"^(.*?)(\**)\s*([A-Za-z_][A-Za-z0-9_]*)\s*((?:\[[^\]]*\])*)$", parts[0])
    if not m or not m.group(1).strip():
        return None
    typ = norm_type(m.group(1))
    decls = [(m.group(3), m.group(2), m.group(4).replace(" ", ""))]
    for d in parts[1:]:
        dm = re.match(r"^(\**)\s*([A-Za-z_][A-Za-z0-9_]*)\s*((?:\[[^\]]*\])*)$", d)
        if not dm:
            return None
        decls.append((dm.group(2), dm.group(1), dm.group(3).replace(" ", "")))
    return typ, decls


def struct_fields(text, name):
    """[(field, type, extents, line)] of `struct name { ... }` in declaration order; preprocessor lines inside are skipped
    (both branches of an #ifdef are listed: the stubs are compared name by name)"""
    m = re.search(r"struct\s+" + re.escape(name) + r"\s*\{", text)
    if not m:
        return []
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = text[m.end(): i - 1]
    base_line = text[: m.end()].count("\n") + 1
    out = []
    pos = 0
    for stmt in body.split(";"):
        line = base_line + body[:pos].count("\n") + stmt[: len(stmt) - len(stmt.lstrip())].count("\n")
        pos += len(stmt) + 1
        s = " ".join(ln for ln in stmt.split("\n") if not ln.strip().startswith("#")).strip()
        if not s or "{" in s or "}" in s or "(" in s:
            continue
        sd = split_decl(s)
        if not sd:
            continue
        for fname, stars, ext in sd[1]:
            out.append((fname, norm_type(sd[0] + " " + stars), ext, line))
    return out


def globals_(text):
    """{name: (type, extents, line)} of file-scope `extern type name[..];` declarations (not inside a struct)"""
    out = {}
    depth = 0
    for no, ln in enumerate(text.split("\n"), 1):
        if depth == 0:
            m = re.match(r"^\s*extern\s+(.+?);\s*$", ln)
            if m and "(" not in ln and "{" not in ln:
                sd = split_decl(m.group(1).strip())
                if sd:
                    for gname, stars, ext in sd[1]:
                        out[gname] = (norm_type(sd[0] + " " + stars), ext, no)
        depth += ln.count("{") - ln.count("}")
    return out


def prototypes(text):
    """{name: (return type, [parameter types], line)}"""
    out = {}
    flat = text
    for m in re.finditer(r"^[ \t]*((?:[A-Za-z_][A-Za-z0-9_]*[ \t\*]+)+)([A-Za-z_][A-Za-z0-9_]*)[ \t]*\(([^;{}()]*)\)[ \t]*;", flat, flags=re.M):
        ret, name, params = norm_type(m.group(1)), m.group(2), m.group(3)
        if ret.split()[0] in ("return", "else", "typedef"):
            continue
        ptypes = []
        for prm in params.split(","):
            prm = prm.strip()
            if prm in ("", "void"):
                continue
            pm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)?\s*((?:\[[^\]]*\])*)$", prm)
            t = pm.group(1).strip() if pm and pm.group(1).strip() else prm      # "int" alone is a type without a name
            if pm and not pm.group(1).strip():
                t = prm
            ptypes.append(norm_type(t + (" *" if pm and pm.group(3) else "")))
        out[name] = (ret, ptypes, flat[: m.start()].count("\n") + 1)
    return out


def read(path):
    return strip_comments(open(path, errors="replace").read())


STRUCTS = {"allvars.h": ["global_data_all_processes", "particle_data"]}


def describe(ref_dir):
    """what the reference declares under every name the stubs declare"""
    out = {"reference": "names, types, extents and line numbers only", "structs": {}, "globals": {}, "prototypes": {}, "missing_in_reference": []}
    for hdr in ("allvars.h", "proto.h", "ngravs.h"):
        stub, ref = read(os.path.join(STUB, hdr)), read(os.path.join(ref_dir, hdr))
        for sname in STRUCTS.get(hdr, []):
            rf = struct_fields(ref, sname)
            rmap = {}
            for k, f in enumerate(rf):   # a field declared under several #ifdef branches: all its forms, the first one's place
                if f[0] in rmap:
                    rmap[f[0]][4].append([f[1], f[2]])
                else:
                    rmap[f[0]] = (f[1], f[2], f[3], k, [[f[1], f[2]]])
            want = {}
            for f in struct_fields(stub, sname):
                if f[0] in rmap:
                    t, ext, line, order, forms = rmap[f[0]]
                    want[f[0]] = {"type": t, "extent": ext, "order": order, "line": "%s:%d" % (hdr, line), "forms": forms}
                else:
                    out["missing_in_reference"].append("%s: struct %s field %s" % (hdr, sname, f[0]))
            out["structs"][sname] = want
        rg = globals_(ref)
        for name, (t, ext, _) in globals_(stub).items():
            if name in rg:
                out["globals"][name] = {"type": rg[name][0], "extent": rg[name][1], "line": "%s:%d" % (hdr, rg[name][2])}
            elif name not in ("All", "P"):
                out["missing_in_reference"].append("%s: global %s" % (hdr, name))
        rp = prototypes(ref)
        for name, (ret, params, _) in prototypes(stub).items():
            if name in rp:
                out["prototypes"][name] = {"returns": rp[name][0], "parameters": rp[name][1], "line": "%s:%d" % (hdr, rp[name][2])}
            else:
                out["missing_in_reference"].append("%s: prototype %s" % (hdr, name))
    return out


def stub_view():
    """the stubs parsed the same way (the test compares this with the committed JSON)"""
    out = {"structs": {}, "globals": {}, "prototypes": {}}
    for hdr in ("allvars.h", "proto.h", "ngravs.h"):
        stub = read(os.path.join(STUB, hdr))
        for sname in STRUCTS.get(hdr, []):
            out["structs"][sname] = [(f[0], f[1], f[2]) for f in struct_fields(stub, sname)]
        for name, (t, ext, _) in globals_(stub).items():
            out["globals"][name] = (t, ext)
        for name, (ret, params, _) in prototypes(stub).items():
            out["prototypes"][name] = (ret, params)
    return out


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    json.dump(describe(ref), sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")
