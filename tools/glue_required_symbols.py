#!/usr/bin/env python3
"""Which symbols must host/gadget_glue.c define so that the link recipe of INTEGRATION.md closes?

The recipe drops six translation units of the reference (gravtree.o forcetree.o pm_periodic.o domain.o peano.o
gravtree_forcetest.o) and keeps the rest.  This script reads the reference's C files ONLY TO LIST NAMES: every non-static
function (and file-scope object) a dropped unit defines and some kept unit mentions.  It runs in the build container
(/root/reference is not on the GPU box); its output, tests/golden/glue_required_symbols.json, is data -- names and the
reference lines where they are defined / used -- and tests/test_host_glue.py asserts with `nm` that the compiled glue defines
every one of them, for each option set.

usage: python tools/glue_required_symbols.py [/root/reference] > tests/golden/glue_required_symbols.json
"""
import json
import os
import re
import sys

DROPPED = ["gravtree.c", "forcetree.c", "pm_periodic.c", "domain.c", "peano.c", "gravtree_forcetest.c"]
# units of Makefile.reference's OBJS (Makefile.reference:164-171); pm_nonperiodic.c is compiled but ngravs disables it
NOT_UNITS = {"allvars.c"}   # globals live here; it defines no functions

FUNC_DEF = re.compile(r"^(?!static\b)(?:[A-Za-z_][A-Za-z0-9_ \*]*?)\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}]*)\)\s*$")
KEYWORDS = {"if", "for", "while", "switch", "return", "sizeof", "else", "do"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def active_regions(lines):
    """yield (lineno, line, guard) where guard is the innermost #if/#ifdef expression stack as a tuple"""
    stack = []
    for no, ln in enumerate(lines, 1):
        s = ln.strip()
        if s.startswith("#if"):
            stack.append(s)
        elif s.startswith("#else") and stack:
            stack[-1] = "!(" + stack[-1] + ")"
        elif s.startswith("#elif") and stack:
            stack[-1] = s
        elif s.startswith("#endif") and stack:
            stack.pop()
        yield no, ln, tuple(stack)


def definitions(path):
    """non-static functions defined at file scope: name -> (line, guards).  Gadget style: the signature ends a line at column 0
    and the next non-blank line is '{'."""
    lines = strip_comments(open(path, errors="replace").read()).split("\n")
    out = {}
    rows = list(active_regions(lines))
    for k, (no, ln, guard) in enumerate(rows):
        if ln and not ln[0].isspace() and not ln.startswith("#"):
            m = FUNC_DEF.match(ln.rstrip())
            if m and m.group(1) not in KEYWORDS:
                nxt = next((rows[j][1] for j in range(k + 1, min(k + 4, len(rows))) if rows[j][1].strip()), "")
                if nxt.startswith("{"):   # the body opens at column 0 on the next line
                    out.setdefault(m.group(1), (no, [g for g in guard]))
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    units = sorted(f for f in os.listdir(ref) if f.endswith(".c") and f not in NOT_UNITS)
    kept = [u for u in units if u not in DROPPED]
    defs = {}
    for u in DROPPED:
        for name, (line, guard) in definitions(os.path.join(ref, u)).items():
            defs.setdefault(name, {"defined": "%s:%d" % (u, line), "guards": guard, "used_by": []})
    for u in kept:
        text = strip_comments(open(os.path.join(ref, u), errors="replace").read())
        lines = text.split("\n")
        for name, rec in defs.items():
            pat = re.compile(r"\b%s\s*\(" % re.escape(name))
            for no, ln in enumerate(lines, 1):
                if pat.search(ln):
                    rec["used_by"].append("%s:%d" % (u, no))
                    break
    need = {k: v for k, v in sorted(defs.items()) if v["used_by"]}
    json.dump({"recipe_drops": DROPPED, "kept_units": kept,
               "note": "non-static functions the dropped units define and a kept unit calls; names + reference lines only",
               "required": need}, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
