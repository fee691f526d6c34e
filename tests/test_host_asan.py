"""CPU: the pure-C pieces of the multi-task host layer (host/ngravs_host.c: top-tree rounds, cut, import request, tree from its
child table) under AddressSanitizer + UndefinedBehaviorSanitizer -- 40 random particle sets (uniform and clumped, 1-8 tasks,
1-3 species, depth limits), tools/host_asan/harness.c.  The device-side entry points the file calls are stubbed (never reached)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_under_sanitizers(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "gadget-2.0.7-ngravs_amd", "host", "ngravs_host.c")
    text = open(src).read()
    # stubs for everything the file calls but does not define (the library's device-side entry points)
    called = set(re.findall(r"\b(ngravs_(?!host_)[a-z0-9_]+)\s*\(", text)) | {"ngravs_host_toptree_borrow"}
    stubs = tmp_path / "stubs.c"
    stubs.write_text("#include <stdlib.h>\n" + "".join("int %s() { abort(); }\n" % f for f in sorted(called)))
    exe = tmp_path / "harness"
    cmd = ["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c99", "-w", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "host_asan", "harness.c"), str(stubs), src, "-o", str(exe), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr or "sanitize" in b.stderr):
        pytest.skip("no sanitizer runtime: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr
