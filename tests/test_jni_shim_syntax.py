"""The JNI shim (jni/lpx_jni.c) cannot be built here — the image has no JDK — but it can be
type-checked: `cc -fsyntax-only -Werror` against include/lpx.h and a build-authored stub of <jni.h> that declares only
the JNIEnv members the shim uses.  This catches signature drift between lpx.h and the shim (a changed argument list,
a renamed struct field)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "jni", "lpx_jni.c")
STUB = os.path.join(ROOT, "tests", "jni_stub")


@pytest.mark.parametrize("cc", ["gcc", "/opt/rocm/lib/llvm/bin/clang"])
def test_jni_shim_type_checks_against_lpx_h(cc):
    if shutil.which(cc) is None and not os.path.exists(cc):
        pytest.skip(cc + " not available")
    subprocess.check_call([cc, "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", STUB, SHIM])


def test_every_native_method_of_the_java_class_has_a_c_definition():
    """jni/java/lpsolver/LpxNative.java declares the native methods; each needs its Java_lpsolver_LpxNative_* twin."""
    java = open(os.path.join(ROOT, "jni", "java", "lpsolver", "LpxNative.java")).read()
    natives = set(re.findall(r"native\s+[\w\[\]]+\s+(\w+)\s*\(", java))
    c = open(SHIM).read()
    defined = set(re.findall(r"NAME\((\w+)\)", c)) - {"fn"}
    assert natives and natives == defined, (natives ^ defined)


def test_build_script_probes_for_a_jdk():
    sh = open(os.path.join(ROOT, "jni", "build.sh")).read()
    assert "JAVA_HOME" in sh and "jni.h" in sh
