#!/bin/sh
# Builds liblpxjni.so next to liblpx.so.  Needs a JDK (JAVA_HOME or javac on PATH); the build image of this
# repository has none, so this script only runs on an integrator's machine.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
if [ -z "$JAVA_HOME" ]; then
  JAVAC=$(command -v javac || true)
  [ -n "$JAVAC" ] || { echo "no JDK found (set JAVA_HOME)"; exit 2; }
  JAVA_HOME=$(dirname "$(dirname "$(readlink -f "$JAVAC")")")
fi
[ -f "$JAVA_HOME/include/jni.h" ] || { echo "no jni.h under $JAVA_HOME/include (a JRE without headers?)"; exit 2; }
LIBDIR="$HERE/../linear_programming_solver_amd"
[ -f "$LIBDIR/liblpx.so" ] || make -C "$LIBDIR/csrc"
cc -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" "$HERE/lpx_jni.c" \
   -L"$LIBDIR" -llpx -Wl,-rpath,"$LIBDIR" -o "$LIBDIR/liblpxjni.so"
"$JAVA_HOME/bin/javac" -d "$HERE/classes" "$HERE/java/lpsolver/LpxNative.java"
# the drop-in LPSolverGpu needs the reference classes on the class path: LPX_REFERENCE_CLASSES=<dir or jar>
if [ -n "$LPX_REFERENCE_CLASSES" ]; then
  "$JAVA_HOME/bin/javac" -cp "$LPX_REFERENCE_CLASSES:$HERE/classes" -d "$HERE/classes" "$HERE/java/lpsolver/LPSolverGpu.java"
fi
echo "built $LIBDIR/liblpxjni.so and $HERE/classes/lpsolver/LpxNative.class"
