#!/bin/bash
# Builds the REFERENCE's own CPU NMS (cython/cpu_nms.pyx:17-68) and box-overlap function (cython/bbox.pyx:15-55), both
# unmodified, read where they lie under /root/reference, into oracle/_ref/ -- test infrastructure, build container only
# (oracle/_ref/ is git-ignored and nothing on the GPU box needs it: the vectors they produce are committed under
# tests/golden/).
#
# Toolchain: the .pyx is Cython-0.2x / numpy-1.x era code (`np.int_t` buffers, `np.int` dtype).  The image's main
# interpreter (python3.10, Cython 3.2, numpy 2.2) rejects it at compile time (numpy 2 dropped `int_t` from its .pxd),
# but the image also carries an Anaconda python3.9 with Cython 0.29.24 + numpy 1.26.4, which compiles it as is.
# No reference build system is run (its setup.py also wants nvcc for gpu_nms) and no header / library stand-in is
# written: cython -> gcc on the one file.  The .pyx is symlinked into a scratch directory only so that Cython does
# not take the reference's directory name (`cython/`, which has an __init__.py) as the package name.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${REF:-/root/reference}
PY=${REF_PYTHON:-/opt/conda/bin/python3.9}
OUT="$HERE/_ref"
[ -f "$REF/cython/cpu_nms.pyx" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
[ -x "$PY" ] || { echo "no $PY (Cython 0.29 / numpy 1.x interpreter): oracle/_ref not built"; exit 0; }
mkdir -p "$OUT"
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
NPINC="$("$PY" -c 'import numpy; print(numpy.get_include())')"
PYINC="$("$PY" -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
EXT="$("$PY" -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
for MOD in cpu_nms bbox; do
  [ -f "$REF/cython/$MOD.pyx" ] || continue
  ln -s "$REF/cython/$MOD.pyx" "$TMP/$MOD.pyx"
  "$PY" -m cython -3 "$TMP/$MOD.pyx" -o "$TMP/$MOD.c"
  gcc -O2 -fPIC -shared -fno-fast-math -ffp-contract=off -Wno-cpp -I"$NPINC" -I"$PYINC" "$TMP/$MOD.c" -o "$OUT/$MOD$EXT"
  echo "built $OUT/$MOD$EXT"
done
