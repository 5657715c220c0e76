#!/bin/bash
# Build libbiomedkg_hip.so (gfx950 only) in-tree.  Usage: [JOBS=N] [EXTRA_FLAGS=...] build.sh
#
# Incremental: every object carries a depfile (hipcc -MD -> obj/<name>.d) listing the source and EVERY header
# it pulled in (common.h, rowpiece.h, gemm_epilogue.h, gemm_tile_bf16.h, include/biomedkg_hip.h, ...).  An object
# is rebuilt when it or its depfile is missing, when anything named in the depfile is newer or gone, when this
# script is newer, or when the compile flags differ from the ones recorded beside it (obj/<name>.flags) -- so a
# header edit can never leave a stale object in the .so (tests/test_build_deps.py).
set -e
cd "$(dirname "$0")"
JOBS=${JOBS:-4}
HIPCC=${HIPCC:-hipcc}
OBJ=${OBJ_DIR:-obj}
OUT=${OUT_SO:-libbiomedkg_hip.so}
FLAGS="${EXTRA_FLAGS} --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
mkdir -p "$OBJ"

needs_build() {  # $1 source, $2 object
  local o=$2 d=${2%.o}.d fl=${2%.o}.flags
  [ -f "$o" ] && [ -f "$d" ] && [ -f "$fl" ] || return 0
  [ "$(cat "$fl")" = "$FLAGS" ] || return 0
  [ build.sh -nt "$o" ] && return 0
  # depfile: "obj/x.o: a.hip b.h \<newline> c.h ..."; system headers under /opt/rocm are skipped
  local dep
  for dep in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$d"); do
    case "$dep" in /opt/*|/usr/*) continue ;; esac
    [ -e "$dep" ] || return 0
    [ "$dep" -nt "$o" ] && return 0
  done
  return 1
}

pids=()
rebuilt=()
for f in *.hip; do
  o=$OBJ/${f%.hip}.o
  if needs_build "$f" "$o"; then
    rebuilt+=("${f%.hip}")
    ( $HIPCC $FLAGS -MD -MF "${o%.o}.d" -c "$f" -o "$o" && printf '%s' "$FLAGS" > "${o%.o}.flags" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge $JOBS ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
# objects whose source is gone must not stay in the library
for o in "$OBJ"/*.o; do
  [ -f "$(basename "${o%.o}").hip" ] || rm -f "$o" "${o%.o}.d" "${o%.o}.flags"
done
if [ ${#rebuilt[@]} -gt 0 ] || [ ! -f "$OUT" ]; then
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
fi
echo "rebuilt: ${rebuilt[*]:-(nothing)}"
echo "built $(pwd)/$OUT"
