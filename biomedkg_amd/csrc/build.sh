#!/bin/bash
# Build libbiomedkg_hip.so (gfx950 only) in-tree.  Usage: build.sh [-j N]
set -e
cd "$(dirname "$0")"
JOBS=${JOBS:-4}
FLAGS="${EXTRA_FLAGS} --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ ../../include/biomedkg_hip.h -nt "$o" ]; then
    ( hipcc $FLAGS -c "$f" -o "$o" ) &
    pids+=($!)
    if [ ${#pids[@]} -ge $JOBS ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o libbiomedkg_hip.so obj/*.o
echo "built $(pwd)/libbiomedkg_hip.so"
