#!/bin/bash
# Dev: run a command once per library variant under variants/ (copied over the in-tree .so for the run).
# usage: scripts/dev_variants.sh <cmd...>
cd "$(dirname "$0")/.."
keep=$(mktemp)
cp biomedkg_amd/csrc/libbiomedkg_hip.so "$keep"
for v in variants/lib_*.so; do
  cp "$v" biomedkg_amd/csrc/libbiomedkg_hip.so
  echo "== $v"
  "$@"
done
cp "$keep" biomedkg_amd/csrc/libbiomedkg_hip.so
