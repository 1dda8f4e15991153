#!/bin/bash
# build a library variant that differs from the default build in ONE object:  bash tools/build_variant.sh <name> <source stem> "<extra hipcc flags>" [TRACE=1]
# (copies the default objects, rebuilds that object with the flags, relinks into ns2vc_amd/lib/variants/<name>/)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; stem=$2; flags=$3; shift 3
make -s -j8 -C "$ROOT/ns2vc_amd/csrc"
V="$ROOT/ns2vc_amd/lib/variants/$name"
mkdir -p "$V/obj"
cp -p "$ROOT"/ns2vc_amd/lib/obj/*.o "$V/obj/"
rm -f "$V/obj/$stem.o"
make -s -C "$ROOT/ns2vc_amd/csrc" OUT="$V" DEFS="$flags" "$@"
ls -la "$V/libns2vc_hip.so" | awk '{print $5, $9}'
