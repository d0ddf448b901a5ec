#!/bin/bash
# The DEV=1 library (cycle stamps, NHANS_ABLATE) next to the default one without disturbing it:
# build_ab/libnhans_hip_dev.so, built by the product Makefile (same compiler, flags and ISA check).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_ab/dev"
make -C "$ROOT/n-hans_amd/csrc" -j4 DEV=1 O="$ROOT/build_ab/dev" TARGET="$ROOT/build_ab/libnhans_hip_dev.so" "$@"
ls -la "$ROOT/build_ab/libnhans_hip_dev.so"
