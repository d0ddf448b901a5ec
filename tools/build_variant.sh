#!/bin/bash
# Dev tool: a variant of the library built with extra compile flags, by the product Makefile:
#     tools/build_variant.sh NAME -DFLAG=... -> build_ab/variants/libnhans_NAME.so   (A/B on the GPU box: tools/ab_variant_libs.sh)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$ROOT/build_ab/variants/$name"
make -C "$ROOT/n-hans_amd/csrc" -j4 O="$ROOT/build_ab/variants/$name" TARGET="$ROOT/build_ab/variants/libnhans_$name.so" EXTRA="$*"
ls -la "$ROOT/build_ab/variants/libnhans_$name.so"
