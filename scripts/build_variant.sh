#!/bin/bash
# Builds a variant of libkajiya_amd.so with extra compiler flags into kajiya_amd/<name>.so, out of tree, for A/B runs on the GPU:
#   scripts/build_variant.sh libkajiya_amd_w8.so -DKJ_BVH_WIDTH=8
#   KJ_AMD_LIB=$PWD/kajiya_amd/libkajiya_amd_w8.so python scripts/traversal_microbench.py
# Experiment switches that exist today: -DKJ_BVH_WIDTH=8 (8-wide nodes, measured 24 % slower), -DKJ_BVH_FOLD_INVD (one FMA per plane: the ray's
# reciprocal direction folded into the node's decode scale; unmeasured).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
TMP=$(mktemp -d)
mkdir -p "$TMP/kajiya_amd" && cp -r "$ROOT/kajiya_amd/csrc" "$TMP/kajiya_amd/" && cp -r "$ROOT/include" "$TMP/"
cd "$TMP/kajiya_amd/csrc" && rm -f *.o
sed -i "s/^HIPFLAGS := /HIPFLAGS := $* /; s/^HOSTFLAGS := /HOSTFLAGS := $* /" Makefile
make -s -j8
cp "$TMP/kajiya_amd/libkajiya_amd.so" "$ROOT/kajiya_amd/$NAME"
rm -rf "$TMP"
echo "built kajiya_amd/$NAME with: $*"
