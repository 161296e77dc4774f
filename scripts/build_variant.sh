#!/bin/bash
# build_variant.sh NAME "FLAGS": kajiya_amd/libkajiya_amd_NAME.so = the product library compiled with extra FLAGS, from a scratch copy of the
# sources (the product's objects stay as they are). Select it at run time with KJ_AMD_LIB=kajiya_amd/libkajiya_amd_NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift; shift     # further arguments go to make as they are (e.g. TAA_EXTRA=-f...)
W=/tmp/kj_variant_$NAME
rm -rf $W && mkdir -p $W/kajiya_amd $W/include
cp -r $ROOT/kajiya_amd/csrc $W/kajiya_amd/csrc && cp $ROOT/include/*.h $W/include/
rm -f $W/kajiya_amd/csrc/*.o
make -s -j16 -C $W/kajiya_amd/csrc EXTRA="$FLAGS" "$@" OUT=$ROOT/kajiya_amd/libkajiya_amd_$NAME.so 2>&1 | grep -v "warning\|^ \|note:" | tail -5
ls -la $ROOT/kajiya_amd/libkajiya_amd_$NAME.so
