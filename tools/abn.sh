#!/bin/bash
# tools/abn.sh <config> <frames> <variant>... -- developer tool: decode/encode kernel times of the built library
# and of several build_var variants, two rounds, on one box
C=$1; F=$2; shift 2
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["decode_ms"], d["encode_ms"], {k:v for k,v in list(d["dec"].items())[:4]}, {k:v for k,v in list(d["enc"].items())[:3]})'
for i in 1 2; do
  echo "base:"; timeout -k 10 200 python tools/ktimes.py $C $F | tail -n 1 | python3 -c "$P" || exit 1
  for V in "$@"; do
    echo "$V:"; JPEG_AMD_LIB=$PWD/build_var/libjpeg_amd_$V.so timeout -k 10 200 python tools/ktimes.py $C $F | tail -n 1 | python3 -c "$P" || exit 1
  done
done
