#!/bin/bash
# tools/ab.sh <variant> [config] [frames] -- developer tool: kernel times of the built library and of
# build_var/libjpeg_amd_<variant>.so taking turns on one box (boxes differ by several per cent)
V=$1; C=${2:-metric_4k_420_q90}; F=${3:-128}
for i in 1 2 3; do
  echo "base:"; timeout -k 10 200 python tools/ktimes.py $C $F | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms'], d['encode_ms'], {k:v for k,v in list(d['dec'].items())[:4]}, {k:v for k,v in list(d['enc'].items())[:2]})" || exit 1
  echo "$V:"; JPEG_AMD_LIB=$PWD/build_var/libjpeg_amd_$V.so timeout -k 10 200 python tools/ktimes.py $C $F | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['decode_ms'], d['encode_ms'], {k:v for k,v in list(d['dec'].items())[:4]}, {k:v for k,v in list(d['enc'].items())[:2]})" || exit 1
done
