#!/bin/bash
# tools/mkvar.sh <name> <file.hip> [extra hipcc flags] -- developer tool: a variant of the library with ONE
# translation unit rebuilt under extra flags (experiments: -DX_...), as build_var/libjpeg_amd_<name>.so.
# Run it with JPEG_AMD_LIB=build_var/libjpeg_amd_<name>.so python tools/ktimes.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
mkdir -p "$ROOT/build_var"
OBJS=""
for f in api pixels huffdec huffsync huffenc codestream; do
	if [ "$f" = "$SRC" ]; then
		EXT=hip; [ "$f" = codestream ] && EXT=cpp
		/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I"$ROOT/include" "$@" \
			-x hip -c -o "$ROOT/build_var/${f}_$NAME.o" "$ROOT/jpeg_amd/csrc/$f.$EXT"
		OBJS="$OBJS $ROOT/build_var/${f}_$NAME.o"
	else
		OBJS="$OBJS $ROOT/jpeg_amd/csrc/$f.o"
	fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_var/libjpeg_amd_$NAME.so" $OBJS
echo "built build_var/libjpeg_amd_$NAME.so"
