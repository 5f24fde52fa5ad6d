#!/bin/bash
# tools/mkvar_all.sh <name> [extra hipcc flags] -- developer tool: the whole library rebuilt under extra flags as
# build_var/libjpeg_amd_<name>.so (see tools/mkvar.sh for one translation unit).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/build_var"
OBJS=""
for f in api pixels huffdec huffsync huffenc codestream; do
	EXT=hip; [ "$f" = codestream ] && EXT=cpp
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I"$ROOT/include" "$@" \
		-x hip -c -o "$ROOT/build_var/${f}_$NAME.o" "$ROOT/jpeg_amd/csrc/$f.$EXT" &
	OBJS="$OBJS $ROOT/build_var/${f}_$NAME.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_var/libjpeg_amd_$NAME.so" $OBJS
echo "built build_var/libjpeg_amd_$NAME.so"
