#!/bin/bash
# tools/profile.sh <tag> [extra bench args] -- the rocprofv3 passes behind profiles/rNN_<tag>_*: run on the GPU box,
#   gpurun --timeout 1100 -- 'bash tools/profile.sh r02_a'
# One rocprofv3 run per pass (kernel trace; FETCH_SIZE; WRITE_SIZE; three SQ passes), results under
# gpurun_out/prof_<tag>/; tools/pmc_summary.py and tools/pmc_sq_summary.py turn them into the committed JSONs.
set -o pipefail
TAG=$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp || exit 1
export TMPDIR=/tmp
STATS="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-mjpeg --no-workloads --one-stream $*"
PMC="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-mjpeg --no-workloads --one-stream --host-synth 4 $*"
run() {  # name, limit, rocprofv3 args...
	local name=$1 limit=$2
	shift 2
	echo "== $name" | tee -a "$OUT/log.txt"
	timeout -s KILL "$limit" rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $CMD >> "$OUT/log.txt" 2>&1 || { echo "pass $name failed" | tee -a "$OUT/log.txt"; exit 1; }
}
CMD=$STATS
run stats 300 --kernel-trace --stats
CMD=$PMC
run fetch 240 --pmc FETCH_SIZE
run write 240 --pmc WRITE_SIZE
run sq_insts 240 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
run sq_cycles 240 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
run sq_lds 240 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
find "$OUT" -name "*.csv" | head -40
echo done
