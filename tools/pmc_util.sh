#!/bin/bash
# tools/pmc_util.sh -- how full the waves are: one rocprofv3 counter pass (run on the GPU box:
#   gpurun -- 'bash tools/pmc_util.sh'), SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) per kernel =
# the share of lanes that take part in a vector instruction.  Output under gpurun_out/prof_util/.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $ROOT/gpurun_out/prof_util
timeout -s KILL 200 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $ROOT/gpurun_out/prof_util -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-mjpeg --no-workloads --one-stream --host-synth 4 > $ROOT/gpurun_out/prof_util.log 2>&1
echo rc $?
