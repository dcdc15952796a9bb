#!/bin/bash
# rocprofv3 counter passes of one command (program after "--", no wrappers): two SQ passes, FETCH_SIZE, WRITE_SIZE.
# usage: bash tools/pmc_passes.sh <out_dir> <env assignments or ""> -- python3 <script> args...
OUT=$1; shift
ENVS=$1; shift
shift   # the "--"
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  env $ENVS true
  ( export $ENVS; rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 )
done
find $OUT -name "*counter_collection.csv" | sort
