#!/bin/bash
TAG=$1; shift
OUT=/root/repo/gpurun_out/probe2_$TAG; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -- python3 /root/repo/tools/flat_probe.py $* > $OUT/pmc1.log 2>&1
