#!/bin/bash
# usage (GPU box): tools/pmc_probe.sh <tag> <voices> <frames> <flat> [lanes]
TAG=$1; shift
OUT=/root/repo/gpurun_out/probe_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 /root/repo/tools/flat_probe.py $* > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 /root/repo/tools/flat_probe.py $* > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 /root/repo/tools/flat_probe.py $* > $OUT/pmc2.log 2>&1
