#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of ONE config leg of bench.py (tools/leg_prof.py c2|c4), the
# counters of tools/profile_gpu.sh.  Summarise with
#   S2R_PROF_SETUP=130 S2R_PROF_WARMUP=4 S2R_PROF_STEPS=16 tools/summarize_prof.py gpurun_out/prof_<tag> s2r_render_general_kernel
# (run_config_leg's launch order: 2 periods + 2 + warm-up, the timed steps, 16 launches timed with HIP events).
# usage: tools/profile_leg.sh <tag> c2|c4
set -u
TAG=${1:-leg}; LEG=${2:-c2}
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export S2R_BENCH_SETTLE=0      # (bench.py: no extra warm-up blocks — tools/summarize_prof.py counts launches)
cd /tmp
CMD="python3 /root/repo/tools/leg_prof.py $LEG 16 4"
# (the counter passes serialise kernels: the two-stream fills wait for each other across streams in the kernels, so every pass but the trace runs on one stream)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace pass failed"
export S2R_OVERLAP=0 S2R_FUSED=0
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1 || echo "pmc1 failed"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1 || echo "pmc2 failed"
rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1 || echo "pmc3 failed"
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1 || echo "pmc4 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "write failed"
ls $OUT
