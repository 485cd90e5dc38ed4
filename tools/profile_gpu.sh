#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats plus PMC passes for the render kernel.
# Output: gpurun_out/prof_<tag>/{trace,pmc1,pmc2,pmc_fetch,pmc_write}/...csv  — summarise with
# tools/summarize_prof.py and copy the summaries into profiles/ (bench.py quotes profiles/r03/c3_summary.json).
# usage: tools/profile_gpu.sh <tag> [bench.py args...]
set -u
TAG=${1:-run}; shift || true
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export S2R_BENCH_SETTLE=0      # (bench.py: no extra warm-up blocks — tools/summarize_prof.py counts launches)
cd /tmp
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-config-legs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 /root/repo/bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
# (the counter passes serialise kernels; the two-stream fills wait for each other across streams inside the kernels: one stream for these passes)
export S2R_OVERLAP=0 S2R_FUSED=0
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc1 -- python3 /root/repo/bench.py $ARGS > $OUT/pmc1.log 2>&1 || echo "pmc1 failed"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 /root/repo/bench.py $ARGS > $OUT/pmc2.log 2>&1 || echo "pmc2 failed"
rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/pmc3 -- python3 /root/repo/bench.py $ARGS > $OUT/pmc3.log 2>&1 || echo "pmc3 failed"
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc4 -- python3 /root/repo/bench.py $ARGS > $OUT/pmc4.log 2>&1 || echo "pmc4 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 /root/repo/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 /root/repo/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "write failed"
ls -R $OUT | head -40
