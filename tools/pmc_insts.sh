#!/bin/bash
# Instruction-mix PMC passes for a command (run on the GPU box): tools/pmc_insts.sh <tag> <cmd...>
# Writes gpurun_out/<tag>_{A,B}/..._counter_collection.csv ; summarise with tools/pmc_insts.py
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d gpurun_out/${tag}_A -o ${tag}_A --output-format csv -- "$@" > gpurun_out/${tag}_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS -d gpurun_out/${tag}_B -o ${tag}_B --output-format csv -- "$@" > gpurun_out/${tag}_B.log 2>&1
