#!/bin/bash
# Builds tools/ubench/issue_rates3.hip, checks the ISA of every kernel (tools/ubench/check_isa.py) and, on a GPU box,
# runs it.  usage (from the repo root): gpurun -- 'bash tools/ubench/run_issue_rates3.sh > gpurun_out/issue_rates3.txt 2>&1'
set -e
cd "$(dirname "$0")"
mkdir -p _build
hipcc --offload-arch=gfx950 -O2 -w issue_rates3.hip -o _build/issue_rates3 --save-temps=obj
python3 check_isa.py
./_build/issue_rates3
