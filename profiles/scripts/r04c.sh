#!/bin/bash
mkdir -p gpurun_out/r04c
PSNODE_POISON=1 PSNODE_LIB_PATH=build/var_old/lib.so timeout 300 python profiles/scripts/r04_defect_a_pattern.py > gpurun_out/r04c/a_pattern.txt 2>&1
cat gpurun_out/r04c/a_pattern.txt
