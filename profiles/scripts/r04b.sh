#!/bin/bash
mkdir -p gpurun_out/r04b
O=gpurun_out/r04b
DEFECT_VERBOSE=1 PSNODE_LIB_PATH=build/var_old/lib.so timeout 300 python profiles/scripts/r04_defects.py a1 4 > $O/a1_old.txt 2>&1
DEFECT_VERBOSE=1 PSNODE_POISON=1 PSNODE_LIB_PATH=build/var_old/lib.so timeout 300 python profiles/scripts/r04_defects.py a1 4 > $O/a1_old_poison.txt 2>&1
timeout 300 python profiles/scripts/r04_defect_b_diff.py > $O/b_diff.txt 2>&1
cat $O/a1_old.txt $O/a1_old_poison.txt $O/b_diff.txt
