#!/bin/bash
# round 4, first GPU call: the two fenced defects under discriminator builds + the poison run of the GPU suite
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
for v in old nop sync; do
  PSNODE_LIB_PATH=build/var_$v/lib.so timeout 300 python profiles/scripts/r04_defects.py a 5 > $O/defect_a_$v.txt 2>&1; echo "rc $?" >> $O/defect_a_$v.txt
done
timeout 300 python profiles/scripts/r04_defects.py a 5 > $O/defect_a_intree.txt 2>&1; echo "rc $?" >> $O/defect_a_intree.txt
PSNODE_DEBUG_GIS_NULL=1 timeout 600 python profiles/scripts/r04_defects.py b 2 > $O/defect_b_null.txt 2>&1; echo "rc $?" >> $O/defect_b_null.txt
PSNODE_DEBUG_GIS_NULL=1 PSNODE_POISON=1 timeout 600 python profiles/scripts/r04_defects.py b 2 > $O/defect_b_null_poison.txt 2>&1; echo "rc $?" >> $O/defect_b_null_poison.txt
PSNODE_POISON=1 timeout 600 python profiles/scripts/r04_defects.py b 2 > $O/defect_b_fenced_poison.txt 2>&1; echo "rc $?" >> $O/defect_b_fenced_poison.txt
PSNODE_POISON=1 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_dist.py > $O/pytest_poison.txt 2>&1; echo "rc $?" >> $O/pytest_poison.txt
tail -3 $O/*.txt
