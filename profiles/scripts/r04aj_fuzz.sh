#!/bin/bash
# round 4: extended fuzz of the two-role backward kernels (K4f / K7f saved instances at hidden <= 64 against K5; the models incl. DAE_02 / K9 against the fp64 walk)
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r04aj_fuzz.txt; : > $O
for seed in 11 12 13 14 15 16 17 18; do
  echo "== fuzz_backward seed $seed (150 cases)" >> $O
  timeout 900 python profiles/scripts/fuzz_backward.py $seed 150 2>&1 | grep -v amdgpu | tail -3 >> $O
done
for seed in 21 22 23 24; do
  echo "== fuzz_backward seed $seed (100 cases, larger shapes)" >> $O
  FUZZ_BMAX=300 FUZZ_TMAX=40 timeout 900 python profiles/scripts/fuzz_backward.py $seed 100 2>&1 | grep -v amdgpu | tail -3 >> $O
done
for seed in 31 32 33 34; do
  echo "== fuzz_models seed $seed (100 cases)" >> $O
  timeout 1200 python profiles/scripts/fuzz_models.py $seed 100 2>&1 | grep -v amdgpu | tail -3 >> $O
done
echo "== one leg with NaN-poisoned host buffers" >> $O
PSNODE_POISON=1 timeout 900 python profiles/scripts/fuzz_backward.py 41 150 2>&1 | grep -v amdgpu | tail -3 >> $O
PSNODE_POISON=1 timeout 1200 python profiles/scripts/fuzz_models.py 42 100 2>&1 | grep -v amdgpu | tail -3 >> $O
cat $O
