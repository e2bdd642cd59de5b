#!/bin/bash
# round 4: K9 two-role -- chain-only timing (gradient waves without their outer products), kernel stats
R=$GRAFT_REPO_ROOT; cd $R
PSNODE_LIB_PATH=build/var_k9abl/lib.so python profiles/scripts/train_step_models.py dae02 2>&1 | grep -v amdgpu | sed 's/^/chain-only: /'
python profiles/scripts/train_step_models.py dae02 2>&1 | grep -v amdgpu
bash profiles/scripts/r04aa_dae02_train.sh r04ac 2>&1 | grep -A12 "calls" | cut -c1-140
