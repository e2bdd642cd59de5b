#!/bin/bash
# round 4: two-role agreement test; K9 (DAE_02 latent backward, hidden 64) SQ breakdown; K3w / K9w numbers
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -k "two_role" 2>&1 | tail -3
bash profiles/scripts/pmc_sq.sh r04z_k9_rk4 latent64_backward --train --workload dae02 --steps 2 --warmup 1 > /dev/null
cat $O/r04z_k9_rk4_pmc_sq.txt | cut -c60-140
(timeout 600 python profiles/scripts/r04s_latent_wide_time.py 2>&1 | grep -v amdgpu.ids; timeout 900 python profiles/scripts/r04q_direct_encode_widths.py 2>&1 | grep -v amdgpu.ids) > $O/r04z_latent_wide.txt; cat $O/r04z_latent_wide.txt
rm -f $O/pmc_r04z_*.log
