#!/bin/bash
# Timing ablation of K3a (latent hidden 16): PSNODE_ABLATE=8 removes the per-step input loads (results WRONG; only the time is read).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc
for v in 0 8; do
  D=/tmp/abl3_$v; mkdir -p $D
  for f in *.hip; do
    FORM="-mllvm -amdgpu-mfma-vgpr-form"; case $f in psnode_dae_backward.hip|psnode_latent64_bwd.hip) FORM="";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FORM -I$R/include -DPSNODE_ABLATE=$v -c $f -o $D/${f%.hip}.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $D/lib.so
  PSNODE_LIB_PATH=$D/lib.so python $R/bench.py --workload ode02_latent16 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ABLATE $v kernel_ms %.3f' % d['roofline']['kernel_ms'])"
done
