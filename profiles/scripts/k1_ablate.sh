#!/bin/bash
# Timing ablations of the K1 kernel (results are WRONG by construction; only the launch time is read).
# usage: bash profiles/scripts/k1_ablate.sh "0 1 2 3 4 6 7"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc
for v in $1; do
  D=/tmp/abl_$v; mkdir -p $D
  for f in psnode_capi psnode_generic psnode_mfma psnode_latent psnode_rows; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I$R/include -DPSNODE_ABLATE=$v -c $f.hip -o $D/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $D/lib.so
  PSNODE_LIB_PATH=$D/lib.so python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ABLATE $v kernel_ms %.3f' % d['roofline']['kernel_ms'])"
done
