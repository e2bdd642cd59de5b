#!/bin/bash
# Interleaved A/B of K1 micro-variants on one box: "name:flags" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc
for spec in "$@"; do name=${spec%%:*}; flags=${spec#*:}; D=/tmp/vab_$name; mkdir -p $D
  for f in psnode_capi psnode_generic psnode_mfma psnode_latent psnode_rows; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I$R/include $flags -c $f.hip -o $D/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $D/lib.so; done
for rep in 1 2; do for spec in "$@"; do name=${spec%%:*}
  PSNODE_LIB_PATH=/tmp/vab_$name/lib.so python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep %-12s kernel_ms %.3f' % ('$name', d['roofline']['kernel_ms']))"; done; done
