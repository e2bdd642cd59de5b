#!/bin/bash
# Speed + accuracy of K1 build variants.  usage: bash profiles/scripts/k1_variants.sh "name:flags" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  D=/tmp/var_$name; mkdir -p $D
  for f in psnode_capi psnode_generic psnode_mfma psnode_latent psnode_rows; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I$R/include $flags -c $f.hip -o $D/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $D/lib.so
  echo "== $name ($flags)"
  PSNODE_LIB_PATH=$D/lib.so python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   kernel_ms %.3f  value %.4g  frac %.3f' % (d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))"
  PSNODE_LIB_PATH=$D/lib.so python $R/profiles/scripts/accuracy_report.py 2>&1 | sed 's/^/   /'
done
