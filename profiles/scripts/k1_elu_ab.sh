R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc
for v in 0 3 4; do D=/tmp/ab_$v; mkdir -p $D
  for f in psnode_capi psnode_generic psnode_mfma psnode_latent psnode_rows; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I$R/include -DPSNODE_ELU=$v -c $f.hip -o $D/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o $D/lib.so; done
for rep in 1 2 3; do for v in 0 3 4; do PSNODE_LIB_PATH=/tmp/ab_$v/lib.so python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $rep ELU=$v kernel_ms %.3f' % d['roofline']['kernel_ms'])"; done; done
