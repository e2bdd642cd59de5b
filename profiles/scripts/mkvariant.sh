#!/bin/bash
# Build a library variant that differs from the in-tree build in a few translation units only:
#   bash profiles/scripts/mkvariant.sh <name> "<extra flags>" file1.hip [file2.hip ...]
# -> build/var_<name>/lib.so = the in-tree objects (build/obj/) with the named units recompiled under the extra flags.  build/ is git-ignored but
# travels to the GPU box with the snapshot (load it with PSNODE_LIB_PATH=build/var_<name>/lib.so).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/py_psnode_amd/csrc
name=$1; flags=$2; shift 2
D=$R/build/var_$name
mkdir -p $D
AGPR="psnode_dae_backward.hip psnode_latent64_bwd.hip"
objs=()
for o in $R/build/obj/*.o; do
  case $o in *-hip-amdgcn-*) continue;; esac
  b=$(basename $o .o)
  skip=0; for f in "$@"; do [ "$f" = "$b.hip" ] && skip=1; done
  [ $skip = 0 ] && objs+=($o)
done
pids=()
for f in "$@"; do
  form="-mllvm -amdgpu-mfma-vgpr-form"; for a in $AGPR; do [ "$a" = "$f" ] && form=""; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $form -I$R/include -I$C -Wall -Wno-unused-function $flags -c $C/$f -o $D/${f%.hip}.o &
  pids+=($!)
  objs+=($D/${f%.hip}.o)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $D/lib.so
echo "built $D/lib.so ($flags: $*)"
