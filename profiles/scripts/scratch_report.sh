#!/bin/bash
# Lists every kernel of the library that the compiler gave scratch memory (spills), with its register counts.
# usage: bash profiles/scripts/scratch_report.sh [extra hipcc flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R/py_psnode_amd/csrc; mkdir -p /tmp/scratch_report
for f in psnode_mfma psnode_mfma_h128 psnode_mfma_h32 psnode_backward psnode_dae_backward psnode_latent64_bwd psnode_latent_bwd psnode_latent64 psnode_latent psnode_latent_dpp psnode_rows psnode_backward_wide psnode_dae_backward_wide psnode_generic psnode_generic_bwd psnode_loss; do
  FORM="-mllvm -amdgpu-mfma-vgpr-form"; case $f in psnode_dae_backward|psnode_latent64_bwd) FORM="";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $FORM -I../../include -I. "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/scratch_report/$f.o 2>&1 | python3 -c "
import sys,re,subprocess
name=None; rec={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: name=m.group(1); rec[name]={}
    for key in ('VGPRs:','AGPRs','ScratchSize','Occupancy','LDS Size'):
        if key in line and name: rec[name][key.strip(':')]=line.split(':')[-1].split('[')[0].strip()
n_k=len(rec); bad=0
for n,r in rec.items():
    if int(r.get('ScratchSize','0'))>0:
        bad+=1
        d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
        print('  $f', re.sub(r'psnode::\(anonymous namespace\)::','',d).split('(')[0][:80], r)
print('$f: %d kernels, %d with scratch' % (n_k,bad))
" &
done; wait
