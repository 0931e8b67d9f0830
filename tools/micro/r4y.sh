#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wprod" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "merged or t8_train or tfc" 2>&1 | tail -1
cat > /tmp/wp.py <<'PY'
import sys, os
R=os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, R); sys.path.insert(0, R+'/videotransformer-pytorch_amd')
import torch, vtx
from vtx import ops
a=torch.randn(768,768,device='cuda'); b=torch.randn(768,768,device='cuda'); x=torch.randn(768,device='cuda'); z=torch.randn(768,device='cuda')
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('wprod 768^3 nn+y %.1f us' % t(lambda: ops.wprod(a,b,x=x,z=z,beta_z=1.0)), 'tn %.1f us' % t(lambda: ops.wprod(a,b,ta=True)), 'nt %.1f us' % t(lambda: ops.wprod(a,b,tb=True)))
PY
for i in 1 2; do
echo -n "prev: "; VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_prev.so python /tmp/wp.py
echo -n "new:  "; python /tmp/wp.py
done
