cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from gen6d_amd import ops, lib
g=torch.Generator().manual_seed(1)
W=(torch.rand((512,32768),generator=g)-0.5).cuda(); b=torch.zeros(512).cuda()
for B in (2,8,16,32):
    x=(torch.rand((B,32768),generator=g)-0.5).cuda()
    for mf in (1,0):
        lib.set_knob("gemv_mfma", mf)
        for _ in range(3): ops.linear_gemv(x,W,b,2)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.linear_gemv(x,W,b,2)
        e1.record(); torch.cuda.synchronize()
        print("B",B,"mfma",mf,"%.1f us"%(e0.elapsed_time(e1)/20*1e3))
PY
