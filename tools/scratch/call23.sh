cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from gen6d_amd import ops, lib
g=torch.Generator().manual_seed(1)
W=(torch.rand((512,32768),generator=g)-0.5).cuda(); b=torch.zeros(512).cuda()
for B in (8,16,32):
    x=(torch.rand((B,32768),generator=g)-0.5).cuda()
    out=torch.empty((B,512),device='cuda')
    for split in (0, 32, 64):
        fn=lambda: ops.conv(x.view(1,1,1,B,32768), W.view(512,1,32768), b, out.view(1,1,1,B,512), out_act=2, split_k=split)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ref=ops.linear_gemv(x,W,b,2)
        print("B",B,"conv split",split,"%.1f us"%(e0.elapsed_time(e1)/20*1e3), "maxdiff", float((out-ref).abs().max()))
PY
