cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_timed_gpu.py -x -q -k "resize_bilinear or detector_headline" 2>&1 | tail -3) > gpurun_out/c31_t.log; tail -2 gpurun_out/c31_t.log
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from gen6d_amd import ops
from gen6d_amd.network.detector import Detector
x=torch.rand((16,3,480,640),device='cuda')
sizes=[Detector._scale_size(480,640,s) for s in (0.5,0.0,-0.5,-1.0)]
for _ in range(3): ops.resize_bilinear_pyramid(x,sizes)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.resize_bilinear_pyramid(x,sizes)
e1.record(); torch.cuda.synchronize()
print("pyramid resize, 16 queries: %.1f us"%(e0.elapsed_time(e1)/20*1e3))
PY
