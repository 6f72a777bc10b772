import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from gen6d_amd import ops, synth
from gen6d_amd.pipeline import TensorPipeline
dev = torch.device("cuda")
pipe = TensorPipeline(dev); pipe.build()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fulls = synth.imgs_to_tensor(synth.synth_images(4, 480, 640, seed=100)).to(dev)[torch.arange(B) % 4]
crops = synth.imgs_to_tensor(synth.synth_images(4, 128, 128, seed=200)).to(dev)[torch.arange(B) % 4]
ops.SERIAL = True
pipe.query(fulls, crops); pipe.query(fulls, crops); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    pipe.query(fulls, crops); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::") or (ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.cpu_parent is None):
        st = [s for s in (ev.stack or []) if "gen6d_amd" in s or "bench.py" in s]
        cnt[(ev.name, st[0].split("/root/")[-1] if st else "?")] += 1
for (name, where), n in sorted(cnt.items(), key=lambda t: -t[1]):
    if name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::reshape", "aten::expand", "aten::permute", "aten::unsqueeze", "aten::_unsafe_view", "aten::empty_strided", "aten::transpose", "aten::t", "aten::squeeze", "aten::alias", "aten::detach", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::narrow", "aten::is_nonzero", "aten::unbind", "aten::contiguous"):
        continue
    print(n, name, where)
