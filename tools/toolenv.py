"""Shared by the measurement tools: `G6D_LIB_PATH=<profiling build .so>` selects the library the package loads (ablation builds of
tools/*ablate*.sh) and `KNOBS="wino_wide=0,w43_chunk_us=2.4"` sets launch-policy knobs (include/gen6d_hip.h) before the first launch.
The package itself reads neither: import this module first."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gen6d_amd import lib  # noqa: E402

if os.environ.get("G6D_LIB_PATH"):
    lib.LIB_PATH = os.environ["G6D_LIB_PATH"]
for item in filter(None, os.environ.get("KNOBS", "").split(",")):
    name, val = item.split("=")
    lib.set_knob(name.strip(), float(val))
# PYSW="gen6d_amd.network.backbone.SPLIT16_TRUNK=0,...": module-level A/B switches of the package (tools only)
for item in filter(None, os.environ.get("PYSW", "").split(",")):
    path, val = item.split("=")
    mod, attr = path.strip().rsplit(".", 1)
    import importlib
    setattr(importlib.import_module(mod), attr, type(getattr(importlib.import_module(mod), attr))(int(val)))
