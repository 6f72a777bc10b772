"""MI355X-native Gen6D inference hot path (see DESIGN.md / INTEGRATION.md)."""
import os as _os

# Several queries are kept in flight on separate streams (TensorPipeline lanes, Gen6DEstimator.predict_many, eval.run_queries).
# ROCm maps a process's streams onto 4 hardware queues by default, where a fourth lane collides with the others (measured on the
# bench: 3 lanes 142.7, 4 lanes 113-135, 4 lanes with 8 queues 147.2 images/s).  The HIP runtime reads the variable when it
# initialises (first device call), so setting it at package import is early enough; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
