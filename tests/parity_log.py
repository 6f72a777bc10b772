"""Record achieved parity errors (test name, tensor, error vs fp64 oracle / golden, tolerance) as JSON lines."""
import json
import os


def record(test, what, err, tol, ref_noise=None, note=""):
    path = os.environ.get("G6D_PARITY_LOG")
    if not path:
        return
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "tensor": what, "err": float(err), "tol": float(tol),
                                "ref_fp32_noise": None if ref_noise is None else float(ref_noise), "note": note}) + "\n")
    except OSError:
        pass
