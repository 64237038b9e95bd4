"""Is the training step power-limited?  Runs bench.py's timed region (same arguments) with every model parameter set to zero: the kernels,
launch order and byte counts are those of the real step, every activation and gradient is 0 -- no data toggling in the matrix pipes.
tools/diag/microbench_conv.py with MB_ZERO=1 is the per-layer version (identical instruction stream, 10-23 % faster on zeros).
    python tools/diag/exp_zero_params_step.py [bench.py arguments]"""
import runpy
import sys
from pathlib import Path

root = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
import torch                         # noqa: E402
import pssr2_amd.models as M         # noqa: E402


def zeroed(cls):
    class Zeroed(cls):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            with torch.no_grad():
                for p in self.parameters():
                    p.zero_()
    Zeroed.__name__ = cls.__name__
    return Zeroed


M.ResUNet, M.RDResUNet = zeroed(M.ResUNet), zeroed(M.RDResUNet)
sys.argv = [str(root / "bench.py"), "--no-extras", "--no-cpu-baseline"] + sys.argv[1:]
runpy.run_path(str(root / "bench.py"), run_name="__main__")
