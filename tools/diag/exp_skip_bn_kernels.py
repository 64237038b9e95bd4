"""Timing experiment (results are wrong on purpose): the c2 training bench with the BatchNorm finalize / backward-coefficient launches removed,
to price the 74 five-microsecond kernels on the step's dependency chain.  usage: python tools/diag/exp_skip_tiny.py [0|1] -- bench args"""
import sys, runpy
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[2]))
skip = sys.argv[1] == "1"
sys.argv = ["bench.py"] + sys.argv[2:]
if skip:
    from pssr2_amd import ops
    # the two eager warm-up steps run the real kernels (scale / shift / coefficients keep plausible values); the captured step has none
    real_f, real_c, n = ops.bn_finalize, ops.bn_bwd_coefs, [0, 0]
    def fin(*a, **k):
        n[0] += 1
        if n[0] <= 74: real_f(*a, **k)
    def coef(*a, **k):
        n[1] += 1
        if n[1] <= 74: real_c(*a, **k)
    ops.bn_finalize, ops.bn_bwd_coefs = fin, coef
runpy.run_path(str(__import__('pathlib').Path(__file__).resolve().parents[2] / 'bench.py'), run_name='__main__')
