"""helpers shared by the tests"""
import numpy as np


def both_inputs(oracle, amd, **kw):
    """the same parameters as an oracle input and as a product input"""
    return oracle.make_input(**kw), amd.make_input(**kw)


def ulp_diff(a, b):
    """distance in units in the last place between two float64 arrays"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-2**63) - ia[ia < 0]
    ib[ib < 0] = np.int64(-2**63) - ib[ib < 0]
    return np.abs(ia - ib)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / scale) if scale > 0 else float(np.max(np.abs(a - b)))
