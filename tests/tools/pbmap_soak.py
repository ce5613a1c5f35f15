"""Soak check of the host-side plane code (pbmap_register.h: RegisterPbMap's matcher + closed-form pose, mergePlanes, pool_sensor_planes,
the colour constraints) against the numpy restatement (oracle/pbmap_ref.py): the seeded random-scene tests of tests/test_pbmap_register.py
run again with every seed shifted.  No GPU.  python tests/tools/pbmap_soak.py [n_shifts [first_shift]]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
n_shifts = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
_orig = np.random.default_rng
shift = 0
def _shifted(seed=None, *a, **k):
    return _orig(seed + shift if isinstance(seed, (int, np.integer)) else seed, *a, **k)
np.random.default_rng = _shifted
import test_pbmap_register as T
cases = [("matches_numpy_restatement mode %d" % m, lambda m=m: T.test_matches_numpy_restatement_on_noisy_cluttered_scenes(m)) for m in (0, 1, 2, 3)]
cases += [("merge_planes random sets", T.test_merge_planes_random_sets_match_the_numpy_restatement),
          ("pool_sensor_planes random sets", T.test_pool_sensor_planes_random_sets_match_the_numpy_restatement)]
cases += [("colour restatement mode %d" % m, lambda m=m: T.test_colour_matches_numpy_restatement_on_random_scenes(m)) for m in (0, 2)]
bad = notes = 0
for k in range(n_shifts):
    shift = first + 37 * k
    for name, fn in cases:
        try:
            fn()
        except Exception as e:                   # an assertion of the test: report the draw and go on
            tb = traceback.extract_tb(e.__traceback__)[-1]
            # not comparisons of the library with the restatement: the tests' bound on the distance to the GROUND TRUTH under 4 mm of noise
            # (1 degree / 3 cm), and their check that the scenario mix met enough pooled / dropped cases
            soft = tb.line is not None and ("gt_rot" in tb.line or "pooled_somewhere" in tb.line or "statuses" in tb.line)
            bad += 0 if soft else 1
            notes += 1 if soft else 0
            print("shift %d: %s %s at %s:%d: %s" % (shift, name, "scenario bound (not a comparison with the restatement)" if soft else "FAILED",
                                                    os.path.basename(tb.filename), tb.lineno, str(e)[:200]), flush=True)
print("pbmap soak: %d seed shifts x %d random-scene tests, %d mismatches with the numpy restatement, %d draws outside a scenario bound of the test" % (n_shifts, len(cases), bad, notes))
sys.exit(1 if bad else 0)
