"""Re-runs given cases of tests/test_gpu_random_scenes.py's photon test and prints the failing line.
   usage: python tools/sweep_debug.py 1304 1352 csgmesh:20 ..."""
import os, sys, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_random_scenes as T
for a in sys.argv[1:]:
    case = ("csgmesh", int(a.split(":")[1])) if ":" in a else int(a)
    try:
        T.test_random_scene_photon_maps_match_oracle(case)
        print(case, "ok")
    except Exception:
        tb = traceback.format_exc().strip().splitlines()
        print(case, "|".join(l.strip() for l in tb[-4:])[:500])
