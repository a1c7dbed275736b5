"""A wider sweep of tests/test_gpu_random_scenes.py than the suite runs: seeds [first, first + count) through both test
functions, failures listed, nothing stops at the first one.  (The photon test also asserts that a scene keeps more than
300 regular records, a sanity check of the generator, not of parity: seeds 152 and 232 keep 256 and 259 and are otherwise
record-for-record identical; so are the other 14 seeds below 1250 that trip it.)  Round 1: seeds 100-1249 clean: 1150
scenes traced and rendered, 288 photon passes, every record and frame as the oracle's; seed 488's frame overflows a HitVec
on both sides (now in the test suite).  End of round 2: seeds 1300-1449 and, with SWEEP_KIND=csgmesh (scene_kit.csg_mesh_scene:
CSG trees with mesh leaves), seeds 8-157: clean but for six scenes that trip only the 300-record sanity check after their maps
compared equal (tools/sweep_debug.py prints the failing line of a case).  Round 3: that sanity check is a pytest.skip in FRONT of
the comparisons; the sweep counts skips apart from failures, and a scene that is compared is compared to the end (frames included).
   usage (GPU box): [SWEEP_KIND=csgmesh] python tools/gpu_random_sweep.py [first [count]]"""
import os, sys, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import test_gpu_random_scenes as T
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
import pytest
bad, skipped = [], []
for seed in range(first, first + count):
    for name in ("test_random_scene_traces_and_frames_match_oracle", "test_random_scene_photon_maps_match_oracle"):
        if name.endswith("photon_maps_match_oracle") and seed % 4:
            continue
        try:
            getattr(T, name)(("csgmesh", seed) if os.environ.get("SWEEP_KIND") == "csgmesh" else seed)  # csgmesh: CSG trees with mesh leaves
        except pytest.skip.Exception as e:
            skipped.append((seed, name))
            print("skip", seed, name, str(e)[:120], flush=True)
        except Exception as e:
            bad.append((seed, name, repr(e)[:300]))
            print("FAIL", seed, name, repr(e)[:300], flush=True)
    if seed % 10 == 9:
        print("... seed", seed, "failures so far", len(bad), flush=True)
print(("FAILED %d" % len(bad) if bad else "ALL OK (%d seeds)" % count) + ", %d skipped (too few photons stored)" % len(skipped))
sys.exit(1 if bad else 0)
