"""BASELINE configs[4] at full size on one GPU (collected last: the big one).

A rank of the 100 M-photon job ends up holding the complete maps and rendering a band of the 4K frame.  The oracle cannot
follow to this size, so the checks are the size-independent ones of tools/c5_scale_check.py: the library's own record
accounting over the photon-index ranges, rrt_photon_query against a brute-force closed-cube search over the exported
records (counts exact, k smallest squared distances bit-exact) and a finite frame."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_c5_per_rank_share_100m_photons_4k(repo):
    spec = importlib.util.spec_from_file_location("c5_scale_check", os.path.join(repo, "tools/c5_scale_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    photons = int(os.environ.get("RRT_TEST_C5_PHOTONS", "100000000"))
    rep = mod.run(photons, 3840, 2160)
    assert rep["photons_emitted"] == photons + max(10_000, photons // 500)
    assert rep["records_regular"] >= photons
    assert rep["gather_queries"] >= 3840 * 2160          # every pixel's first hit queries both maps, plus the vueons' hits
    assert rep["query_map0"]["largest_count"] > 1000      # the queries really were where the photons are
