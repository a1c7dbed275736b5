"""The oracle against the only outputs the reference holds for this path: render/rgb.ppm, render/rgb.png and
render/depth.ppm — scene2 in photon mode at the reference's settings (photon_scene.rs:22-23: 5 M + 10 k photons per
light, 1024^2).  CPU only; about a minute on 8 cores (the photon pass dominates), hence `slow`.

The GPU product goes through the same check in tests/test_gpu_photon.py; the envelope is in tests/golden_envelope.py.
"""
import os

import numpy as np
import pytest

import golden_envelope as ge
import oracle_binding as ob

abi = ob.abi


@pytest.fixture(scope="module")
def oracle_frame(repo):
    cpu, cam = ob.OracleScene.load(os.path.join(repo, "assets/scenes/scene2.txt"))
    assert cpu.mode == abi.MODE_PHOTON and (cam.width, cam.height) == (1024, 1024)
    cpu.pre_render(1)
    rgb, depth, ids = cpu.render(cam)
    st = cpu.stats()
    assert st["photons_emitted"] == 5_010_000 and st["photons_stored_regular"] > 5_000_000
    return ob.quantise(rgb), ob.quantise(depth), ids


@pytest.mark.slow
def test_oracle_scene2_lies_inside_the_reference_noise_envelope(oracle_frame):
    frame, depth, ids = oracle_frame
    from PIL import Image

    dref = np.asarray(Image.open(os.path.join(ge.REPO, "tests/golden/render/depth.ppm")))
    assert np.array_equal(depth, dref)                 # render/depth.ppm (every byte 255: all hits at distance >= 1)
    m = ge.check(frame, ids)
    # measured when the thresholds were set (seed 1): 1.93 / 2.45 / 0.06 % / 2.7 %, objects <= 0.73; whole-image MAE
    # 2.42 vs rgb.ppm and 1.88 vs rgb.png, the two reference runs being 1.87 apart
    assert m["mae_vs_png"] < m["mae_reference_runs"] + 0.1, m


@pytest.mark.slow
def test_envelope_has_teeth(oracle_frame):
    """Defects a whole-image MAE / PSNR bound lets through must fail the envelope; the reference's own runs must pass."""
    frame, _, ids = oracle_frame
    a, b = ge.load_reference_runs()
    for run in (a, b):
        m = ge.evaluate(run.astype(np.uint8), ids, a, b)
        assert not ge.failures(m) and m["max_ratio_rest"] <= 0.5 and m["max_ratio_pil"] <= 0.5
    f = frame.astype(np.float64)
    sky = f[ids == 1].mean(axis=0)
    defects = {
        "shifted by 2 pixels": np.roll(f, 2, axis=1),
        "3 % brighter": np.clip(f * 1.03, 0, 255),
        "coin missing": np.where((ids == 4)[..., None], sky, f),
        "floor shadows lifted": np.where(((ids == 0) & (f.sum(axis=2) < 120))[..., None], np.minimum(f + 15, 255), f),
        "pipe 4/255 darker": np.where((ids == 9)[..., None], np.maximum(f - 4, 0), f),
    }
    for name, img in defects.items():
        m = ge.evaluate(img, ids, a, b)
        assert ge.failures(m), (name, m)
        # ... while the bound this check replaced (whole-image MAE < 3.0 against rgb.ppm) lets the local ones through
        if name in ("coin missing", "pipe 4/255 darker"):
            assert m["mae_vs_ppm"] < 3.0, (name, m["mae_vs_ppm"])
