import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rrt = importlib.import_module("rust-raytracer_amd")
for photons in (1_000_000, 5_000_000, 20_000_000, 100_000_000):
    params = rrt.default_params(photons_per_light=photons, caustic_photons_per_light=max(10_000, photons // 500))
    scene, cam = rrt.Scene.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "assets/scenes/scene2.txt"), params=params)
    print("photons", photons, flush=True)
    scene.pre_render(1)
    cam.width, cam.height = 512, 512
    scene.render(cam)
    st = scene.stats()
    print("   candidates per query pair %.0f, gathered per query pair %.0f" % (st["gather_candidates"] / (st["gather_queries"] / 2), st["photons_gathered"] / (st["gather_queries"] / 2)), flush=True)
