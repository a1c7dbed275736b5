import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rrt = importlib.import_module("rust-raytracer_amd")
params = rrt.default_params(photons_per_light=5_000_000, caustic_photons_per_light=10_000)
scene, cam = rrt.Scene.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "assets/scenes/scene2.txt"), params=params)
for i in range(3): scene.pre_render(1 + i)
print(scene.stats()["ms_photon_shoot"], scene.stats()["ms_photon_build"])
