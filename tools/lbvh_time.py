import sys, time, importlib, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench_extra as bx
rrt = importlib.import_module("rust-raytracer_amd")
for rep in range(2):
    scene, cam, n = bx.build_mesh_scene(rrt, bvh_builder=1)
    t0 = time.time(); scene.pre_render(1); print("pre_render %.1f ms, bvh %.1f ms" % ((time.time() - t0) * 1e3, scene.stats()["ms_bvh_build"]), flush=True)
