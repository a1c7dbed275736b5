import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
rrt = importlib.import_module("rust-raytracer_amd")
import bench_extra
for it in range(3):
    t0=time.perf_counter()
    scene, cam, ntri = bench_extra.build_mesh_scene(rrt)
    t1=time.perf_counter()
    scene.pre_render(1)
    t2=time.perf_counter()
    print("build_scene (python, add_object) %.1f ms, pre_render %.1f ms, ms_bvh_build %.1f"%((t1-t0)*1e3,(t2-t1)*1e3, scene.stats()["ms_bvh_build"]))
