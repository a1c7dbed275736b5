"""rrt_trace_noalloc (the entry point that allocates nothing) against rrt_trace, through the C ABI.

Two calls in a row on the same GPU, the second with a smaller input: the static scratch array still holds the first
call's rays and results, none of which may leak into the second call's outputs.  Bar: bit-exact (integer ids, and f32
results of the same kernel on the same inputs).
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scene_kit as kit

rrt = ob.rrt
abi = rrt._abi
pytestmark = pytest.mark.gpu

FIELDS = ("object_id", "prim_id", "distance", "position", "normal", "uv")


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def test_trace_noalloc_equals_trace_on_two_calls_in_a_row(repo):
    path = os.path.join(repo, "assets/scenes/scene2.txt")       # planes, cuboids, CSG, a mesh, a sphere
    gpu, _ = rrt.Scene.load(path, mode_override=abi.MODE_SCENE)
    gpu.pre_render(1)
    o1, d1 = kit.random_rays(abi.NOALLOC_MAX_RAYS, 11, origin_scale=6.0, towards=(0.0, 2.0, 5.0), spread=5.0)   # the maximum
    o2, d2 = kit.random_rays(5000, 12, origin_scale=6.0, towards=(0.0, 2.0, 5.0), spread=5.0)                  # smaller, other rays
    want1, want2 = gpu.trace(o1, d1), gpu.trace(o2, d2)          # the existing entry point
    assert (want1["object_id"] != abi.MISS).mean() > 0.15
    got1 = gpu.trace(o1, d1, noalloc=True)
    got2 = gpu.trace(o2, d2, noalloc=True)
    for k in FIELDS:
        assert same_bits(got1[k], want1[k]), k
        assert same_bits(got2[k], want2[k]), k
    # outputs the caller does not ask for are skipped, the others unchanged; one ray; a third call after the small one
    import ctypes as C

    ids = np.full(7, 123, np.uint32)
    o3, d3 = np.ascontiguousarray(o1[:7]), np.ascontiguousarray(d1[:7])
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    rc = rrt.lib.rrt_trace_noalloc(gpu._h, o3.ctypes.data_as(f32p), d3.ctypes.data_as(f32p), 7, ids.ctypes.data_as(u32p), None, None, None, None, None)
    assert rc == 0 and np.array_equal(ids, want1["object_id"][:7])
    # the oracle agrees with what both entry points returned
    cpu, _ = ob.OracleScene.load(path, mode_override=abi.MODE_SCENE)
    c2 = cpu.trace(o2, d2)
    for k in FIELDS:
        assert same_bits(got2[k], c2[k]), k


def test_trace_noalloc_rejects_what_it_cannot_hold(repo):
    gpu, _ = rrt.Scene.load(os.path.join(repo, "assets/scenes/quadratics.txt"), mode_override=abi.MODE_SCENE)
    o, d = kit.random_rays(abi.NOALLOC_MAX_RAYS + 1, 3)
    with pytest.raises(rrt.RrtError) as e:                       # call order first, as rrt_trace
        gpu.trace(o[:10], d[:10], noalloc=True)
    assert e.value.code == abi.ERR_STATE
    gpu.pre_render(1)
    with pytest.raises(rrt.RrtError) as e:
        gpu.trace(o, d, noalloc=True)
    assert e.value.code == abi.ERR_LIMIT and "65536" in str(e.value)
    ok = gpu.trace(o[:-1], d[:-1], noalloc=True)                 # exactly the maximum is fine
    want = gpu.trace(o[:-1], d[:-1])
    for k in FIELDS:
        assert same_bits(ok[k], want[k]), k
    empty = gpu.trace(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), noalloc=True)
    assert len(empty["object_id"]) == 0
