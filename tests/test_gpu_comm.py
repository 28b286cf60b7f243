"""The collective entry points of the C-ABI (sh_comm_init_all / sh_bcast_weights / sh_gather_landmarks):
several GPUs driven from ONE process through the library's own RCCL binding.  A one-GPU box runs the group of one (the calls go
through RCCL all the same); with more devices visible the same test shards a cohort over two of them."""
import ctypes
import os

import numpy as np
import pytest

from conftest import BONES
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.dist import LocalGroup, shard_bounds
from shoulder_amd.engine import Engine, ShoulderHipError
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu


def n_devices():
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0


def cohort(n):
    verts, faces = load_stl(os.path.join(BONES, "humerus_left.stl"))
    T = synth.similarity_transforms(n, verts, seed=4242)
    return [(synth.apply_similarity(T[i], verts), faces) for i in range(n)]


def lane(device, weights):
    e = Engine(device)
    e.load_rfc()
    e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
    e.set_params(unet_dtype=_lib.UNET_BF16)
    return e


@pytest.mark.parametrize("rows", [0, 512])
def test_group_broadcasts_the_parameters_and_gathers_the_records(unet_weights, rows):
    world = 2 if n_devices() >= 2 else 1
    print("sh_comm_* ran as a group of %d device(s)%s" % (world, "" if world > 1 else " -- the n > 1 legs (grouped send / recv, multi-rank broadcast) were NOT exercised"))
    meshes = cohort(6)
    # the records one context computes alone, with the real parameters
    ref = lane(0, unet_weights)
    ref.set_record_rows(rows)
    ref.upload(meshes)
    want = ref.run()
    # the group: only the root holds the real parameters, the others a block of the same shape with other values
    engines = [lane(r, unet_weights if r == 0 else np.zeros_like(unet_weights)) for r in range(world)]
    g = LocalGroup(engines)
    g.bcast_weights(root=0)
    for r, e in enumerate(engines):
        s, c = shard_bounds(len(meshes), world, r)
        e.set_record_rows(rows)
        e.upload(meshes[s:s + c])
    for e in engines:
        e.submit(fetch=False)
    for e in engines:
        e.collect()
    got = g.gather_landmarks()
    assert got.dtype == want.dtype and len(got) == len(meshes)
    assert (got["status"] == 0).all()
    assert got.tobytes() == want.tobytes()
    # the receivers' host copies follow the block (sh_param_block_commit): a later repack uses the broadcast parameters
    for e in engines[1:]:
        ptr, nbytes = e.param_block()
        assert nbytes == ref.param_block()[1]
    for e in engines + [ref]:
        e.close()


def test_group_calls_reject_what_is_not_the_group(unet_weights):
    a = lane(0, unet_weights)
    b = lane(0, unet_weights)
    L = a.L
    arr = (ctypes.c_void_p * 1)(a.h)
    # before sh_comm_init_all
    assert L.sh_bcast_weights(arr, 1, 0) == -3 and b"sh_comm_init_all" in L.sh_last_error(a.h)
    out = np.zeros(1, _lib.LANDMARKS_DTYPE)
    assert L.sh_gather_landmarks(arr, 1, out.ctypes.data) == -3
    # two contexts of one device are lanes, not ranks
    two = (ctypes.c_void_p * 2)(a.h, b.h)
    assert L.sh_comm_init_all(two, 2) == -1 and b"same device" in L.sh_last_error(a.h)
    assert L.sh_comm_init_all(arr, 0) == -1
    g = LocalGroup([a])
    with pytest.raises(ShoulderHipError):
        g.bcast_weights(root=1)                     # root out of range
    with pytest.raises(ShoulderHipError):
        g.gather_landmarks()                        # no run to report
    a.close(); b.close()
