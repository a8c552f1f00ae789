#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the CPU oracle.

Provenance (read before trusting a fixture):
  * bx_kat.json        — NOT generated here.  Known-answer vectors copied from SURVEY.md §8(a) (made during
                         the survey from the disassembly of the reference's committed MSVC objects).  The
                         oracle must reproduce them; this script only verifies that and refuses to run otherwise.
  * demo_scene.json    — inputs are the reference's own assets/scenes/demo.json:48-108 (3 entities: TRS,
                         collider, rigid body — data, not code); expected world matrices come from the oracle.
  * transform_cases.npz, physics_cases.npz, pairs_case.npz, sleep_case.npz — seeded inputs + oracle outputs ("spec-derived";
                         the reference has no tests, so nothing of its own pins these — parity unpinned).

Run from the repo root:  python oracle/tools/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po  # noqa: E402
from banggameengine_amd import synth  # noqa: E402
from helpers import DT, build_oracle, parent_i32, run_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hexrow(a):
    return ["0x%08x" % int(v) for v in np.asarray(a, np.float32).view(np.uint32)]


def check_kat():
    kat = json.load(open(os.path.join(GOLD, "bx_kat.json")))
    for case in kat["mtxSRT"]:
        got = po.mtx_srt(case["scale"], case["euler"], case["translation"])
        want = np.array([int(x, 16) for x in case["expect_bits"]], np.uint32)
        assert np.array_equal(got.view(np.uint32), want), case["name"]
    for case in kat["mtxMul"]:
        a = np.array([int(x, 16) for x in case["a_bits"]], np.uint32).view(np.float32)
        b = np.array([int(x, 16) for x in case["b_bits"]], np.uint32).view(np.float32)
        want = np.array([int(x, 16) for x in case["expect_bits"]], np.uint32)
        assert np.array_equal(po.mtx_mul(a, b).view(np.uint32), want), case["name"]
    print("bx_kat.json reproduced bit for bit")


def demo_scene():
    # assets/scenes/demo.json:48-108 (values copied as data)
    ents = [
        dict(id="cj", position=[0.0, 7.0, -5.0], rotationEuler=[0.0, 0.0, 0.0], scale=[0.05, 0.05, 0.05]),
        dict(id="ground", position=[0.0, -0.01, 0.0], rotationEuler=[0.0, 0.0, 0.0], scale=[0.05, 1, 0.05],
             collider=dict(shape="box", size=[50.0, 1.0, 50.0]),
             rigidBody=dict(type="Static", layer=1, mask=4294967295)),
        dict(id="checkpoint", position=[5.0, 1.0, 5.0], rotationEuler=[0.0, 0.0, 0.0], scale=[1.0, 1.0, 1.0]),
    ]
    sc = po.RefScene()
    for e in ents:
        i = sc.CreateEntity()
        sc.AddTransform(i, e["position"], e["rotationEuler"], e["scale"])
        if "collider" in e:
            sc.AddCollider(i, 0, e["collider"]["size"])
            sc.AddRigidBody(i, po.BODY_STATIC, 0.0, 1, 0xFFFFFFFF)
    sc.PhysicsSystemUpdate(DT)
    sc.TransformSystemUpdate()
    out = dict(source="assets/scenes/demo.json:48-108 (inputs); expected = CPU oracle", entities=[])
    for k, e in enumerate(ents):
        t = sc.GetTransform(k + 1)
        out["entities"].append(dict(e, expect_world_bits=hexrow(t["world"])))
    json.dump(out, open(os.path.join(GOLD, "demo_scene.json"), "w"), indent=1)


def transform_cases():
    rng = np.random.default_rng(20251031)
    cases = {}

    def forest(n, p_child, window):
        parent = np.full(n, 0xFFFFFFFF, np.uint32)
        for i in range(1, n):
            if rng.random() < p_child:
                parent[i] = rng.integers(max(0, i - window), i)
        return parent

    specs = {
        "flat": np.full(600, 0xFFFFFFFF, np.uint32),
        "chains4": synth.parents(synth.CHAINS4, 0, 1024),
        "subtree64": synth.parents(synth.SUBTREE64, 0, 640),
        "forest": forest(3000, 0.85, 40),
        "deep_chain": np.concatenate([[0xFFFFFFFF], np.arange(0, 599)]).astype(np.uint32),
        "wide_root": np.concatenate([[0xFFFFFFFF], np.zeros(700)]).astype(np.uint32),
    }
    for name, parent in specs.items():
        n = len(parent)
        pos, euler, scale = synth.trs(0xC0FFEE + n, 0, n)
        has_tf = np.ones(n, np.uint8)
        if name == "forest":
            has_tf[rng.choice(n, 60, replace=False)] = 0  # parents without a Transform: children become roots
        sc = po.RefScene().bulk_build(parent_i32(parent), pos, euler, scale, has_transform=has_tf)
        sc.TransformSystemUpdate()
        world, dirty = sc.bulk_world()
        cases[name] = dict(parent=parent, has_tf=has_tf, pos=pos, euler=euler, scale=scale, world=world)
    flat = {}
    for name, c in cases.items():
        for k, v in c.items():
            flat[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "transform_cases.npz"), **flat)


def physics_cases():
    out = {}
    for name, n, ticks in (("flat10k", 512, 120), ("chains4", 512, 60), ("subtree64", 640, 60)):
        wl = synth.config(name, n=n)
        ref = run_oracle(build_oracle(wl), wl, ticks)
        world, _ = ref.bulk_world()
        pos, euler = ref.bulk_pose()
        out[f"{name}.ticks"] = np.int64(ticks)
        out[f"{name}.n"] = np.int64(n)
        out[f"{name}.world"] = world
        out[f"{name}.pos"] = pos
        out[f"{name}.euler"] = euler
        out[f"{name}.linvel"] = ref.bulk_bodies()["linvel"]
    # spinning bodies: exponential-map orientation + euler write-back every tick
    wl = synth.config("flat10k", n=256)
    angvel = (synth.velocity(777, 0, 256) * np.float32(3.0)).astype(np.float32)
    ref = run_oracle(build_oracle(wl), wl, 30, angvel=angvel)
    world, _ = ref.bulk_world()
    pos, euler = ref.bulk_pose()
    out.update({"spin.angvel": angvel, "spin.world": world, "spin.pos": pos, "spin.euler": euler,
                "spin.quat": ref.bulk_bodies()["quat"], "spin.ticks": np.int64(30), "spin.n": np.int64(256)})
    np.savez_compressed(os.path.join(GOLD, "physics_cases.npz"), **out)


def pairs_case():
    n = 2000
    wl = synth.Workload("cube", synth.FLAT, n, 0xBA5E0004, pos_box=synth.CUBE)
    wl.pos = (wl.pos * np.float32(16.0 / 262.0)).astype(np.float32)  # dense: ~ 1 pair per body
    ref = build_oracle(wl, aabbs=True)
    run_oracle(ref, wl, 3)
    pairs = ref.pairs("sweep")
    assert np.array_equal(pairs, ref.pairs("brute"))
    np.savez_compressed(os.path.join(GOLD, "pairs_case.npz"), pos=wl.pos, euler=wl.euler, scale=wl.scale, vel=wl.vel,
                        pairs=pairs, aabb=ref.bulk_bodies()["aabb"], ticks=np.int64(3))
    print("pairs_case:", len(pairs), "pairs for", n, "bodies")


def _sleep_scene():
    """96 Dynamic bodies in zero gravity with speeds around Bullet's sleeping thresholds (0.8 linear, 1.0 angular)."""
    n = 96
    wl = synth.config("flat10k", n=n)
    speed = np.tile(np.array([0.0, 0.3, 0.79, 0.8, 0.81, 1.4], np.float32), n // 6)
    d = synth.velocity(0x51EE9, 0, n)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    vel = (d * speed[:, None]).astype(np.float32)
    angvel = np.zeros((n, 3), np.float32)
    angvel[::4, 1] = np.float32(0.99)
    angvel[1::8, 2] = np.float32(1.0)
    return wl, vel, angvel


def sleep_case():
    wl, vel, angvel = _sleep_scene()
    ref = build_oracle(wl)
    ref.SetPhysicsOptions(0.0, po.ORIENT_IDEAL, False)
    out = dict(vel=vel, angvel=angvel, n=np.int64(wl.n))
    ticks = (238, 239, 240, 241, 260)
    for k in range(ticks[-1] + 1):
        ref.PhysicsSystemUpdate(DT)
        ref.TransformSystemUpdate()
        if k == 0:
            ref.bulk_set_velocity(vel, angvel)
        if k in ticks:
            st, tm = ref.bulk_activation()
            out[f"state.{k}"] = st.astype(np.uint8)
            out[f"time.{k}"] = np.where(st == 1, tm, 0).astype(np.float32)
            out[f"pos.{k}"] = ref.bulk_pose()[0]
            out[f"linvel.{k}"] = ref.bulk_bodies()["linvel"]
    out["ticks"] = np.array(ticks, np.int64)
    np.savez_compressed(os.path.join(GOLD, "sleep_case.npz"), **out)
    print("sleep_case: states at tick 260:", np.bincount(out["state.260"], minlength=5).tolist())


if __name__ == "__main__":
    check_kat()
    demo_scene()
    transform_cases()
    physics_cases()
    pairs_case()
    sleep_case()
    print("fixtures written to", GOLD)
