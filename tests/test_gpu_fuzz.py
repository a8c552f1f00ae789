"""Randomised scenes through the C ABI against the oracle, every tick (GPU).

Each seed draws a forest, bodies (types, shapes, sizes, masses, layers / masks, friction), trigger volumes and a feature set —
ground plane, Bullet's orientation scheme, broadphase, Bullet's sub-step clock — and then a script of edits between ticks:
teleports, velocity sets (linear and angular), re-created bodies, bodies whose type / shape / size / mass / filter change or
that are removed, MarkDirty, re-parenting, trigger volumes added and removed.  Some entities have no Transform (their
children are roots).  After EVERY tick the complete
observable state must equal the oracle's bit for bit: Transforms, dirty flags, world matrices, body state, activation,
contact counts, pair set, trigger events.  The directed tests cover each feature alone; this one covers their combinations.
"""
import os

import numpy as np
import pytest

import banggameengine_amd as B
from banggameengine_amd import synth
from helpers import DT, assert_bits_equal, build_oracle
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _forest(rng, n):
    parent = np.full(n, 0xFFFFFFFF, np.uint32)
    p_child = rng.choice([0.0, 0.5, 0.8, 0.95])
    window = int(rng.choice([3, 40, 400]))
    for i in range(1, n):
        if rng.random() < p_child:
            parent[i] = rng.integers(max(0, i - window), i)
    return parent


# BGE_FUZZ_SEEDS=n [BGE_FUZZ_FIRST=k]: a longer campaign, seeds k .. k + n - 1 (seeds 64.. alternate between the two styles)
_FIRST = int(os.environ.get("BGE_FUZZ_FIRST", "0"))
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + int(os.environ.get("BGE_FUZZ_SEEDS", "64"))))
def test_random_scene_and_edit_script_match_oracle_every_tick(seed, monkeypatch):
    if seed % 5 == 0:
        monkeypatch.setenv("BGE_TRIGGER_GRID_MIN", "0")    # the ghosts look their bodies up in the broadphase grid
    # (islands of more than 128 contact points are solved by a workgroup, level by level; every other seed sends those of more than 8 there)
    monkeypatch.setenv("BGE_ISLAND_BIG_POINTS", "8" if seed % 8 in (1, 2) else "128")
    rng = np.random.default_rng(1000 + seed)
    # seeds 24..39 and 52..63: larger scenes with long parent chains (tiles that overflow into further passes), hundreds of collision
    # filter combinations (the palette's 32-class and 255-class boundaries), physics and transforms as separate calls (the
    # adapter's pattern), normal matrices, several ticks per call, and two long runs in which bodies come to rest and sleep
    style_b = 24 <= seed < 40 or 52 <= seed < 64 or (seed >= 64 and seed % 3 == 2)
    n = int(rng.integers(300, 3000)) if not style_b else int(rng.integers(3000, 9000))
    split = style_b and bool(seed & 1)
    normals = style_b and seed % 4 in (0, 1)
    n_ticks = 40 if seed not in (29, 33) else 330
    ground = bool(seed & 1)
    basis = bool(seed & 2)
    broadphase = bool(seed & 4) or seed % 12 >= 8
    clock = seed % 12 in (3, 6, 9, 11)       # Bullet's stepSimulation accumulator with varying dt
    # round 3: Dynamic boxes collide with the Static / Kinematic box colliders scattered through the scene (with or without the plane);
    # restitution from a generator of its own, so that the seeds of rounds 1-2 draw the scenes they always drew
    static_contacts = seed % 3 != 1
    restitution = np.random.default_rng(7000 + seed).choice([0.0, 0.0, 0.4, 0.9], 1 << 16).astype(np.float32)
    wl = synth.Workload("fuzz", synth.FLAT, n, 500 + seed)
    wl.parent = _forest(rng, n)
    if style_b:
        at = 1
        while at < n - 10:                                   # chain segments up to 700 deep
            length = int(rng.integers(2, rng.choice([8, 80, 700])))
            for i in range(at + 1, min(at + length, n)):
                wl.parent[i] = i - 1
            at += length + int(rng.integers(1, 50))
    side = float(rng.choice([6.0, 20.0, 60.0])) if not style_b else float(rng.choice([25.0, 80.0]))
    wl.pos = rng.uniform(-side, side, (n, 3)).astype(np.float32)
    if ground:
        wl.pos[:, 1] = rng.uniform(0.1, 3.0, n).astype(np.float32)
    wl.euler = rng.uniform(-180.0, 180.0, (n, 3)).astype(np.float32)
    wl.euler[rng.random(n) < 0.2] = 0.0
    wl.scale = np.ones((n, 3), np.float32)
    odd = rng.random(n) < 0.2
    wl.scale[odd] = rng.uniform(0.3, 3.0, (int(odd.sum()), 3)).astype(np.float32)
    wl.body_type = rng.choice([255, 255, 0, 1, 1, 1, 2], n).astype(np.uint8)
    n_bare = int(rng.integers(0, 12))        # the first entities have no Transform: whoever hangs under them is a root
    has_transform = np.ones(n, np.uint8)
    has_transform[:n_bare] = 0
    wl.body_type[:n_bare] = 255
    shape = rng.choice([0, 0, 1], n).astype(np.uint8)
    size = rng.uniform(0.1, 1.2, (n, 3)).astype(np.float32)
    if seed % 3 == 0:
        mass = rng.uniform(0.05, 30.0, n).astype(np.float32)          # more distinct masses than the palette holds
    else:
        mass = rng.choice([0.25, 1.0, 1.0, 3.0, 50.0], n).astype(np.float32)
    layer = rng.choice([1, 1, 2, 4, 8], n).astype(np.uint32)
    mask = rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFD, 0xFFFFFFFB, 3], n).astype(np.uint32)
    if style_b:
        kinds = int(rng.choice([40, 40, 400]))               # (layer, mask) combinations: beyond 32, beyond 255
        layer = (np.uint32(1) << rng.integers(0, 8, n).astype(np.uint32)) | (rng.integers(0, kinds, n).astype(np.uint32) << np.uint32(8))
        mask = np.where(rng.random(n) < 0.8, 0xFFFFFFFF, 0xFFFF00FF | (rng.integers(0, 255, n) << 8)).astype(np.uint32)
    friction = rng.choice([0.5, 0.5, 0.1, 1.0], n).astype(np.float32)
    body_kw = dict(mass=mass, shape=shape, size=size, layer=layer, mask=mask)

    ref = build_oracle(wl, orient_mode=po.ORIENT_BASIS if basis else po.ORIENT_IDEAL, aabbs=broadphase, has_transform=has_transform, **body_kw)
    n_trig = int(rng.integers(0, 9)) if broadphase else 0
    trig = None
    if n_trig:
        te = (n_bare + rng.choice(n - n_bare, n_trig, replace=False)).astype(np.uint32)
        trig = (te, rng.choice([0, 1], n_trig).astype(np.uint8), rng.uniform(0.5, side / 2, (n_trig, 3)).astype(np.float32),
                rng.choice([0, 4, 2], n_trig).astype(np.uint32), rng.choice([0xFFFFFFFF, 1, 6], n_trig).astype(np.uint32),
                (rng.random(n_trig) < 0.3).astype(np.uint8), (rng.random(n_trig) < 0.9).astype(np.uint8))
        for k in range(n_trig):
            ref.AddTriggerVolume(int(te[k]) + 1, int(trig[1][k]), trig[2][k], int(trig[3][k]), int(trig[4][k]), bool(trig[5][k]), bool(trig[6][k]))
    if ground:
        for i in range(n):
            ref.SetFriction(i + 1, float(friction[i]))
        ref.SetGroundPlane(True)
    if static_contacts:
        for i in range(n):
            ref.SetRestitution(i + 1, float(restitution[i]))
        ref.SetStaticContacts(True)
    # round 3, later: Dynamic boxes collide with each other (islands of several bodies) in half of the not-too-dense scenes
    dynamic_contacts = seed % 4 in (1, 2) and side >= 20.0
    if dynamic_contacts:
        if not static_contacts:
            for i in range(n):
                ref.SetRestitution(i + 1, float(restitution[i]))
        ref.SetDynamicContacts(True)
    if clock:
        ref.SetAccumulator(True, DT, 4)
    flags = B.TICK_ALL | (B.TICK_BULLET_BASIS if basis else 0) | (B.TICK_BROADPHASE if broadphase else 0)
    if normals:
        flags |= B.TICK_NORMAL_MATRICES
    vel = (rng.normal(size=(n, 3)) * 3.0).astype(np.float32)

    pair_cap = max(256 * n, 4096)
    with B.World(pair_capacity=pair_cap) as w:
        w.set_topology(wl.parent, has_transform)
        w.upload_trs(wl.pos, wl.euler, wl.scale)
        w.upload_bodies(wl.body_type, **body_kw)
        if trig:
            w.upload_triggers(*trig)
        if ground:
            w.upload_friction(friction)
            w.set_ground_plane(True)
        if static_contacts:
            w.upload_restitution(restitution[:n])
            w.set_static_contacts(True)
        if dynamic_contacts:
            w.upload_restitution(restitution[:n])
            w.set_dynamic_contacts(True)
        parent = wl.parent.copy()
        updates_done = 0                               # PhysicsSystem::Update calls so far
        body_born = np.zeros(1 << 16, np.int64)        # updates_done when the entity's body components were put on a Transform owner:
        #                                                the body is in the world after the NEXT update (EnsureRigidBody)
        kept_body = np.zeros(1 << 16, bool)            # the entity lost its Transform while its body was in the world (it lives on)
        body_edit_pending = np.zeros(1 << 16, bool)    # ... and its body components were edited since
        most_asleep = 0
        gravity_y = -9.81
        history = {}                                   # entity -> [(tick, edit)] for failure messages
        for tick in range(n_ticks):
            # ---- an edit between ticks, mirrored on both sides
            what = rng.choice(["none", "none", "teleport", "velocity", "spin", "recreate", "change", "dirty", "reparent", "triggers", "transform", "gravity", "grow"])
            if n_ticks > 40 and (rng.random() < 0.93 or (tick > 20 and what in ("velocity", "spin"))):
                what = "none"                                  # the long runs are mostly left alone: bodies settle and fall asleep
            if what == "teleport":
                first, cnt = int(rng.integers(n_bare, n - 20)), int(rng.integers(1, 20))
                p = rng.uniform(-side, side, (cnt, 3)).astype(np.float32)
                if ground:
                    p[:, 1] = np.abs(p[:, 1]) * np.float32(0.1) + np.float32(0.2)
                e = rng.uniform(-180.0, 180.0, (cnt, 3)).astype(np.float32)
                ref.bulk_set_trs(first, p, e, None)
                w.upload_trs(p, e, None, first=first)
            elif what in ("velocity", "spin"):
                ang = np.zeros((n, 3), np.float32)      # (the oracle's bulk call zeroes what it is not given)
                if what == "spin":
                    sel = rng.random(n) < 0.3
                    ang[sel] = rng.normal(size=(int(sel.sum()), 3)).astype(np.float32)
                lin = vel * np.float32(rng.uniform(0.2, 2.0))
                ref.bulk_set_velocity(lin, ang)
                w.set_velocities(lin, ang)
            elif what == "recreate":
                first, cnt = int(rng.integers(n_bare, n - 30)), int(rng.integers(1, 30))
                for e in range(first, first + cnt):
                    ref.MarkBodyDirty(e + 1)
                    history.setdefault(e, []).append((tick, "recreate"))
                    body_edit_pending[e] |= not has_transform[e]
                w.upload_bodies(wl.body_type[first:first + cnt], first=first, **{k: v[first:first + cnt] for k, v in body_kw.items()})
            elif what == "change":
                # other components on the same entities: type (incl. none: the body is removed), collider, mass, filter
                first, cnt = int(rng.integers(n_bare, n - 30)), int(rng.integers(1, 30))
                sl = slice(first, first + cnt)
                was_none = wl.body_type[sl] == 255
                wl.body_type[sl] = rng.choice([255, 0, 1, 1, 2], cnt).astype(np.uint8)
                shape[sl] = rng.choice([0, 1], cnt).astype(np.uint8)
                size[sl] = rng.uniform(0.1, 1.2, (cnt, 3)).astype(np.float32)
                mass[sl] = rng.choice([0.5, 2.0, 7.0], cnt).astype(np.float32)
                layer[sl] = rng.choice([1, 2, 4], cnt).astype(np.uint32)
                mask[sl] = rng.choice([0xFFFFFFFF, 0xFFFFFFFD, 5], cnt).astype(np.uint32)
                for e in range(first, first + cnt):
                    history.setdefault(e, []).append((tick, f"change to type {wl.body_type[e]}"))
                    body_edit_pending[e] |= not has_transform[e]
                    if was_none[e - first] and wl.body_type[e] != 255 and has_transform[e]:
                        body_born[e] = updates_done
                    if wl.body_type[e] == 255:
                        ref.RemoveRigidBody(e + 1)
                        kept_body[e] = False
                    else:
                        ref.AddCollider(e + 1, int(shape[e]), size[e])
                        ref.AddRigidBody(e + 1, int(wl.body_type[e]), float(mass[e]), int(layer[e]), int(mask[e]))
                        if ground:
                            ref.SetFriction(e + 1, float(friction[e]))
                        if static_contacts or dynamic_contacts:
                            ref.SetRestitution(e + 1, float(restitution[e]))   # (a component added now starts from the default 0)
                w.upload_bodies(wl.body_type[sl], first=first, **{k: v[sl] for k, v in body_kw.items()})
                if ground:
                    w.upload_friction(friction[sl], first=first)
                if static_contacts or dynamic_contacts:
                    w.upload_restitution(restitution[sl], first=first)
            elif what == "triggers" and trig and rng.random() < 0.5 and has_transform[int(trig[0][0])]:
                # a trigger volume is retuned: other shape / size (the ghost's shape is rebuilt), maybe another layer or mask (the
                # ghost is re-added: its remembered overlaps are dropped), one-shot on or off
                k = 0
                e = int(trig[0][k])
                act = ref.TriggerIsActive(e + 1)
                t1, t2, t3, t4, t5 = (a.copy() for a in trig[1:6])
                t1[k] = rng.choice([0, 1])
                t2[k] = rng.uniform(0.5, side / 2, 3).astype(np.float32)
                if rng.random() < 0.5:
                    t3[k] = rng.choice([0, 4, 2])
                    t4[k] = rng.choice([0xFFFFFFFF, 1, 6])
                t5[k] = rng.random() < 0.3
                ref.AddTriggerVolume(e + 1, int(t1[k]), t2[k], int(t3[k]), int(t4[k]), bool(t5[k]), bool(act))
                history.setdefault(e, []).append((tick, f"retuned: layer {t3[k]} mask {t4[k]:#x} one-shot {t5[k]} active {act}"))
                still_active = np.array([ref.TriggerIsActive(int(x) + 1) for x in trig[0]], np.uint8)
                trig = (trig[0], t1, t2, t3, t4, t5, still_active)
                w.upload_triggers(*trig)
            elif what == "triggers" and trig:
                # one trigger volume goes, another entity gets one
                te = trig[0]
                gone = int(rng.integers(0, len(te)))
                ref.RemoveTriggerVolume(int(te[gone]) + 1)
                fresh = int(rng.integers(n_bare, n))
                while fresh in te:
                    fresh = int(rng.integers(n_bare, n))
                new_size = rng.uniform(0.5, side / 2, 3).astype(np.float32)
                ref.AddTriggerVolume(fresh + 1, 0, new_size, 0, 0xFFFFFFFF, False, True)
                history.setdefault(fresh, []).append((tick, "trigger added"))
                history.setdefault(int(te[gone]), []).append((tick, "trigger removed"))
                keep = [k for k in range(len(te)) if k != gone]
                still_active = w.trigger_active(te[keep]).astype(np.uint8)   # TriggerVolume::active as the system left it (one-shots that fired)
                trig = (np.append(te[keep], np.uint32(fresh)).astype(np.uint32), np.append(trig[1][keep], np.uint8(0)).astype(np.uint8),
                        np.vstack([trig[2][keep], new_size[None, :]]).astype(np.float32), np.append(trig[3][keep], np.uint32(0)).astype(np.uint32),
                        np.append(trig[4][keep], np.uint32(0xFFFFFFFF)).astype(np.uint32), np.append(trig[5][keep], np.uint8(0)).astype(np.uint8),
                        np.append(still_active, np.uint8(1)).astype(np.uint8))
                w.upload_triggers(*trig)
            elif what == "transform":
                # Transforms come and go (RemoveTransform / AddTransform): children of an entity that loses its Transform become
                # roots WITHOUT being marked dirty, a new Transform starts from what is uploaded, and — as the adapter does for
                # a fresh index — the entity's body components are uploaded again
                touched = set()
                for _ in range(int(rng.integers(1, 4))):
                    e = int(rng.integers(0, n))
                    if trig is not None and rng.random() < 0.3:
                        e = int(rng.choice(trig[0]))      # a trigger volume's entity: its ghost stays where it was last posed
                    if e in touched:
                        continue    # (a Transform given back to a body that lived on and taken away again before the next physics
                    touched.add(e)  #  update would have had to leave the body where it was: Transform and body share the position array)
                    history.setdefault(e, []).append((tick, "Transform removed" if has_transform[e] else "Transform added"))
                    if has_transform[e]:
                        # (a RigidBody on the entity stays: the reference keeps stepping the Bullet body of an entity that lost only
                        #  its Transform — EnsureRigidBody returns before it looks at the runtime, PhysicsSystem.cpp:389-393 — and
                        #  so does the world: the body keeps its slot, collides, enters triggers, and is never re-created)
                        ref.RemoveTransform(e + 1)
                        has_transform[e] = 0
                        kept_body[e] = wl.body_type[e] != 255 and updates_done > body_born[e]
                        w.set_topology(parent, has_transform)
                    else:
                        p3 = rng.uniform(-side, side, (1, 3)).astype(np.float32)
                        if ground:
                            p3[:, 1] = np.float32(0.5)
                        e3 = rng.uniform(-180.0, 180.0, (1, 3)).astype(np.float32)
                        s3 = np.ones((1, 3), np.float32)
                        ref.AddTransform(e + 1, p3[0], e3[0], s3[0])
                        has_transform[e] = 1
                        w.set_topology(parent, has_transform)
                        w.upload_trs(p3, e3, s3, first=e)
                        if wl.body_type[e] != 255 and e >= n_bare and (not kept_body[e] or body_edit_pending[e]):
                            # a body that never existed on the device (the entity had no Transform when it was uploaded), or whose
                            # components changed while it had none: the reference (re)creates it now, dirty as it still is
                            w.upload_bodies(wl.body_type[e:e + 1], first=e, **{k: v[e:e + 1] for k, v in body_kw.items()})
                            if ground:
                                w.upload_friction(friction[e:e + 1], first=e)
                            if static_contacts:
                                w.upload_restitution(restitution[e:e + 1], first=e)
                            if not kept_body[e]:
                                body_born[e] = updates_done
                        kept_body[e] = body_edit_pending[e] = False
            elif what == "gravity" and not ground:
                # another gravity (the per-class force table is rebuilt); not with the plane: bodies sleep there, and Bullet's
                # setGravity skips sleepers where this library takes gravity per tick
                gravity_y = float(np.float32(rng.choice([-9.81, -3.0, 0.0, 4.5])))
                ref.SetPhysicsOptions(gravity_y, po.ORIENT_BASIS if basis else po.ORIENT_IDEAL, broadphase)
            elif what == "grow":
                # the scene grows: new entities at the end, some hanging under old ones, some with bodies
                add = int(rng.integers(1, 40))
                new_parent = np.where(rng.random(add) < 0.5, rng.integers(n_bare, n, add), 0xFFFFFFFF).astype(np.uint32)
                parent = np.concatenate([parent, new_parent])
                has_transform = np.concatenate([has_transform, np.ones(add, np.uint8)])
                p3 = rng.uniform(-side, side, (add, 3)).astype(np.float32)
                if ground:
                    p3[:, 1] = rng.uniform(0.2, 2.0, add).astype(np.float32)
                e3 = rng.uniform(-180.0, 180.0, (add, 3)).astype(np.float32)
                s3 = np.ones((add, 3), np.float32)
                bt = rng.choice([255, 0, 1, 1, 2], add).astype(np.uint8)
                grown = dict(mass=rng.choice([0.5, 1.0, 4.0], add).astype(np.float32), shape=rng.choice([0, 1], add).astype(np.uint8),
                             size=rng.uniform(0.1, 1.0, (add, 3)).astype(np.float32), layer=rng.choice([1, 2, 4], add).astype(np.uint32),
                             mask=np.full(add, 0xFFFFFFFF, np.uint32))
                for k in range(add):
                    eid = ref.CreateEntity()
                    assert eid == n + k + 1
                    ref.AddTransform(eid, p3[k], e3[k], s3[k])
                    if new_parent[k] != 0xFFFFFFFF:
                        ref.SetParent(eid, int(new_parent[k]) + 1)
                    if bt[k] != 255:
                        ref.AddCollider(eid, int(grown["shape"][k]), grown["size"][k])
                        ref.AddRigidBody(eid, int(bt[k]), float(grown["mass"][k]), int(grown["layer"][k]), int(grown["mask"][k]))
                        if ground:
                            ref.SetFriction(eid, 0.5)
                ref.n = n + add
                w.set_topology(parent, has_transform)
                w.upload_trs(p3, e3, s3, first=n)
                w.upload_bodies(bt, first=n, **grown)
                if ground:
                    w.upload_friction(np.full(add, 0.5, np.float32), first=n)
                body_born[n:n + add] = updates_done
                wl.body_type = np.concatenate([wl.body_type, bt])
                mass, shape, size, layer, mask = (np.concatenate([body_kw[k], grown[k]]) for k in ("mass", "shape", "size", "layer", "mask"))
                body_kw = dict(mass=mass, shape=shape, size=size, layer=layer, mask=mask)
                friction = np.concatenate([friction, np.full(add, 0.5, np.float32)])
                restitution[n:n + add] = 0.0                          # (the new components' default, on both sides)
                vel = np.concatenate([vel, (rng.normal(size=(add, 3)) * 3.0).astype(np.float32)])
                n += add
                wl.n = n
            elif what == "dirty":
                first, cnt = int(rng.integers(n_bare, n - 50)), int(rng.integers(1, 50))
                for e in range(first, first + cnt):
                    ref.MarkDirty(e + 1)
                w.mark_dirty(first, cnt)
            elif what == "reparent":
                moved = set()
                for _ in range(int(rng.integers(1, 6))):
                    c = int(rng.integers(1, n))
                    if c in moved:
                        continue    # (A -> B -> A inside ONE edit: Scene::SetParent marks the child dirty twice, a topology that ends
                    moved.add(c)    #  where it began shows bge_world_set_topology nothing — seed 5562; the adapter sees the dirty flags)
                    p = int(rng.integers(0, c)) if rng.random() < 0.8 else None     # (a parent below the child: no cycles)
                    parent[c] = 0xFFFFFFFF if p is None else p
                    ref.SetParent(c + 1, 0 if p is None else p + 1)
                w.set_topology(parent, has_transform)

            # ---- the tick
            if clock:
                dt = float(np.float64(rng.choice([0.3, 0.5, 1.0, 1.0, 1.7, 2.5, 5.5])) * np.float64(DT))
                ref.PhysicsSystemUpdate(dt)
                updates_done += 1
                got_n = w.step_simulation(dt, 4, DT, gravity=(0.0, gravity_y, 0.0), flags=flags)
                assert got_n == ref.LastSubSteps(), f"tick {tick}: {got_n} sub-steps, oracle {ref.LastSubSteps()}"
            elif split:
                ref.PhysicsSystemUpdate(DT)
                updates_done += 1
                w.tick(dt=DT, gravity=(0.0, gravity_y, 0.0), flags=flags & ~(B.TICK_TRANSFORMS | B.TICK_NORMAL_MATRICES))
                tf_now = has_transform.astype(bool)
                assert_bits_equal(w.download_pose()[0][tf_now], ref.bulk_pose()[0][tf_now], f"seed {seed} tick {tick}: position after the physics call")
                assert np.array_equal(w.download_dirty()[tf_now], ref.bulk_world()[1].astype(bool)[tf_now]), f"seed {seed} tick {tick}: dirty after the physics call"
                w.tick(dt=DT, flags=B.TICK_TRANSFORMS | (B.TICK_NORMAL_MATRICES if normals else 0))
                got_n = 1
            else:
                reps = 1 if not style_b or trig or rng.random() < 0.7 else int(rng.integers(2, 5))   # several ticks in one call
                for _ in range(reps):
                    ref.PhysicsSystemUpdate(DT)
                    updates_done += 1
                    if _ + 1 < reps:
                        ref.TransformSystemUpdate()
                w.tick(dt=DT, gravity=(0.0, gravity_y, 0.0), flags=flags, ticks=reps)
                got_n = 1
            ref.TransformSystemUpdate()

            # ---- everything observable
            tag = f"seed {seed} tick {tick} (after {what})"
            tf = has_transform.astype(bool)
            pos, euler = w.download_pose()
            rpos, reuler = ref.bulk_pose()
            badp = np.flatnonzero((pos.view(np.uint32) != rpos.view(np.uint32)).any(axis=1) & tf)
            detail = "" if not len(badp) else (f" [entities {badp[:6].tolist()}; entity {badp[0]}: type {wl.body_type[badp[0]]}, sub-steps {got_n}, here {pos[badp[0]].tolist()} "
                                               f"oracle {rpos[badp[0]].tolist()}, velocity here {w.download_bodies()['linvel'][badp[0]].tolist()} oracle "
                                               f"{ref.bulk_bodies()['linvel'][badp[0]].tolist()}, parent {parent[badp[0]]}, born {body_born[badp[0]]} of {updates_done}, "
                                               f"history {history.get(int(badp[0]))}; parent's history {history.get(int(parent[badp[0]])) if parent[badp[0]] != 0xFFFFFFFF else None}]")
            if os.environ.get("BGE_FUZZ_TRACE_ENTITY"):
                te_ = int(os.environ["BGE_FUZZ_TRACE_ENTITY"])
                bade_ = np.flatnonzero((euler.view(np.uint32) != reuler.view(np.uint32)).any(axis=1) & tf)
                if te_ < 0 and len(bade_):
                    te_ = int(bade_[0])          # (-1: the first entity whose rotationEuler differs)
                te_ = max(te_, 0)
                print(f"tick {tick} after {what}: sub-steps {got_n}: entity {te_}: here {pos[te_].tolist()} v {w.download_bodies()['linvel'][te_].tolist()} | oracle {rpos[te_].tolist()} "
                      f"v {ref.bulk_bodies()['linvel'][te_].tolist()} | type {wl.body_type[te_]} Transform {has_transform[te_]} parent {parent[te_]} history {history.get(te_)}"
                      f" | euler here {euler[te_].view(np.uint32).tolist()} oracle {reuler[te_].view(np.uint32).tolist()} | quat here {w.download_bodies()['quat'][te_].view(np.uint32).tolist()} "
                      f"oracle {ref.bulk_bodies()['quat'][te_].view(np.uint32).tolist()} | flags {flags:#x}"
                      f" | state here {w.download_activation()[0][te_]} oracle {ref.bulk_activation()[0][te_]} | plane contacts here {w.download_contacts()[0][te_] if ground else None} "
                      f"oracle {ref.GroundContacts(te_ + 1)[0] if ground else None}"
                      + (f" | pairs here {[r.tolist() for r in w.download_dynamic_pairs()[0] if te_ in r[:2]]} oracle {[r.tolist() for r in ref.DynamicPairs()[0] if te_ + 1 in r[:2]]}" if dynamic_contacts else "")
                      + (f" | boxes here {w.download_box_contacts()[1][te_].tolist()} oracle {[(o - 1, len(r)) for o, r in ref.BoxContacts(te_ + 1)]}" if static_contacts else ""))
            assert_bits_equal(pos[tf], rpos[tf], f"{tag}: position{detail}")
            assert_bits_equal(euler[tf], reuler[tf], f"{tag}: rotationEuler")
            want_world, want_dirty = ref.bulk_world()
            assert_bits_equal(w.download_world()[tf], want_world[tf], f"{tag}: world")
            assert np.array_equal(w.download_dirty()[tf], want_dirty.astype(bool)[tf]), f"{tag}: dirty flags"
            if normals:
                assert_bits_equal(w.download_normal()[tf], po.normal_matrices(want_world)[tf], f"{tag}: normal matrices")
            assert w.dirty_count() == ref.CountDirtyTransforms(), f"{tag}: CountDirtyTransforms"
            rb, gb = ref.bulk_bodies(), w.download_bodies()
            ex = rb["exists"]
            # (the oracle's bulk velocity seeding also writes static / kinematic records; a body that lives on without a Transform
            #  keeps the type it had, whatever its components have been changed to since)
            dyn = ex & (wl.body_type == 1) & ~body_edit_pending[:n]
            bad = np.flatnonzero((gb["linvel"].view(np.uint32) != rb["linvel"].view(np.uint32)).any(axis=1) & dyn)
            detail = "" if not len(bad) else (f" [entity {bad[0]}: type {wl.body_type[bad[0]]}, Transform {has_transform[bad[0]]}, body lives on {kept_body[bad[0]]}, "
                                              f"pending edit {body_edit_pending[bad[0]]}, born {body_born[bad[0]]} of {updates_done}, history {history.get(int(bad[0]))}]")
            assert_bits_equal(gb["linvel"][dyn], rb["linvel"][dyn], f"{tag}: linear velocity{detail}")
            assert_bits_equal(gb["angvel"][dyn], rb["angvel"][dyn], f"{tag}: angular velocity")
            assert_bits_equal(gb["quat"][ex], rb["quat"][ex], f"{tag}: quaternion")
            st, tm = w.download_activation()
            rst, rtm = ref.bulk_activation()
            assert np.array_equal(st[ex], rst[ex].astype(np.uint8)), f"{tag}: activation states"
            most_asleep = max(most_asleep, int((st[dyn] == 2).sum()))
            assert_bits_equal(tm[ex & (rst == 1)], rtm[ex & (rst == 1)], f"{tag}: deactivation timers")
            if ground:
                cn, _ = w.download_contacts()
                for e in np.flatnonzero(dyn)[::5]:
                    rn, _ = ref.GroundContacts(int(e) + 1)
                    assert cn[e] == rn, f"{tag}: body {e} has {cn[e]} ground contacts, oracle {rn}"
            if static_contacts:
                nb, hdr, pts = w.download_box_contacts()
                for e in np.flatnonzero(dyn)[::4]:
                    want = ref.BoxContacts(int(e) + 1)
                    assert nb[e] == len(want), f"{tag}: body {e} has {nb[e]} box manifolds, oracle {len(want)} (history {history.get(int(e))})"
                    for k, (other, rows) in enumerate(want):
                        assert hdr[e, k, 0] == other - 1 and hdr[e, k, 1] == len(rows), f"{tag}: body {e} manifold {k}: {hdr[e, k].tolist()} vs ({other - 1}, {len(rows)})"
                        assert_bits_equal(pts[e, k, :len(rows)], rows, f"{tag}: body {e} manifold {k} with box {other - 1}")
            if dynamic_contacts:
                dh, dp = w.download_dynamic_pairs()
                rh, rp = ref.DynamicPairs()
                rh = rh.copy()
                rh[:, :2] -= 1
                assert dh.shape == rh.shape and np.array_equal(dh, rh), f"{tag}: the pair cache of Dynamic boxes: {len(dh)} pairs here, {len(rh)} in the oracle"
                assert_bits_equal(dp, rp, f"{tag}: points of the pair manifolds")
            if broadphase and got_n > 0:
                bad = np.flatnonzero((gb["aabb"].view(np.uint32) != rb["aabb"].view(np.uint32)).any(axis=1) & ex)
                detail = "" if not len(bad) else (f" [entity {bad[0]}: type {wl.body_type[bad[0]]}, Transform {has_transform[bad[0]]}, body lives on {kept_body[bad[0]]}, "
                                                  f"pending edit {body_edit_pending[bad[0]]}, born {body_born[bad[0]]} of {updates_done}, parent {parent[bad[0]]}]")
                assert_bits_equal(gb["aabb"][ex], rb["aabb"][ex], f"{tag}: fed AABBs{detail}")
                assert np.array_equal(w.pairs(cap=pair_cap), ref.pairs("sweep")), f"{tag}: pair set"
            if trig:
                want_ev = ref.TriggerEvents()
                want_ev[:, 1:] -= 1
                got_ev = w.trigger_events()
                if os.environ.get("BGE_FUZZ_TRACE_TRIGGER"):
                    tr = int(os.environ["BGE_FUZZ_TRACE_TRIGGER"])
                    print(f"tick {tick} after {what}: trigger {tr}: oracle {[tuple(e) for e in want_ev.tolist() if e[1] == tr][:5]} gpu {[tuple(e) for e in got_ev.tolist() if e[1] == tr][:5]} "
                          f"active oracle {ref.TriggerIsActive(tr + 1)} gpu {w.trigger_active(np.array([tr], np.uint32))[0]}")
                if not np.array_equal(got_ev, want_ev):
                    a, b = set(map(tuple, got_ev.tolist())), set(map(tuple, want_ev.tolist()))
                    first = sorted((a - b) | (b - a))[0][1]
                    raise AssertionError(f"{tag}: trigger events: only here {sorted(a - b)[:6]}, only in the oracle {sorted(b - a)[:6]}; "
                                         f"triggers on {trig[0].tolist()}, Transform-less among them {[int(e) for e in trig[0] if not has_transform[e]]}; "
                                         f"trigger {first}: history {history.get(int(first))}, active in the oracle {ref.TriggerIsActive(int(first) + 1)}, "
                                         f"here {w.trigger_active(np.array([first], np.uint32))[0]}, one-shot {trig[5][list(trig[0]).index(first)]}")
        if n_ticks > 40:
            assert most_asleep > 20, f"seed {seed}: only {most_asleep} bodies ever slept in the long run"


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_sharded_over_several_worlds_equals_the_unsharded_oracle(seed):
    """SURVEY 8(e) + 8(f) rank 4 under random scenes: a forest is partitioned by subtree over 2..6 worlds (all on this GPU),
    each world ticks its shard, and (i) every entity's pose and world matrix equals the ONE unsharded oracle's bit for bit,
    (ii) the slab exchange reproduces the unsharded pair set, each pair once — with bodies far wider than a slab, bodies on
    children, static / kinematic bodies and filter masks in the mix."""
    from banggameengine_amd import sharding
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(3000, 20000))
    wl = synth.Workload("fuzz", synth.FLAT, n, 900 + seed)
    wl.parent = _forest(rng, n)
    side = float(rng.choice([15.0, 40.0]))
    wl.pos = rng.uniform(-side, side, (n, 3)).astype(np.float32)
    wl.euler = rng.uniform(-180.0, 180.0, (n, 3)).astype(np.float32)
    wl.scale = np.ones((n, 3), np.float32)
    wl.body_type = rng.choice([255, 0, 1, 1, 1, 2], n).astype(np.uint8)
    size = rng.uniform(0.2, 1.0, (n, 3)).astype(np.float32)
    wide = rng.choice(n, 5, replace=False)
    size[wide] = rng.uniform(5.0, 2.0 * side, (5, 3)).astype(np.float32)        # wider than a slab: routed to several
    kw = dict(size=size, shape=rng.choice([0, 0, 1], n).astype(np.uint8), layer=rng.choice([1, 2, 4], n).astype(np.uint32),
              mask=rng.choice([0xFFFFFFFF, 3, 6], n).astype(np.uint32))
    vel = (rng.normal(size=(n, 3)) * 2.0).astype(np.float32)
    nshards, axis, ticks = int(rng.integers(2, 7)), int(rng.integers(0, 3)), int(rng.integers(2, 6))
    ref = build_oracle(wl, aabbs=True, **kw)
    for k in range(ticks):
        ref.PhysicsSystemUpdate(DT)
        ref.TransformSystemUpdate()
        if k == 0:
            ref.bulk_set_velocity(vel, np.zeros((n, 3), np.float32))
    want_pairs = ref.pairs("sweep")
    want_pos, want_euler = ref.bulk_pose()
    want_world = ref.bulk_world()[0]
    rank_of, load, shards = sharding.shard_scene(wl.parent, nshards)
    worlds = []
    try:
        for ids, local_parent in shards:
            w = B.World(pair_capacity=max(256 * len(ids), 4096))
            w.set_topology(local_parent)
            w.upload_trs(wl.pos[ids], wl.euler[ids], wl.scale[ids])
            w.upload_bodies(wl.body_type[ids], **{k: v[ids] for k, v in kw.items()})
            w.set_global_ids(ids)
            for k in range(ticks):
                w.tick(dt=DT, flags=B.TICK_ALL | B.TICK_AABBS)
                if k == 0:
                    w.set_velocities(vel[ids], np.zeros((len(ids), 3), np.float32))
            worlds.append(w)
            pos, euler = w.download_pose()
            assert_bits_equal(pos, want_pos[ids], f"seed {seed}: positions of a shard")
            assert_bits_equal(euler, want_euler[ids], f"seed {seed}: rotationEuler of a shard")
            assert_bits_equal(w.download_world(), want_world[ids], f"seed {seed}: world matrices of a shard")
        sharding.slab_broadphase_local(worlds, axis=axis)
        got = np.concatenate([w.pairs(cap=max(256 * n, 4096)) for w in worlds])
    finally:
        for w in worlds:
            w.close()
    assert sorted(np.concatenate([ids for ids, _ in shards]).tolist()) == list(range(n))   # every entity lives in exactly one shard
    key = got[:, 0].astype(np.uint64) << np.uint64(32) | got[:, 1]
    assert len(np.unique(key)) == len(key), f"seed {seed}: a pair was reported by two slabs"
    assert len(want_pairs) > 100 and np.array_equal(got[np.argsort(key)], want_pairs), f"seed {seed}: {len(got)} pairs, oracle {len(want_pairs)}"
