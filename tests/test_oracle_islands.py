"""Behaviour of the oracle's Dynamic-box-against-Dynamic-box contacts and multi-body simulation islands (oracle/island_ref.h,
physics_ref.h CollideDynamicPairs / StepIsland; CPU only).  What the reference's btDiscreteDynamicsWorld does for two Dynamic
boxes of a scene (src/physics/PhysicsSystem.cpp:122-131, 421-474): they collide, rest on each other, exchange momentum, and
bodies whose AABBs overlap sleep and wake TOGETHER."""
import numpy as np
import pytest

from oracle import pyoracle as po

DT = 1.0 / 120.0
ACTIVE, SLEEPING, WANTS = 1, 2, 3


def _scene(boxes, gravity=-9.81, plane=True, mode=po.ORIENT_IDEAL, dynamic=True, ground_box=False):
    """boxes: list of dicts pos, size (half extents), euler, mass, friction, restitution"""
    ref = po.RefScene()
    ref.SetPhysicsOptions(gravity, mode, False)
    ids = []
    n = 0
    if ground_box:
        g = ref.CreateEntity()
        ref.AddTransform(g, (0.0, -0.5, 0.0), (0, 0, 0), (1, 1, 1))
        ref.AddCollider(g, 0, (50.0, 0.5, 50.0))
        ref.AddRigidBody(g, po.BODY_STATIC, 0.0, 1, 0xFFFFFFFF)
        ref.SetFriction(g, 1.0)
        n += 1
    for b in boxes:
        e = ref.CreateEntity()
        ref.AddTransform(e, b["pos"], b.get("euler", (0, 0, 0)), (1, 1, 1))
        ref.AddCollider(e, 0, b.get("size", (0.5, 0.5, 0.5)))
        ref.AddRigidBody(e, po.BODY_DYNAMIC, b.get("mass", 1.0), b.get("layer", 1), b.get("mask", 0xFFFFFFFF))
        ref.SetFriction(e, b.get("friction", 0.5))
        ref.SetRestitution(e, b.get("restitution", 0.0))
        ids.append(e)
        n += 1
    ref.SetGroundPlane(plane)
    ref.SetStaticContacts(ground_box)
    ref.SetDynamicContacts(dynamic)
    ref.n = n
    return ref, ids


def _run(ref, ticks):
    for _ in range(ticks):
        ref.PhysicsSystemUpdate(DT)
        ref.TransformSystemUpdate()


def _y(ref, e):
    return float(ref.GetTransform(e)["position"][1])


@pytest.mark.parametrize("mode", [po.ORIENT_IDEAL, po.ORIENT_BASIS])
def test_a_box_dropped_on_a_box_rests_on_it_and_both_fall_asleep_together(mode):
    ref, (low, top) = _scene([dict(pos=(0, 0.5, 0)), dict(pos=(0.1, 2.0, 0.05), euler=(0.3, 0.0, 0.0))], mode=mode)
    states, carried = [], []
    for k in range(900):
        ref.PhysicsSystemUpdate(DT)
        ref.TransformSystemUpdate()
        st, _ = ref.bulk_activation()
        states.append((int(st[low - 1]), int(st[top - 1])))
        if 150 <= k < 250:
            carried.append(float(ref.DynamicPairs()[1][0, :, 10].sum()))
    assert abs(_y(ref, low) - 0.5) < 0.01 and abs(_y(ref, top) - 1.5) < 0.02
    hdr, pts = ref.DynamicPairs()
    assert hdr.tolist() == [[low, top, 4]]                               # one pair, body A the lower entity, a full face manifold
    assert np.allclose(pts[0, :, 6:9], [0, -1, 0], atol=1e-3)            # the normal on B (the upper box) points at A, down
    assert abs(np.mean(carried) * 120 - 9.81) < 0.3                      # settled, still awake: the pair carries the upper box's weight
    n_low, rows_low = ref.GroundContacts(low)
    assert n_low == 4 and abs(rows_low[:, 3].sum() * 120 - 2 * 9.81) < 2.0  # the plane carries both (the last solved step, still settling)
    assert states[-1] == (SLEEPING, SLEEPING)
    # the lower box was at rest long before the upper one: it waited (WANTS_DEACTIVATION) until the whole island could sleep
    first_sleep = next(k for k, s in enumerate(states) if SLEEPING in s)
    assert states[first_sleep] == (SLEEPING, SLEEPING)
    assert any(s == (WANTS, ACTIVE) for s in states[:first_sleep])


def test_head_on_collision_conserves_momentum():
    """No gravity, no plane: two boxes of mass 1 and 3 fly at each other.  Every impulse is applied to both bodies of its row with
    opposite sign, so m1 v1 + m2 v2 stays what it was (up to rounding), and with restitution 0 they do not separate faster than
    they met."""
    ref, (a, b) = _scene([dict(pos=(-1.5, 0, 0), mass=1.0), dict(pos=(1.5, 0.2, 0.1), mass=3.0)], gravity=0.0, plane=False)
    _run(ref, 1)
    ref.SetVelocity(a, (2.0, 0.0, 0.0))
    ref.SetVelocity(b, (-1.0, 0.0, 0.0))
    p0 = 1.0 * 2.0 + 3.0 * -1.0
    touched = False
    for _ in range(240):
        _run(ref, 1)
        bodies = ref.bulk_bodies()
        v = bodies["linvel"]
        assert abs(1.0 * v[a - 1, 0] + 3.0 * v[b - 1, 0] - p0) < 1e-4
        assert np.abs(1.0 * v[a - 1, 1:] + 3.0 * v[b - 1, 1:]).max() < 1e-4
        hdr, _ = ref.DynamicPairs()
        touched = touched or (len(hdr) == 1 and hdr[0, 2] > 0)
    assert touched
    v = ref.bulk_bodies()["linvel"]
    assert v[b - 1, 0] - v[a - 1, 0] > -1e-3                             # no longer approaching
    assert v[a - 1, 0] < 1.0                                             # the light one was stopped or thrown back


def test_restitution_between_two_dynamic_boxes_is_the_product():
    def separation_speed(r1, r2):
        ref, (a, b) = _scene([dict(pos=(-1.0, 0, 0), restitution=r1), dict(pos=(1.0, 0, 0), restitution=r2)], gravity=0.0, plane=False)
        _run(ref, 1)
        ref.SetVelocity(a, (1.0, 0.0, 0.0))
        ref.SetVelocity(b, (-1.0, 0.0, 0.0))
        _run(ref, 200)
        v = ref.bulk_bodies()["linvel"]
        return float(v[b - 1, 0] - v[a - 1, 0])
    # (restitution 0 still separates them at erp2 x penetration / dt = 0.2 x (2 / 120) x 120 = 0.4: Bullet's positional correction)
    assert separation_speed(0.9, 0.0) < 0.45
    assert separation_speed(0.9, 0.9) > 1.2                              # 0.81 x the closing speed of 2


def test_bodies_that_never_meet_take_the_free_body_path_bit_for_bit():
    rng = np.random.default_rng(5)
    boxes = [dict(pos=(4.0 * k, float(rng.uniform(0.6, 3.0)), 0.0), euler=tuple(rng.uniform(-1, 1, 3)), mass=float(rng.choice([0.5, 2.0])))
             for k in range(12)]
    out = []
    for dynamic in (False, True):
        ref, ids = _scene(boxes, dynamic=dynamic)
        _run(ref, 400)
        b = ref.bulk_bodies()
        out.append((ref.bulk_pose(), b["linvel"].copy(), b["angvel"].copy(), ref.bulk_activation()))
        if dynamic:
            assert len(ref.DynamicPairs()[0]) == 0
    for x, y in zip(out[0], out[1]):
        if isinstance(x, tuple):
            for u, v in zip(x, y):
                assert np.array_equal(np.asarray(u).view(np.uint32), np.asarray(v).view(np.uint32))
        else:
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_a_tower_of_five_stands_and_sleeps_and_a_thrown_box_wakes_the_island():
    tower = [dict(pos=(0.0, 0.5 + 1.0 * k, 0.0), friction=0.8) for k in range(5)]
    bullet = dict(pos=(-6.0, 2.5, 0.0), size=(0.3, 0.3, 0.3), mass=2.0)
    ref, ids = _scene(tower + [bullet], mode=po.ORIENT_BASIS)
    _run(ref, 500)
    st, _ = ref.bulk_activation()
    assert all(st[e - 1] == SLEEPING for e in ids)
    for k, e in enumerate(ids[:5]):
        p = ref.GetTransform(e)["position"]
        assert abs(p[1] - (0.5 + k)) < 0.02 and abs(p[0]) < 0.02, p
    assert len(ref.DynamicPairs()[0]) == 4                                # neighbours only
    # re-create the small box (the reference's way to wake a body: RigidBody.dirty) and throw it at the tower's third box
    ref.SetTRS(ids[5], pos=(-2.0, 2.6, 0.0))
    ref.MarkBodyDirty(ids[5])
    _run(ref, 1)
    ref.SetVelocity(ids[5], (8.0, 1.0, 0.0))
    woke = False
    for _ in range(120):
        _run(ref, 1)
        st, _ = ref.bulk_activation()
        woke = woke or all(st[e - 1] != SLEEPING for e in ids[:5])
    assert woke                                                           # the whole tower left ISLAND_SLEEPING at once
    _run(ref, 1500)
    st, _ = ref.bulk_activation()
    assert all(st[e - 1] == SLEEPING for e in ids)                        # and everything came to rest again
    assert max(abs(ref.GetTransform(e)["position"][0]) for e in ids[:5]) > 0.05   # the tower was hit


def test_a_sleeping_body_woken_by_its_island_gets_no_gravity_in_that_call_and_stays_slow_wants_deactivation():
    """btDiscreteDynamicsWorld::applyGravity runs once per stepSimulation call and skips sleeping bodies; buildIslands turns a
    sleeping body of an island that has an active body into WANTS_DEACTIVATION with timer 0; updateActivationState leaves it there
    while it is slow (wantsSleeping() is true for that state)."""
    ref, (rest, drop) = _scene([dict(pos=(0, 0.5, 0)), dict(pos=(0.0, 45.0, 0.0))])
    _run(ref, 330)
    st, _ = ref.bulk_activation()
    assert st[rest - 1] == SLEEPING and st[drop - 1] == ACTIVE           # the first one sleeps while the second still falls
    seen_wants = False
    for _ in range(400):
        _run(ref, 1)
        st, tm = ref.bulk_activation()
        hdr, _ = ref.DynamicPairs()
        if len(hdr) and st[rest - 1] == WANTS:
            seen_wants = True
    assert seen_wants
    _run(ref, 600)
    st, _ = ref.bulk_activation()
    assert st[rest - 1] == SLEEPING and st[drop - 1] == SLEEPING
    assert abs(_y(ref, drop) - 1.5) < 0.02


def test_filters_apply_between_dynamic_boxes():
    ref, (a, b) = _scene([dict(pos=(0, 0.5, 0), layer=1, mask=0xFFFFFFFD), dict(pos=(0.0, 2.0, 0.0), layer=2, mask=0xFFFFFFFF)])
    _run(ref, 400)
    assert len(ref.DynamicPairs()[0]) == 0
    assert abs(_y(ref, b) - 0.5) < 0.01                                   # fell through the other box onto the plane


def test_a_box_on_a_box_on_a_static_box():
    """All three kinds of manifold in one island: plane off, a Static floor box, two Dynamic boxes stacked on it."""
    ref, (low, top) = _scene([dict(pos=(0, 0.5, 0)), dict(pos=(0.2, 1.8, 0.1), size=(0.4, 0.3, 0.4), euler=(0.5, 0, 0))], plane=False, ground_box=True)
    _run(ref, 800)
    assert abs(_y(ref, low) - 0.5) < 0.01 and abs(_y(ref, top) - 1.3) < 0.02
    st, _ = ref.bulk_activation()
    assert st[low - 1] == SLEEPING and st[top - 1] == SLEEPING
    assert len(ref.BoxContacts(low)) == 1 and len(ref.BoxContacts(low)[0][1]) == 4
