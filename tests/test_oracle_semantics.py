"""Semantic edge cases of the reference that the oracle (and therefore the GPU path) must honour.
Numbering follows SURVEY.md Appendix B; every case cites the reference lines it encodes."""
import numpy as np

from oracle import pyoracle as po

from helpers import DT

I = [0, 0, 0]


def _scene(n, **kw):
    sc = po.RefScene()
    ids = [sc.CreateEntity() for _ in range(n)]
    return sc, ids


def test_b1_dirty_semantics():
    # TransformSystem.cpp:18-30: own dirty => local+world; ancestor dirty only => world from CACHED local
    sc, (a, b) = _scene(2)
    sc.AddTransform(a, [1, 2, 3], I, [1, 1, 1])
    sc.AddTransform(b, [0, 1, 0], I, [1, 1, 1])
    sc.SetParent(b, a)
    sc.TransformSystemUpdate()
    assert sc.CountDirtyTransforms() == 0
    w0 = sc.GetTransform(b)["world"].copy()
    # mutate the child's TRS WITHOUT MarkDirty, dirty only the parent: child's world uses its stale local
    sc.SetTRS(b, pos=[0, 50, 0], mark_dirty=False)
    sc.SetTRS(a, pos=[2, 2, 3], mark_dirty=True)
    sc.TransformSystemUpdate()
    wb = sc.GetTransform(b)
    assert np.allclose(wb["world"][12:15], [2, 3, 3])          # parent moved by +1 in x, child offset still (0,1,0)
    assert np.allclose(wb["local"][12:15], [0, 1, 0])          # cached local untouched
    assert not np.array_equal(wb["world"], w0)
    # a clean subtree under clean ancestors is untouched
    sc.SetTRS(b, pos=[0, 60, 0], mark_dirty=False)
    before = sc.GetTransform(b)["world"].copy()
    sc.TransformSystemUpdate()
    assert np.array_equal(sc.GetTransform(b)["world"], before)


def test_b2_roots_and_transformless_middle_node():
    # Scene.cpp:523-533 + TransformSystem.cpp:12-16
    sc, (a, m, c) = _scene(3)
    sc.AddTransform(a, [1, 0, 0], I, [1, 1, 1])
    sc.AddTransform(c, [0, 0, 5], I, [1, 1, 1])
    sc.SetParent(m, a)   # m has no Transform
    sc.SetParent(c, m)
    sc.TransformSystemUpdate()
    assert sc.CountDirtyTransforms() == 0
    # c is a root (its parent has no Transform): world == local, parent chain ignored
    assert np.allclose(sc.GetTransform(c)["world"][12:15], [0, 0, 5])


def test_b3_cycles_cannot_be_built_through_the_reference_api():
    """SURVEY App. B.3 says nodes in a parent cycle are never reached by the DFS.  True — but closing a cycle through
    Scene::SetParent (Scene.cpp:354-393) first calls MarkHierarchyDirty (Scene.cpp:535-550), which recurses around the
    cycle until the stack overflows; the faithful restatement reproduces that crash, so it is not exercised here.
    The C ABI takes a raw parent array, where a cycle CAN be expressed: such entities are parked in never-ticked
    "limbo" tiles (tests/test_synth_flatten_capi.py::test_flatten_cycles_go_to_limbo, tests/test_gpu_parity.py)."""
    sc, (a, b, c) = _scene(3)
    for e in (a, b, c):
        sc.AddTransform(e, [1, 1, 1], I, [1, 1, 1])
    sc.SetParent(b, c)          # a chain is fine
    sc.TransformSystemUpdate()
    assert sc.CountDirtyTransforms() == 0


def test_b4_destroy_orphans_children_and_reuses_ids():
    # Scene.cpp:43-83, :21-41
    sc, (a, b) = _scene(2)
    sc.AddTransform(a, [10, 0, 0], I, [1, 1, 1])
    sc.AddTransform(b, [0, 1, 0], I, [1, 1, 1])
    sc.SetParent(b, a)
    sc.TransformSystemUpdate()
    sc.DestroyEntity(a)
    assert sc.GetParent(b) == 0 and sc.GetTransform(b)["dirty"]
    sc.TransformSystemUpdate()
    assert np.allclose(sc.GetTransform(b)["world"][12:15], [0, 1, 0])
    assert sc.CreateEntity() == a  # LIFO id reuse
    assert sc.GetTransform(a) is None


def test_b5_add_transform_twice_keeps_value_marks_dirty():
    # Scene.cpp:97-101: emplace keeps the old value
    sc, (a,) = _scene(1)
    sc.AddTransform(a, [3, 4, 5], I, [2, 2, 2])
    sc.TransformSystemUpdate()
    assert sc.AddTransform(a)
    t = sc.GetTransform(a)
    assert t["dirty"] and np.allclose(t["position"], [3, 4, 5])


def test_b6_physics_uses_local_trs_and_ignores_scale_and_parents():
    # PhysicsSystem.cpp:57-64, 941-948
    sc, (p, c) = _scene(2)
    sc.AddTransform(p, [100, 0, 0], I, [3, 3, 3])
    sc.AddTransform(c, [0, 10, 0], I, [5, 5, 5])
    sc.SetParent(c, p)
    sc.AddCollider(c)
    sc.AddRigidBody(c, po.BODY_DYNAMIC, 1.0)
    sc.PhysicsSystemUpdate(DT)
    body = sc.GetBody(c)
    dt = np.float32(DT)
    vy = np.float32(0) + np.float32(np.float32(np.float32(-9.81) * np.float32(1.0)) * np.float32(1.0)) * dt
    assert body["linvel"][1] == vy
    assert body["origin"][0] == 0.0 and body["origin"][1] == np.float32(10) + vy * dt  # local position, no parent offset
    assert np.array_equal(sc.GetTransform(c)["position"], body["origin"])
    assert sc.GetTransform(c)["dirty"]  # B.8: every Dynamic body is marked dirty every tick


def test_b7_teleport_rule():
    # PhysicsSystem.cpp:963-987
    sc, (d, k, s) = _scene(3)
    for e, t in ((d, po.BODY_DYNAMIC), (k, po.BODY_KINEMATIC), (s, po.BODY_STATIC)):
        sc.AddTransform(e, [0, 5, 0], I, [1, 1, 1])
        sc.AddCollider(e)
        sc.AddRigidBody(e, t, 1.0)
    for _ in range(3):
        sc.PhysicsSystemUpdate(DT)
        sc.TransformSystemUpdate()
    v_before = sc.GetBody(d)["linvel"].copy()
    assert v_before[1] < 0
    # dirty Dynamic: re-posed from the Transform, velocities zeroed, then stepped
    sc.SetTRS(d, pos=[7, 7, 7])
    sc.PhysicsSystemUpdate(DT)
    b = sc.GetBody(d)
    g_dt = np.float32(np.float32(np.float32(-9.81) * np.float32(1.0)) * np.float32(1.0)) * np.float32(DT)
    assert b["linvel"][1] == g_dt and b["origin"][0] == 7.0
    # Kinematic / Static follow the Transform when dirty and are never written back
    sc.SetTRS(k, pos=[1, 2, 3])
    sc.SetTRS(s, pos=[4, 5, 6])
    sc.PhysicsSystemUpdate(DT)
    assert np.array_equal(sc.GetBody(k)["origin"], np.float32([1, 2, 3]))
    assert np.array_equal(sc.GetBody(s)["origin"], np.float32([4, 5, 6]))
    assert np.array_equal(sc.GetTransform(k)["position"], np.float32([1, 2, 3]))
    assert np.array_equal(sc.GetBody(k)["linvel"], np.zeros(3, np.float32))


def test_b9_body_recreation_resets_velocity_and_clamps_mass():
    # PhysicsSystem.cpp:398-477: mass = max(mass, 0.01) for Dynamic; body.dirty => new body, zero velocity
    sc, (d,) = _scene(1)
    sc.AddTransform(d, [0, 5, 0], I, [1, 1, 1])
    sc.AddCollider(d)
    sc.AddRigidBody(d, po.BODY_DYNAMIC, 0.0)   # clamped to 0.01
    sc.PhysicsSystemUpdate(DT)
    sc.TransformSystemUpdate()
    inv_m = np.float32(1.0) / np.float32(0.01)
    f = np.float32(-9.81) / inv_m   # btRigidBody::setGravity: acceleration / m_inverseMass (a division in the reference's build)
    assert sc.GetBody(d)["linvel"][1] == (f * inv_m) * np.float32(DT)
    sc.PhysicsSystemUpdate(DT)
    sc.TransformSystemUpdate()
    v2 = sc.GetBody(d)["linvel"][1]
    sc.MarkBodyDirty(d)
    sc.PhysicsSystemUpdate(DT)
    assert abs(sc.GetBody(d)["linvel"][1]) < abs(v2)  # restarted from zero


def test_parent_times_local_order():
    # Transform.cpp:30: world = parentWorld * local (NOT local * parent): the parent's translation is transformed
    # by the child's scale/rotation
    sc, (a, b) = _scene(2)
    sc.AddTransform(a, [0, 7, -5], I, [0.05, 0.05, 0.05])
    sc.AddTransform(b, [1, 2, 3], [0.3, -1.2, 2.5], [2, 3, 4])
    sc.SetParent(b, a)
    sc.TransformSystemUpdate()
    w = sc.GetTransform(b)["world"].view(np.uint32)
    assert list(w[12:16]) == [0xC0F3D898, 0xC19FDD2F, 0xC15D5EEE, 0x3F800000]


def test_rigidbody_without_collider_or_transform_has_no_body():
    # PhysicsSystem.cpp:1254-1258, :389-393
    sc, (a, b) = _scene(2)
    sc.AddTransform(a, [0, 1, 0], I, [1, 1, 1])
    sc.AddRigidBody(a, po.BODY_DYNAMIC, 1.0)            # no collider
    sc.AddCollider(b)
    sc.AddRigidBody(b, po.BODY_DYNAMIC, 1.0)            # no transform
    sc.PhysicsSystemUpdate(DT)
    assert sc.GetBody(a) is None and sc.GetBody(b) is None
    assert np.array_equal(sc.GetTransform(a)["position"], np.float32([0, 1, 0]))
