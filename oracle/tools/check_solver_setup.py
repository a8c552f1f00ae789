#!/usr/bin/env python3
"""The rest of the contact path of oracle/contact_ref.h and oracle/boxbox_ref.h against the reference's COMPILED Bullet.

TEST INFRASTRUCTURE (oracle/): reads /root/reference/build/bin/RelWithDebInfo/SandboxCity.exe as bytes through objdump and
interprets the listing symbolically (oracle/tools/symx.py); nothing in it is loaded or run.  VERDICT r02 item 5.

Every function below is located in the symbol-less exe by what it reaches (constants, callers, vtable slots), executed
symbolically on the control paths the restatement takes (body A a Dynamic rigid body; body B a static object — no rigid body
behind its solver body — or, for oracle/island_ref.h, a second Dynamic rigid body; warm starting and split impulse on; no contact flags), and the expression tree of every value it stores
is compared with the tree of the restatement, written below in the structure of the C++ (DotXZY, InvMassPlusDot, XformPoint ...
are the helpers of contact_ref.h).  Trees are normalised for the commutativity of + and x, for the sign rules of neg with x and /
and for a + (-b) = a - b only — not for associativity: Bullet is built with MSVC /fp:fast in this exe, the compiler reassociated
several sums, and the restatement follows the compiled association wherever the two differ.

  1. setupContactConstraint   (the function that stores the immediate 1e10 into m_upperLimit)
  2. setupFrictionConstraint  (the callee of addFrictionConstraint, itself called from convertContact)
  3. convertContact: friction rows are NOT warm-started (setFrictionConstraintImpulse zeroes m_appliedImpulse and the function
     never reads m_appliedImpulseLateral1), rel_pos / getVelocityInLocalPointNoDelta / rel_vel / the lateral direction as text
  4. computeGyroscopicImpulseImplicit_Body with btMatrix3x3::solve33 and btQuaternion::operator*= inside
  5. btPersistentManifold::refreshContactPoints, sortCachedPoints (gContactCalcArea3Points = true in .data), getCacheEntry,
     btManifoldResult::addContactPoint up to getCacheEntry (pointA and the two invXform local points)
  6. btRigidBody::updateInertiaTensor, the capsule's localGetSupportingVertex (+ WithoutMargin), the convex-plane single contact
     (what reaches btManifoldResult::addContactPoint: normal, point, depth)

Prints one line per fact and "RESULT: ..." at the end; exit code 1 on any mismatch, 2 when the reference is absent.
"""
import os
import re
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from symx import Machine, function_listing, show, norm2  # noqa: E402

P = lambda n, o=0: ("ptr", n, o)  # noqa: E731
COOKIE = 0x1402DFCF0


# ---------------------------------------------------------------- restatement helpers (oracle/contact_ref.h, operator for operator)
def _E():
    from check_bx_order import E
    if not hasattr(E, "__truediv__"):
        E.__truediv__ = lambda a, b: E(("div", a.t, E.w(b).t))
    return E


class V3:
    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    def __iter__(self):
        return iter((self.x, self.y, self.z))


def I(sp, off):
    return _E()(("in", sp, off))


def C(x):
    return _E()(("const", struct.unpack("<f", struct.pack("<f", x))[0]))


def vec(sp, off):
    return V3(I(sp, off), I(sp, off + 4), I(sp, off + 8))


def add(a, b):
    return V3(a.x + b.x, a.y + b.y, a.z + b.z)


def sub(a, b):
    return V3(a.x - b.x, a.y - b.y, a.z - b.z)


def scale(a, s):
    return V3(a.x * s, a.y * s, a.z * s)


def dot(a, b):
    return a.x * b.x + a.y * b.y + a.z * b.z


def dot_xzy(a, b):
    return (a.x * b.x + a.z * b.z) + a.y * b.y


def inv_mass_plus_dot(inv_mass, n, v):
    return (inv_mass + n.z * v.z) + (n.x * v.x + n.y * v.y)


def cross(a, b):
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x)


def mat_vec(rows, v):
    return V3(*[r.x * v.x + r.y * v.y + r.z * v.z for r in rows])


def xform_point(rows, o, l):
    return V3(*[(oo + l.y * r.y) + (l.x * r.x + l.z * r.z) for r, oo in zip(rows, o)])


def xform_point_b(rows, o, l):
    r0, r1, r2 = rows
    return V3((o.x + l.z * r0.z) + (l.x * r0.x + l.y * r0.y), (o.y + l.y * r1.y) + (l.x * r1.x + l.z * r1.z), (o.z + l.y * r2.y) + (l.x * r2.x + l.z * r2.z))


def same(got, want):
    return norm2(got) == norm2(want.t if hasattr(want, "t") else want)


def report(tag, pairs, names=None, leaf=None):
    """pairs: [(label, compiled tree, restated E)]; prints the line, the first difference in full; returns ok."""
    bad = [(l, g, w) for l, g, w in pairs if not same(g, w)]
    print(f"{tag}: {len(pairs) - len(bad)} of {len(pairs)} stored values identical as expression trees" + ("" if not bad else "  <-- MISMATCH"))
    for l, g, w in bad[:2]:
        print(f"    {l}\n      compiled   : {show(norm2(g), names, leaf)[:700]}\n      restatement: {show(norm2(w.t), names, leaf)[:700]}")
    return not bad


# ---------------------------------------------------------------- 1. setupContactConstraint
def find_setup_contact(text_ins):
    from check_bullet_order import _function_start
    hits = [pc for pc, mn, ops in text_ins if mn == "movl" and ops.startswith("$0x501502f9,0x94(")]
    return hits


def contact_row_restated(bounce, pen_pos, above_split, two=False):
    E = _E()
    n, r = vec("cp", 0x40), vec("rp1", 0)
    r2 = vec("rp2", 0)
    inv_i = [vec("rb0", 0x180), vec("rb0", 0x190), vec("rb0", 0x1A0)]
    ang_f = vec("rb0", 0x2C0)
    inv_mass = I("rb0", 0x1D0)
    dt, sor, erp2, cfm, slop, warm = I("info", 0xC), I("info", 0x1C), I("info", 0x24), I("info", 0x34), I("info", 0x4C), I("info", 0x50)
    lin = add(vec("bodyA", 0xB0), vec("bodyA", 0xD0))
    ang = add(vec("bodyA", 0xC0), vec("bodyA", 0xE0))
    inv_dt = 1.0 / dt
    torque_axis = cross(r, n)
    ang_comp = V3(*[a * f for a, f in zip(mat_vec(inv_i, torque_axis), ang_f)])
    v = cross(ang_comp, r)
    denom0 = inv_mass_plus_dot(inv_mass, n, v)
    zero = V3(C(0.0), C(0.0), C(0.0))
    if two:
        inv_i2 = [vec("rb1", 0x180), vec("rb1", 0x190), vec("rb1", 0x1A0)]
        torque_axis2 = cross(n, r2)                                        # rel_pos2 x -n
        ang_comp2 = V3(*[a * f for a, f in zip(mat_vec(inv_i2, torque_axis2), vec("rb1", 0x2C0))])
        denom1 = inv_mass_plus_dot(I("rb1", 0x1D0), n, cross(r2, ang_comp2))
        rb_vel2 = add(vec("rb1", 0x1B0), cross(vec("rb1", 0x1C0), r2))
    else:
        denom1, rb_vel2 = C(0.0), zero
    jac = sor / ((denom0 + denom1) + inv_dt * cfm)
    penetration = I("cp", 0x50) + slop
    rb_vel = add(vec("rb0", 0x1B0), cross(vec("rb0", 0x1C0), r))
    rb_rel = n.x * (rb_vel.x - rb_vel2.x) + n.y * (rb_vel.y - rb_vel2.y) + n.z * (rb_vel.z - rb_vel2.z)
    restitution = E(("max", (-(rb_rel * I("cp", 0x60))).t, C(0.0).t)) if bounce else E(("max", C(0.0).t, C(0.0).t))
    applied = I("cp", 0x84) * warm
    if two:
        l2 = add(vec("bodyB", 0xB0), vec("bodyB", 0xD0))
        vel2 = dot_xzy(torque_axis2, add(vec("bodyB", 0xC0), vec("bodyB", 0xE0))) + ((-(l2.x * n.x) - l2.z * n.z) - l2.y * n.y)
    else:
        b_lin = V3(*[C(0.0) + I("bodyB", 0xB0 + 4 * k) for k in range(3)])
        b_ang = V3(*[C(0.0) + I("bodyB", 0xC0 + 4 * k) for k in range(3)])
        vel2 = dot_xzy(b_lin, zero) + dot_xzy(b_ang, zero)          # (body B's side: products with the zeroed normal, exact zeros)
    vel1 = dot_xzy(n, lin) + dot_xzy(torque_axis, ang)
    rel_vel = vel2 + vel1
    velocity_error = restitution - rel_vel
    if pen_pos:
        positional = C(0.0)
        velocity_error = velocity_error - penetration * inv_dt
    else:
        positional = -((penetration * erp2) * inv_dt)
    pen_imp, vel_imp = positional * jac, velocity_error * jac
    out = {}
    for k, c in enumerate(torque_axis):
        out[4 * k] = c
    for k, c in enumerate(n):
        out[0x10 + 4 * k] = c
    for k, c in enumerate(ang_comp):
        out[0x40 + 4 * k] = c
    out[0x84] = jac
    out[0x70] = applied
    out[0x88] = (pen_imp + vel_imp) if above_split else vel_imp
    out[0x98] = C(0.0) if above_split else pen_imp
    out[0x8C] = jac * (inv_dt * cfm)
    out[0x90] = C(0.0)
    out[0x94] = C(1e10)
    lin_f, ang_f2, inv_m = vec("bodyA", 0x70), vec("bodyA", 0x60), vec("bodyA", 0x80)
    body = {}
    for k in range(3):
        nk, ik, lk = list(n)[k], list(inv_m)[k], list(lin_f)[k]
        body[0x40 + 4 * k] = I("bodyA", 0x40 + 4 * k) + lk * ((ik * nk) * applied)
        body[0x50 + 4 * k] = I("bodyA", 0x50 + 4 * k) + list(ang_comp)[k] * (list(ang_f2)[k] * applied)
    body_b = {}
    if two:
        lin_fb, ang_fb, inv_mb = vec("bodyB", 0x70), vec("bodyB", 0x60), vec("bodyB", 0x80)
        for k in range(3):
            out[0x20 + 4 * k] = list(torque_axis2)[k]
            out[0x30 + 4 * k] = -list(n)[k]
            out[0x50 + 4 * k] = list(ang_comp2)[k]
            body_b[0x40 + 4 * k] = I("bodyB", 0x40 + 4 * k) - list(lin_fb)[k] * ((list(inv_mb)[k] * list(n)[k]) * applied)
            body_b[0x50 + 4 * k] = I("bodyB", 0x50 + 4 * k) + list(ang_comp2)[k] * (list(ang_fb)[k] * applied)
    return out, body, body_b


def check_setup_contact(pe, all_ins):
    from check_bullet_order import _function_start
    stores = find_setup_contact(all_ins)
    ok = len(stores) == 1
    fn = pe.base + _function_start(pe, stores[0] - pe.base) if stores else 0
    print(f"setupContactConstraint: the one function that stores 1e10 (m_upperLimit) as an immediate, at VA {fn:#x}" + ("" if ok else "  <-- NOT unique"))
    if not ok:
        return False
    ins = function_listing(pe, fn, 0x1000)
    # the two solver-body pointers are pool + id * 256: the instruction after each `add 0x18(%rcx),reg`
    adds = [i for i, (pc, mn, ops) in enumerate(ins) if mn == "add" and ops.startswith("0x18(%rcx),")]
    hooks = {}
    for i in adds:
        reg = ins[i][2].split(",")[1]
        hooks[ins[i + 1][0]] = (lambda name: (lambda m, r=reg: m.gpr.__setitem__(r, P(name))))("bodyB" if reg == "%r8" else "bodyA")
    good = True
    for variant in ((True, False, True, False), (False, True, True, False), (False, False, False, False), (True, True, True, False),
                    (True, False, False, True), (False, True, True, True), (False, False, True, True)):
        bounce, pen_pos, above, two = variant

        def decider(m, pc, mn, lf, v=variant):
            _, fm, fo = lf
            table = {("test", "%r9,%r9"): False, ("test", "%r11,%r11"): not v[3], ("test", "$0x6,%al"): False, ("test", "$0x8,%al"): True,
                     ("testb", "$0x4,0x58(%rbx)"): False, ("cmp", "%rsi,0xf0(%r10)"): False, ("cmp", "%rsi,0xf0(%r8)"): not v[3],
                     ("cmp", "%esi,0x40(%rbx)"): False, ("comiss", "0x70(%rbx),%xmm0"): v[0], ("comiss", "%xmm13,%xmm11"): not v[1],
                     ("comiss", "0x44(%rbx),%xmm11"): v[2]}
            return table.get((fm, fo))
        m = Machine(pe, ins, gpr={"%rcx": P("solver"), "%rdx": P("sc"), "%r8": ("int", 1), "%r9": ("int", 0)},
                    stack_ptrs={0x28: P("cp"), 0x30: P("info"), 0x38: P("relax"), 0x40: P("rp1"), 0x48: P("rp2")},
                    ptr_loads={("bodyA", 0xF0): P("rb0"), ("bodyB", 0xF0): P("rb1") if two else ("int", 0)}, pc_hooks=hooks)
        if two:
            m.gpr["%r9"] = ("int", 2)
        m.decider = decider
        m.run()
        want, body, body_b = contact_row_restated(*variant)
        pairs = [(f"solverConstraint+{k:#x}", m.mem[("sc", k)], w) for k, w in want.items()]
        pairs += [(f"solverBodyA+{k:#x}", m.mem[("bodyA", k)], w) for k, w in body.items()]
        pairs += [(f"solverBodyB+{k:#x}", m.mem[("bodyB", k)], w) for k, w in body_b.items()]
        good &= report(f"  path ({'two rigid bodies' if two else 'body B static'}; |rel_vel| >= threshold: {bounce}, penetration > 0: {pen_pos}, above the split threshold: {above})", pairs)
    return good


# ---------------------------------------------------------------- 2. setupFrictionConstraint (+ 3. convertContact)
def find_convert_contact(pe, setup_va):
    """The one caller of setupContactConstraint."""
    from check_bullet_order import _rip_refs, _function_start
    va, _, raw, rs = pe.secs[0]
    code = pe.b[raw: raw + rs]
    targets = {setup_va}
    # incremental-link thunks that jump to it
    for m in re.finditer(b"\xe9", code):
        o = m.start()
        if o + 5 <= len(code) and pe.base + va + o + 5 + struct.unpack_from("<i", code, o + 1)[0] == setup_va and pe.base + va + o < pe.base + 0x10000 + va:
            targets.add(pe.base + va + o)
    callers = set()
    for m in re.finditer(b"\xe8", code):
        o = m.start()
        if o + 5 <= len(code) and pe.base + va + o + 5 + struct.unpack_from("<i", code, o + 1)[0] in targets:
            callers.add(pe.base + _function_start(pe, va + o))
    return sorted(callers)


def direct_calls(pe, ins):
    from check_bullet_order import _resolve
    return [(pc, _resolve(pe, int(ops, 16))) for pc, mn, ops in ins if mn == "call" and re.fullmatch(r"0x[0-9a-f]+", ops)]


def friction_row_restated(two=False):
    t, r = vec("axis", 0), vec("rp1", 0)
    r2 = vec("rp2", 0)
    inv_i = [vec("rb0", 0x180), vec("rb0", 0x190), vec("rb0", 0x1A0)]
    ang_f = vec("rb0", 0x2C0)
    inv_mass = I("rb0", 0x1D0)
    relax, desired, cfm_slip = I("stk", 0x58), I("stk", 0x68), I("stk", 0x70)
    lin = add(vec("bodyA", 0xB0), vec("bodyA", 0xD0))
    ang = vec("bodyA", 0xC0)
    c = cross(r, t)
    ang_comp = V3(*[a * f for a, f in zip(mat_vec(inv_i, c), ang_f)])
    v = cross(ang_comp, r)
    vel1 = dot_xzy(t, lin) + dot_xzy(c, ang)
    if two:
        c2 = cross(t, r2)
        ang_comp2 = V3(*[a * f for a, f in zip(mat_vec([vec("rb1", 0x180), vec("rb1", 0x190), vec("rb1", 0x1A0)], c2), vec("rb1", 0x2C0))])
        jac = relax / (inv_mass_plus_dot(inv_mass, t, v) + inv_mass_plus_dot(I("rb1", 0x1D0), t, cross(r2, ang_comp2)))
        l2 = add(vec("bodyB", 0xB0), vec("bodyB", 0xD0))
        rel_vel = dot_xzy(c2, vec("bodyB", 0xC0)) + ((vel1 - l2.z * t.z) + (-(l2.x * t.x) - l2.y * t.y))
    else:
        jac = relax / (inv_mass_plus_dot(inv_mass, t, v) + C(0.0))
        z = C(0.0)
        rel_vel = (vel1 + z * z) + (z * z + z * z) + ((z * z + z * z) + z * z)     # (body B's side: exact zeros, as compiled)
    out = {0x84: jac, 0x88: C(0.0) + jac * (desired - rel_vel), 0x8C: cfm_slip, 0x90: -I("cp", 0x54), 0x94: I("cp", 0x54), 0x80: I("cp", 0x54)}
    for k in range(3):
        out[4 * k] = list(c)[k]
        out[0x10 + 4 * k] = list(t)[k]
        out[0x40 + 4 * k] = list(ang_comp)[k]
        out[0x70 + 4 * k] = C(0.0)        # m_appliedImpulse
        if two:
            out[0x20 + 4 * k] = list(c2)[k]
            out[0x30 + 4 * k] = -list(t)[k]
            out[0x50 + 4 * k] = list(ang_comp2)[k]
    return out


def check_friction_and_convert(pe, setup_va):
    callers = find_convert_contact(pe, setup_va)
    ok = len(callers) == 1
    print(f"convertContact: the one caller of setupContactConstraint, at VA {callers[0]:#x}" if ok else f"convertContact: callers {callers}  <-- NOT unique")
    if not ok:
        return False
    cc = function_listing(pe, callers[0], 0x2000)
    calls = direct_calls(pe, cc)
    after = [t for pc, t in calls if pc > next(pc for pc, t in calls if t == setup_va)]
    # addFrictionConstraint = the callee (after setupContactConstraint) that is called most often and itself calls one big function
    from collections import Counter
    add_friction = None
    for t, _ in Counter(after).most_common():
        body = function_listing(pe, t, 0x400)
        inner = [x for _, x in direct_calls(pe, body)]
        big = [x for x in inner if len(function_listing(pe, x, 0x1000)) > 250]
        if len(big) == 1 and any(mn == "mulss" for _, mn, _ in function_listing(pe, big[0], 0x1000)):
            add_friction, setup_friction = t, big[0]
            break
    if add_friction is None:
        print("setupFrictionConstraint: not found  <-- MISMATCH")
        return False
    ins = function_listing(pe, setup_friction, 0x1000)
    first_indexed = next(pc for pc, mn, ops in ins if re.search(r"0xf0\(%r\w+,%r\w+,1\)", ops))

    def sethook(m):
        m.gpr["%rcx"], m.gpr["%r11"], m.gpr["%r10"] = P("bodyB"), P("bodyA"), ("int", 0)

    good = True
    for two in (False, True):
        def decider(m, pc, mn, lf, two=two):
            table = {("test", "%rbx,%rbx"): False, ("test", "%rdi,%rdi"): not two, ("testb", "$0x10,0x80(%r9)"): True}
            return table.get((lf[1], lf[2]))
        m = Machine(pe, ins, gpr={"%rcx": P("solver"), "%rdx": P("sc"), "%r8": P("axis"), "%r9": ("int", 0)},
                    stack_ptrs={0x30: P("cp"), 0x38: P("rp1"), 0x40: P("rp2"), 0x48: P("colObj0"), 0x50: P("colObj1"), 0x60: P("info")},
                    ptr_loads={("bodyA", 0xF0): P("rb0"), ("bodyB", 0xF0): P("rb1") if two else ("int", 0)}, pc_hooks={first_indexed: sethook})
        m.decider = decider
        m.run()
        want = friction_row_restated(two)
        good &= report(f"setupFrictionConstraint at VA {setup_friction:#x} (through addFrictionConstraint at {add_friction:#x}; {'two rigid bodies' if two else 'body B static'})",
                       [(f"solverConstraint+{k:#x}", m.mem[("sc", k)], w) for k, w in want.items()])
    # ---- convertContact: no read of cp.m_appliedImpulseLateral1, friction row's m_appliedImpulse zeroed after the last call
    # %rdi walks the manifold points at +0xb4 from each point's start in this function
    lea = next((ops for pc, mn, ops in cc if mn == "lea" and re.fullmatch(r"0xc4\(%r14\),%rdi", ops)), None)
    reads_lateral = [hex(pc) for pc, mn, ops in cc if re.search(r"(^|,)-0x2c\(%rdi\)", ops) and not ops.endswith("-0x2c(%rdi)")]
    zeroed = False
    for k, (pc, mn, ops) in enumerate(cc):
        if mn == "movups" and re.fullmatch(r"%xmm0,0x70\(%r\w+,%r\w+,1\)", ops):
            writer = next(((m2, o2) for _, m2, o2 in reversed(cc[:k]) if o2.endswith(",%xmm0")), None)
            zeroed = writer == ("xorps", "%xmm0,%xmm0")
    fact = lea is not None and not reads_lateral and zeroed
    print("convertContact: m_appliedImpulseLateral1 (+0x88 of a manifold point) is never read, and after the last addFrictionConstraint the friction "
          "row's m_appliedImpulse (+0x70) is stored from a zeroed register: friction rows are not warm-started: " + ("yes" if fact else "NO  <-- MISMATCH"))
    # ---- rel_pos, velocities, rel_vel and the lateral direction: symbolic run of the loop body from the first read of the point
    start = next(pc for pc, mn, ops in cc if mn == "cmpq" and ops == "$0x0,0xf0(%r9)")
    rec = {}

    class Stop(Exception):
        pass

    def at_setup(m):
        rec["rel_vel"] = m.get("%xmm15")[0]
        rec["vel"] = [m.get("%xmm7")[0], m.get("%xmm8")[0], m.get("%xmm6")[0]]
        sp = m.gpr["%rsp"][2]
        rp1, rp2 = m.memp[("stk", sp + 0x38)], m.memp[("stk", sp + 0x40)]
        rec["rp1"] = [m.cell((rp1[1], rp1[2] + 4 * k)) for k in range(3)]
        rec["rp2"] = [m.cell((rp2[1], rp2[2] + 4 * k)) for k in range(3)]
        m.gpr["%rcx"] = P("scpool")

    def at_aniso(m):
        pass

    def at_add_friction(m):
        a = m.gpr["%rdx"]
        rec["axis"] = [m.cell((a[1], a[2] + 4 * k)) for k in range(3)]
        raise Stop()

    from check_bullet_order import _resolve
    aniso = Counter(t for pc, t in calls if pc > start and t not in (setup_va, add_friction)).most_common(1)[0][0]
    for two in (False, True):
        rec.clear()

        def decider2(m, pc, mn, lf, two=two):
            table = {("cmpq", "$0x0,0xf0(%r9)"): False, ("cmpq", "$0x0,0xf0(%rdx)"): not two, ("comiss", "-0x5c(%rdi),%xmm9"): True,
                     ("testb", "$0x20,0x58(%rbx)"): True, ("testb", "$0x40,0x58(%rax)"): False}
            if lf[1] == "comiss" and "(%rip)" in lf[2] and mn == "jbe":
                return False                           # lat_rel_vel > SIMD_EPSILON
            return table.get((lf[1], lf[2]))
        m = Machine(pe, cc, gpr={"%r9": P("bodyA"), "%rdx": P("bodyB"), "%rdi": P("cp", 0xB4), "%r12": P("colObj0"), "%r13": P("colObj1"), "%rsi": P("solver"),
                                 "%rbx": P("info"), "%rax": P("sc"), "%rbp": P("stk", -0x100), "%r15": P("cp"), "%r14": ("int", 0), "%r10": ("int", 1), "%rcx": ("int", 0)},
                    hooks={setup_va: at_setup, aniso: at_aniso, add_friction: at_add_friction})
        m.gpr["%rsp"] = P("stk", -0x300)
        m.memp[("stk", -0x100 - 0x60)] = P("sc")
        m.memp[("stk", -0x100 + 0x110)] = P("info")
        m.xmm["%xmm9"] = [("const", 0.0)] * 4
        m.xmm["%xmm14"] = [("signmask",)] * 4
        m.decider = decider2
        try:
            m.run(start=start)
        except Stop:
            pass
        lin = add(vec("bodyA", 0xB0), vec("bodyA", 0xD0))
        ang = add(vec("bodyA", 0xC0), vec("bodyA", 0xE0))
        rp1 = sub(vec("cp", 0x30), vec("colObj0", 0x40))
        rp2 = sub(vec("cp", 0x20), vec("colObj1", 0x40))
        vel1 = add(lin, cross(ang, rp1))
        if two:
            vel2 = add(add(vec("bodyB", 0xB0), vec("bodyB", 0xD0)), cross(add(vec("bodyB", 0xC0), vec("bodyB", 0xE0)), rp2))
        else:
            vel2 = V3(C(0.0), C(0.0), C(0.0))
        vel = sub(vel1, vel2)
        n = vec("cp", 0x40)
        rel_vel = dot(vel, n)
        d = sub(vel, scale(n, rel_vel))
        inv_len = 1.0 / _E()(("sqrtf", dot(d, d).t))
        axis = scale(d, inv_len)
        pairs = [(f"rel_pos1.{c}", g, w) for c, g, w in zip("xyz", rec.get("rp1", [("missing",)] * 3), rp1)]
        pairs += [(f"rel_pos2.{c}", g, w) for c, g, w in zip("xyz", rec.get("rp2", [("missing",)] * 3), rp2)]
        pairs += [(f"vel.{c}", g, w) for c, g, w in zip("xyz", rec.get("vel", [("missing",)] * 3), vel)]
        pairs += [("rel_vel", rec.get("rel_vel", ("missing",)), rel_vel)]
        pairs += [(f"lateral direction.{c}", g, w) for c, g, w in zip("xyz", rec.get("axis", [("missing",)] * 3), axis)]
        good &= report(f"convertContact's loop body ({'two rigid bodies' if two else 'body B static'}: rel_pos, getVelocityInLocalPointNoDelta, rel_vel, the normalised lateral direction)", pairs)
    return good and fact


# ---------------------------------------------------------------- 4. computeGyroscopicImpulseImplicit_Body
def check_gyroscopic(pe, convert_bodies_callees):
    """Among the three gyroscopic functions convertBodies calls, the implicit body-frame one is the only one that calls four
    functions besides the stack-cookie check (getRotation, operator*= twice, solve33)."""
    from check_bullet_order import _resolve
    cand = []
    for f in convert_bodies_callees:
        ins = function_listing(pe, f, 0x800)
        calls = [t for _, t in direct_calls(pe, ins)]
        if len([t for t in calls if t != COOKIE and len(function_listing(pe, t, 0x40)) > 3]) >= 4 and any(mn == "divss" for _, mn, _ in ins):
            cand.append((f, ins, calls))
    if len(cand) != 1:
        print(f"computeGyroscopicImpulseImplicit_Body: {len(cand)} candidates  <-- MISMATCH")
        return False
    fn, ins, calls = cand[0]
    get_rotation = calls[0]

    def hook_rotation(m):
        out = m.gpr["%rdx"]
        for k in range(4):
            m.mem[(out[1], out[2] + 4 * k)] = ("in", "q", 4 * k)

    def decider(m, pc, mn, lf):
        if lf[1] == "ucomiss":
            return False          # je: an inverse inertia component is 0 -> it is not
        if lf[1] == "comiss":
            return False          # jbe: |det| <= eps -> it is larger
        return None
    m = Machine(pe, ins, gpr={"%rcx": P("rb"), "%rdx": P("out")}, hooks={get_rotation: hook_rotation, COOKIE: lambda m: None, 0x1402E3310: lambda m: None})
    m.xmm["%xmm2"] = [("in", "step", 0)] + [("const", 0.0)] * 3
    m.decider = decider
    m.run()
    # ---- restatement (contact_ref.h GyroscopicImpulse)
    q = tuple(I("q", 4 * k) for k in range(4))
    W = vec("rb", 0x1C0)
    step = I("step", 0)
    idl = V3(1.0 / I("rb", 0x210), 1.0 / I("rb", 0x214), 1.0 / I("rb", 0x218))

    def quat_times_vec(q, w):
        return (q[3] * w.x + q[1] * w.z - q[2] * w.y, q[3] * w.y + q[2] * w.x - q[0] * w.z, q[3] * w.z + q[0] * w.y - q[1] * w.x, -q[0] * w.x - q[1] * w.y - q[2] * w.z)

    def quat_mul(a, b):
        return (a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1], a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2],
                a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0], a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2])

    def quat_rotate(q, v):
        r = quat_mul(quat_times_vec(q, v), (-q[0], -q[1], -q[2], q[3]))
        return V3(r[0], r[1], r[2])
    ob = quat_rotate((-q[0], -q[1], -q[2], q[3]), W)
    ibo = V3(idl.x * ob.x, idl.y * ob.y, idl.z * ob.z)
    f = scale(cross(ob, ibo), step)
    J = [[idl.x, (idl.z * ob.z - idl.y * ob.z) * step, (idl.z * ob.y - idl.y * ob.y) * step],
         [(idl.x * ob.z - idl.z * ob.z) * step, idl.y, (idl.x * ob.x - idl.z * ob.x) * step],
         [(idl.y * ob.y - idl.x * ob.y) * step, (idl.y * ob.x - idl.x * ob.x) * step, idl.z]]
    col = [V3(J[0][c], J[1][c], J[2][c]) for c in range(3)]
    inv = 1.0 / dot(col[0], cross(col[1], col[2]))
    sol = V3(inv * dot(f, cross(col[1], col[2])), inv * dot(col[0], cross(f, col[2])), inv * dot(col[0], cross(col[1], f)))
    o2 = quat_rotate(q, sub(ob, sol))
    gf = sub(o2, W)
    return report(f"computeGyroscopicImpulseImplicit_Body at VA {fn:#x} (idl = 1 / m_invInertiaLocal, the quaternion products, J, solve33, the Newton step)",
                  [(f"impulse.{c}", m.mem[("out", 4 * k)], w) for k, (c, w) in enumerate(zip("xyz", gf))])


# ---------------------------------------------------------------- 5. the manifold functions
def check_manifold(pe, refresh_va, sort_va, cache_va):
    ok = True
    # ---- refreshContactPoints: one point through both loops
    ins = function_listing(pe, refresh_va, 0x800)
    loop1 = next(pc for pc, mn, ops in ins if mn == "movss" and ops.startswith("-0x14(%rcx)"))
    loop2 = next(pc for pc, mn, ops in ins if mn == "movss" and ops == "0x8(%rdi),%xmm4")
    jumps = [pc for pc, mn, ops in ins if mn in ("js", "jns")]

    def hook1(m):
        if not getattr(m, "seen1", False):
            m.gpr["%rcx"], m.seen1 = P("cp", 0x18), True

    def decider(m, pc, mn, lf):
        if mn in ("js", "jns"):
            return False
        if mn == "jbe":
            return True                      # the point stays: distance <= threshold, 2d distance <= threshold squared
        if lf[1] == "test" and mn == "je":
            return True                      # no contact-processed callback
        return None
    m = Machine(pe, ins, gpr={"%rcx": P("man"), "%rdx": P("trA"), "%r8": P("trB")},
                pc_hooks={loop1: hook1, loop2: lambda m: m.gpr.__setitem__("%rdi", P("cp", 0x48))})
    m.decider = decider
    m.run()
    A = [vec("trA", 0), vec("trA", 0x10), vec("trA", 0x20)]
    B = [vec("trB", 0), vec("trB", 0x10), vec("trB", 0x20)]
    wA = xform_point(A, vec("trA", 0x30), vec("cp", 0))
    wB = xform_point_b(B, vec("trB", 0x30), vec("cp", 0x10))
    n = vec("cp", 0x40)
    dist = dot(sub(wA, wB), n)
    pairs = [(f"m_positionWorldOnA.{c}", m.mem[("cp", 0x30 + 4 * k)], w) for k, (c, w) in enumerate(zip("xyz", wA))]
    pairs += [(f"m_positionWorldOnB.{c}", m.mem[("cp", 0x20 + 4 * k)], w) for k, (c, w) in enumerate(zip("xyz", wB))]
    pairs += [("m_distance1", m.mem[("cp", 0x50)], dist)]
    thr = I("man", 0x364)
    proj = sub(wA, scale(n, dist))
    pd = sub(wB, proj)
    cmp_ok = len(m.compares) == 2 and same(m.compares[0][1], dist) and same(m.compares[0][2], thr) and same(m.compares[1][1], dot(pd, pd)) and same(m.compares[1][2], thr * thr)
    ok &= report(f"btPersistentManifold::refreshContactPoints at VA {refresh_va:#x} (world points as (o + l.y b) + (l.x a + l.z c); body B's x row pairs o with the z product)", pairs)
    print("  its two tests: distance against the breaking threshold, the squared 2d distance of the projected point against its square: "
          + ("as restated" if cmp_ok else "MISMATCH"))
    ok &= cmp_ok
    # ---- sortCachedPoints
    ins = function_listing(pe, sort_va, 0x800)
    flag = next((pc, ops) for pc, mn, ops in ins if mn == "cmp" and "(%rip)" in ops)
    nxt = ins[[pc for pc, _, _ in ins].index(flag[0]) + 1][0]
    va = nxt + int(flag[1].split(",")[1].split("(")[0], 16) if flag[1].startswith("%") else nxt + int(flag[1].split("(")[0], 16)
    area3 = pe.b[pe.r2f(va - pe.base)]

    def decider_s(m, pc, mn, lf):
        if mn == "jae":
            return True                       # no cached point is deeper than the new one
        if lf[1] == "cmp" and "(%rip)" in lf[2]:
            return False                      # gContactCalcArea3Points is true
        if lf[1] == "test" and lf[2] == "%ebx,%ebx":
            return False
        if lf[1] == "cmp" and lf[2] == "$0x1,%ebx":
            return True
        if lf[1] == "cmp" and lf[2] in ("$0x2,%ebx", "$0x3,%ebx"):
            return False
        if mn == "jbe":
            return False
        return None
    m = Machine(pe, ins, gpr={"%rcx": P("man"), "%rdx": P("pt")})
    m.decider = decider_s
    m.run()
    p = [vec("man", 0x10 + 0xD0 * i) for i in range(4)]
    pt = vec("pt", 0)

    def area(a0, b0):
        c = cross(a0, b0)
        return dot(c, c)
    res = [area(sub(pt, p[1]), sub(p[3], p[2])), area(sub(pt, p[0]), sub(p[3], p[2])), area(sub(pt, p[0]), sub(p[3], p[1])), area(sub(pt, p[0]), sub(p[2], p[1]))]
    got = [c[1] for c in m.compares[-4:]]
    E = _E()
    good = area3 == 1 and len(m.compares) >= 8 and all(same(g, E(("abs", r.t))) for g, r in zip(got, res))
    print(f"btPersistentManifold::sortCachedPoints at VA {sort_va:#x}: gContactCalcArea3Points = {area3} in .data, the four areas |(pt - p_a) x (p_b - p_c)|^2 "
          "with left-to-right squares: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # ---- getCacheEntry: the squared distance of the first cached point
    ins = function_listing(pe, cache_va, 0x400)

    class Stop(Exception):
        pass
    m = Machine(pe, ins, gpr={"%rcx": P("man"), "%rdx": P("pt")})

    def decider_c(m, pc, mn, lf):
        if lf[1] == "cmp" and lf[2] == "$0x4,%r9":
            return False
        if lf[1] == "comiss":
            raise Stop()
        return None
    m.decider = decider_c
    try:
        m.run()
    except Stop:
        pass
    d = sub(p[0], pt)
    good = bool(m.compares) and same(m.compares[0][2], dot(d, d)) and same(m.compares[0][1], I("man", 0x364) * I("man", 0x364))
    print(f"btPersistentManifold::getCacheEntry at VA {cache_va:#x}: |p_i.localA - new.localA|^2, left to right, against the squared breaking threshold: "
          + ("as restated" if good else "MISMATCH"))
    ok &= good
    return ok



def check_add_contact_point(pe, add_cp, cache_va):
    """btManifoldResult::addContactPoint up to getCacheEntry: pointA = pointInWorld + normalOnB * depth and the two local points
    (invXform of body 0's and body 1's transforms), not swapped."""
    ins = function_listing(pe, add_cp, 0x1000)
    first_call = direct_calls(pe, ins)[0][1]
    rec = {}

    class Stop(Exception):
        pass

    def threshold(m):
        m.xmm["%xmm0"] = [("in", "thr", 0)] + [("const", 0.0)] * 3

    def cache(m):
        a = m.gpr["%rdx"]
        rec["localA"] = [m.cell((a[1], a[2] + 4 * k)) for k in range(3)]
        rec["localB"] = [m.cell((a[1], a[2] + 0x10 + 4 * k)) for k in range(3)]
        rec["worldA"] = [m.cell((a[1], a[2] + 0x30 + 4 * k)) for k in range(3)]
        raise Stop()

    def decider(m, pc, mn, lf):
        if lf[1] == "comiss":
            return False                  # depth <= breaking threshold
        if lf[1] == "cmp" and lf[2] == "%rsi,%r12":
            return True                   # je: the manifold's body 0 is the wrapper's: not swapped
        return None
    m = Machine(pe, ins, gpr={"%rcx": P("res"), "%rdx": P("nrm"), "%r8": P("pt")},
                ptr_loads={("res", 8): P("man"), ("res", 0x10): P("w0"), ("res", 0x18): P("w1"), ("w0", 0x10): P("obj0"), ("w1", 0x10): P("obj1"), ("man", 0x350): P("obj0")},
                hooks={first_call: threshold, cache_va: cache})
    m.xmm["%xmm3"] = [("in", "depth", 0)] + [("const", 0.0)] * 3
    m.decider = decider
    try:
        m.run()
    except Stop:
        pass
    n, pt, depth = vec("nrm", 0), vec("pt", 0), I("depth", 0)
    point_a = add(pt, scale(n, depth))

    def inv_xform(obj, v):
        rows = [vec(obj, 0x10), vec(obj, 0x20), vec(obj, 0x30)]
        d = sub(v, vec(obj, 0x40))
        return V3(rows[0].x * d.x + rows[1].x * d.y + rows[2].x * d.z, rows[0].y * d.x + rows[1].y * d.y + rows[2].y * d.z, rows[0].z * d.x + rows[1].z * d.y + rows[2].z * d.z)
    pairs = [(f"m_localPointA.{c}", g, w) for c, g, w in zip("xyz", rec.get("localA", [("missing",)] * 3), inv_xform("obj0", point_a))]
    pairs += [(f"m_localPointB.{c}", g, w) for c, g, w in zip("xyz", rec.get("localB", [("missing",)] * 3), inv_xform("obj1", pt))]
    pairs += [(f"m_positionWorldOnA.{c}", g, w) for c, g, w in zip("xyz", rec.get("worldA", [("missing",)] * 3), point_a)]
    return report(f"btManifoldResult::addContactPoint at VA {add_cp:#x} (pointA = point + normal * depth, the two invXform local points)", pairs)


# ---------------------------------------------------------------- 6. inertia tensor, capsule support, convex-plane contact
def check_inertia(pe, all_ins):
    from check_bullet_order import _function_start
    writers = sorted({pe.base + _function_start(pe, pc - pe.base) for pc, mn, ops in all_ins if mn == "movups" and ops == "%xmm0,0x180(%rcx)"})
    for fn in writers:
        ins = function_listing(pe, fn, 0x600)
        if sum(1 for _, mn, _ in ins if mn == "mulss") < 27:
            continue
        m = Machine(pe, ins, gpr={"%rcx": P("rb")}, hooks={COOKIE: lambda m: None})
        m.decider = lambda m, pc, mn, lf: None
        try:
            m.run()
        except Exception:
            continue
        Bm = [vec("rb", 0x10), vec("rb", 0x20), vec("rb", 0x30)]
        il = vec("rb", 0x210)
        pairs = []
        for r in range(3):
            s = V3(Bm[r].x * il.x, Bm[r].y * il.y, Bm[r].z * il.z)
            for c in range(3):
                pairs.append((f"m_invInertiaTensorWorld[{r}][{c}]", m.mem.get(("rb", 0x180 + 16 * r + 4 * c), ("missing",)), s.x * Bm[c].x + s.y * Bm[c].y + s.z * Bm[c].z))
        return report(f"btRigidBody::updateInertiaTensor at VA {fn:#x} (basis.scaled(invInertiaLocal) * basis.transpose(), sums left to right)", pairs)
    print("btRigidBody::updateInertiaTensor: not found  <-- MISMATCH")
    return False



def shape_vtable(pe, ctor_prefix):
    import check_bullet_order as cb
    _, code, rel = cb.coff_section(cb.OBJ, "?CreateShape@PhysicsSystem@@AEBA")
    sites = sorted(off for off, name in rel.items() if name.startswith(ctor_prefix))
    targets = set()
    for off in sites:
        pat = code[off - 13: off]
        for mm in re.finditer(re.escape(pat), pe.b):
            targets.add(pe.call_target(mm.start() + 12))
    ins = cb._disasm(pe, pe.base + targets.pop())
    lea = [x[0] for x in ins if x[1] == "lea" and "(%rip)" in x[2]][-1]
    vt = lea + 7 + struct.unpack("<i", pe.bytes_at_va(lea, 7)[3:])[0]
    return [cb._resolve(pe, struct.unpack("<Q", pe.bytes_at_va(vt + 8 * k, 8))[0]) for k in range(20)]


def check_local_inertia(pe):
    """calculateLocalInertia (vtable slot 7) of btBoxShape and btCapsuleShape.  The box's `mass / 12` is compiled as a product with
    0x3daaaaab (MSVC /fp:fast); the capsule's source already multiplies by the literal 0.08333333 (0x3daaaaaa)."""
    ok = True
    box, caps = shape_vtable(pe, "??0btBoxShape@@"), shape_vtable(pe, "??0btCapsuleShape@@")
    for name, slots in (("btBoxShape", box), ("btCapsuleShape", caps)):
        ins = function_listing(pe, slots[7], 0x400)
        if name == "btCapsuleShape":
            # m_upAxis is 1 (the constructor stores it: check_bullet_order.py): (upAxis + 2) % 3 = 0 through the 0x55555556 multiply
            k = next(i for i, (pc, mn, ops) in enumerate(ins) if mn == "movslq" and ops == "%r9d,%rax")
            hooks = {ins[k + 1][0]: lambda m: m.gpr.__setitem__("%rax", ("int", 0))}
        else:
            hooks = {}
        m = Machine(pe, ins, gpr={"%rcx": P("shape"), "%r8": P("out")}, ptr_loads={("shape", 0): P("vt"), ("shape", 0x50): ("int", 1)},
                    hooks={"*0x58(%rax)": lambda m, f=slots[11]: m.call_function(f), COOKIE: lambda m: None}, pc_hooks=hooks)
        m.xmm["%xmm1"] = [("in", "mass", 0)] + [("const", 0.0)] * 3
        m.decider = lambda m, pc, mn, lf: None
        m.run()
        mass = I("mass", 0)
        if name == "btBoxShape":
            h = [I("shape", 0x30 + 4 * k) + I("shape", 0x40) for k in range(3)]       # getHalfExtentsWithMargin
            l = [x + x for x in h]                                                      # 2 * h, compiled as h + h: the same float
            m12 = mass * C(struct.unpack("<f", struct.pack("<I", 0x3DAAAAAB))[0])
            want = [m12 * (l[1] * l[1] + l[2] * l[2]), m12 * (l[0] * l[0] + l[2] * l[2]), m12 * (l[0] * l[0] + l[1] * l[1])]
        else:
            r, hh = I("shape", 0x30), I("shape", 0x34)
            h = [r, r + hh, r]
            l = [x * 2.0 for x in h]
            sm = mass * C(struct.unpack("<f", struct.pack("<I", 0x3DAAAAAA))[0])
            x2, y2, z2 = [x * x for x in l]
            want = [sm * (y2 + z2), sm * (x2 + z2), sm * (x2 + y2)]
        ok &= report(f"{name}::calculateLocalInertia at VA {slots[7]:#x}", [(f"inertia.{c}", m.mem.get(("out", 4 * k), ("missing",)), w) for k, (c, w) in enumerate(zip("xyz", want))])
    return ok


def capsule_vtable(pe):
    import check_bullet_order as cb
    _, code, rel = cb.coff_section(cb.OBJ, "?CreateShape@PhysicsSystem@@AEBA")
    sites = sorted(off for off, name in rel.items() if name.startswith("??0btCapsuleShape@@"))
    targets = set()
    for off in sites:
        pat = code[off - 13: off]
        for mm in re.finditer(re.escape(pat), pe.b):
            targets.add(pe.call_target(mm.start() + 12))
    ins = cb._disasm(pe, pe.base + targets.pop())
    lea = next(x[0] for x in ins if x[1] == "lea" and "(%rip)" in x[2])
    vt = lea + 7 + struct.unpack("<i", pe.bytes_at_va(lea, 7)[3:])[0]
    return [cb._resolve(pe, struct.unpack("<Q", pe.bytes_at_va(vt + 8 * k, 8))[0]) for k in range(20)]


def check_capsule_support(pe):
    slots = capsule_vtable(pe)
    support, without_margin, get_margin = slots[15], slots[16], slots[11]
    ins = function_listing(pe, support, 0x300)
    E = _E()
    good = True
    for first_wins in (True, False):
        state = {"n": 0}

        def decider(m, pc, mn, lf):
            if lf[1] == "ucomiss":
                return False                       # je: margin == 0 -> it is not
            if lf[1] == "comiss" and mn == "jae":
                return True                        # the direction is long enough for both normalisations
            if lf[1] == "comiss" and mn == "jbe":
                state["n"] += 1
                return first_wins if state["n"] == 2 else False    # first vertex always beats -1e18; jbe taken = the second does not win
            return None
        m = Machine(pe, ins, gpr={"%rcx": P("caps"), "%rdx": P("out"), "%r8": P("dir")}, ptr_loads={("caps", 0): P("vt"), ("caps", 0x50): ("int", 1)},
                    hooks={"*0x80(%rax)": lambda m: m.call_function(without_margin), "*0x58(%rax)": lambda m: m.call_function(get_margin), COOKIE: lambda m: None})
        m.decider = decider
        m.run()
        d = vec("dir", 0)
        inv_len = 1.0 / E(("sqrtf", dot(d, d).t))
        vn = scale(d, inv_len)
        margin, hh = I("caps", 0x40), I("caps", 0x34)
        want = [C(0.0) + margin * vn.x, (hh + margin * vn.y) if first_wins else (margin * vn.y - hh), C(0.0) + margin * vn.z]
        good &= report(f"btCapsuleShape support vertex ({'+' if first_wins else '-'}halfHeight end): localGetSupportingVertex at VA {support:#x}, WithoutMargin at {without_margin:#x}",
                       [(f"vertex.{c}", m.mem[("out", 4 * k)], w) for k, (c, w) in enumerate(zip("xyz", want))])
    return good


def check_convex_plane(pe):
    """btConvexPlaneCollisionAlgorithm::processCollision: the function that reads gContactBreakingThreshold and SIMDSQRT12 and is
    not the (much larger) convex-convex one.  What reaches addContactPoint with the plane's transform the identity, normal
    (0, 1, 0), constant 0 — the only plane the world has — is compared after those values are substituted."""
    from check_bullet_order import _find_by_data_float
    from check_boxbox_order import _find_by_rdata_float
    both = [f for f in _find_by_data_float(pe, 0.02) if f in _find_by_rdata_float(pe, 0x3F3504F3)]
    sizes = {f: len(function_listing(pe, f, 0x4000)) for f in both}
    fn = min(sizes, key=sizes.get) if sizes else None
    if fn is None:
        print("btConvexPlaneCollisionAlgorithm::processCollision: not found  <-- MISMATCH")
        return False
    ins = function_listing(pe, fn, 0x1100)
    rec = {}

    class Stop(Exception):
        pass

    def support(m):
        d = m.gpr["%r8"]
        rec["dir"] = [m.cell((d[1], d[2] + 4 * k)) for k in range(3)]
        out = m.gpr["%rdx"]
        for k in range(4):
            m.mem[(out[1], out[2] + 4 * k)] = ("in", "vtx", 4 * k)
        m.gpr["%rax"] = out

    def add_contact(m):
        a, b = m.gpr["%rdx"], m.gpr["%r8"]
        rec["normal"] = [m.cell((a[1], a[2] + 4 * k)) for k in range(3)]
        rec["point"] = [m.cell((b[1], b[2] + 4 * k)) for k in range(3)]
        rec["depth"] = m.get("%xmm3")[0]
        raise Stop()

    def decider(m, pc, mn, lf):
        if mn == "cmovne":
            return False                       # not swapped: body 0 is the convex
        if lf[1] == "cmpq":
            return False                       # there is a manifold
        if lf[1] == "test" and lf[2] == "%cl,%cl":
            return False                       # hasCollision
        return None
    m = Machine(pe, ins, gpr={"%rcx": P("algo"), "%rdx": P("w0"), "%r8": P("w1"), "%r9": P("dinfo")}, stack_ptrs={0x28: P("result")},
                ptr_loads={("w0", 0x8): P("convex"), ("w0", 0x18): P("trC"), ("w1", 0x8): P("plane"), ("w1", 0x18): P("trP"), ("w0", 0x10): P("objC"),
                           ("w1", 0x10): P("objP"), ("algo", 0x18): P("man"), ("convex", 0): P("vtC"), ("result", 0): P("vtR")},
                hooks={"*0x78(%rax)": support, "*0x18(%rax)": add_contact, COOKIE: lambda m: None})
    m.decider = decider
    try:
        m.run()
    except Stop:
        pass
    # substitute the world's one plane: identity transform, normal (0, 1, 0), constant 0; fold the exact products x*1, x*0, x+0
    ident = {("trP", 0): 1.0, ("trP", 0x14): 1.0, ("trP", 0x28): 1.0, ("plane", 0x54): 1.0}
    zero_spaces = {"trP", "plane"}

    def fold(t):
        if not isinstance(t, tuple):
            return t
        if t[0] == "in":
            if (t[1], t[2]) in ident:
                return ("const", ident[(t[1], t[2])])
            if t[1] in zero_spaces:
                return ("const", 0.0)
            return t
        if t[0] in ("const", "opaque"):
            return t
        a = [fold(x) for x in t[1:]]
        z, one = ("const", 0.0), ("const", 1.0)
        if t[0] == "mul":
            if z in a:
                return z
            if a[0] == one:
                return a[1]
            if a[1] == one:
                return a[0]
        if t[0] == "add":
            if a[0] == z:
                return a[1]
            if a[1] == z:
                return a[0]
        if t[0] == "sub":
            if a[1] == z:
                return a[0]
            if a[0] == z:
                return ("neg", a[1])
        if t[0] == "neg" and a[0] == z:
            return z
        return (t[0],) + tuple(a)
    Cb = [vec("trC", 0), vec("trC", 0x10), vec("trC", 0x20)]
    oC = vec("trC", 0x30)
    vtx = vec("vtx", 0)
    in_plane = V3(*[o + ((r.x * vtx.x + r.y * vtx.y) + r.z * vtx.z) for r, o in zip(Cb, oC)])
    want_dir = [-Cb[1].x, -Cb[1].y, -Cb[1].z]
    depth = in_plane.y
    point = [in_plane.x, in_plane.y - depth, in_plane.z]
    pairs = [(f"direction handed to the support function.{c}", fold(norm2(g)), w) for c, g, w in zip("xyz", rec.get("dir", [("missing",)] * 3), want_dir)]
    pairs += [("depth", fold(norm2(rec.get("depth", ("missing",)))), depth)]
    pairs += [(f"pointInWorld.{c}", fold(norm2(g)), w) for c, g, w in zip("xyz", rec.get("point", [("missing",)] * 3), point)]
    pairs += [(f"normalOnB.{c}", fold(norm2(g)), w) for c, g, w in zip("xyz", rec.get("normal", [("missing",)] * 3), [C(0.0), C(1.0), C(0.0)])]
    return report(f"btConvexPlaneCollisionAlgorithm::processCollision at VA {fn:#x}, the single contact against the plane y = 0", pairs)


# ---------------------------------------------------------------- main
def text_listing():
    import subprocess
    from check_bx_order import EXE
    text = subprocess.run(["objdump", "-d", "--no-show-raw-insn", EXE], capture_output=True, text=True, check=True).stdout
    ins = []
    for line in text.splitlines():
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if m:
            ins.append((int(m.group(1), 16), m.group(2), m.group(3).split("#")[0].strip()))
    return ins


def main():
    from check_bx_order import EXE, Pe
    from check_bullet_order import _function_start
    if not os.path.exists(EXE):
        print("the reference build is not here; nothing checked")
        return 2
    pe = Pe(EXE)
    all_ins = text_listing()
    ok = check_setup_contact(pe, all_ins)
    stores = find_setup_contact(all_ins)
    setup_va = pe.base + _function_start(pe, stores[0] - pe.base)
    ok &= check_friction_and_convert(pe, setup_va)
    # convertBodies: the caller of the three gyroscopic functions (it calls getOrInitSolverBody, the most-called callee of convertContact's first two calls)
    cc = function_listing(pe, find_convert_contact(pe, setup_va)[0], 0x2000)
    get_or_init = direct_calls(pe, cc)[0][1]
    cands = set()
    va0, _, raw, rs = pe.secs[0]
    code = pe.b[raw: raw + rs]
    thunks = {get_or_init}
    for mm in re.finditer(b"\xe9", code[:0x10000]):
        o = mm.start()
        if pe.base + va0 + o + 5 + struct.unpack_from("<i", code, o + 1)[0] == get_or_init:
            thunks.add(pe.base + va0 + o)
    for mm in re.finditer(b"\xe8", code):
        o = mm.start()
        if o + 5 <= len(code) and pe.base + va0 + o + 5 + struct.unpack_from("<i", code, o + 1)[0] in thunks:
            cands.add(pe.base + _function_start(pe, va0 + o))
    gyro_ok = False
    for f in sorted(cands):
        callees = [t for _, t in direct_calls(pe, function_listing(pe, f, 0x1000))]
        small = [t for t in callees if 150 < len(function_listing(pe, t, 0x800)) < 400]
        if len(small) >= 3:
            gyro_ok = check_gyroscopic(pe, sorted(set(small)))
            break
    else:
        print("convertBodies: not found  <-- MISMATCH")
    ok &= gyro_ok
    # the manifold functions: refreshContactPoints is the last direct callee of the convex-plane algorithm; getCacheEntry and
    # addManifoldPoint -> sortCachedPoints hang off btManifoldResult::addContactPoint (reads SIMDSQRT12, calls getCacheEntry first)
    from check_bullet_order import _find_by_data_float
    from check_boxbox_order import _find_by_rdata_float
    both = [f for f in _find_by_data_float(pe, 0.02) if f in _find_by_rdata_float(pe, 0x3F3504F3)]
    cp_fn = min(both, key=lambda f: len(function_listing(pe, f, 0x4000)))
    cp_calls = [t for _, t in direct_calls(pe, function_listing(pe, cp_fn, 0x1100)) if t != COOKIE]
    refresh_va = cp_calls[-1]
    sqrt12 = _find_by_rdata_float(pe, 0x3F3504F3)
    add_cp = None
    for f in sqrt12:
        body = function_listing(pe, f, 0x1000)
        callees = [t for _, t in direct_calls(pe, body)]
        if 300 < len(body) < 1100 and len(callees) >= 3 and sum(1 for _, mn, ops in body if mn == "call" and ops.startswith("*") and "(%rip)" in ops) >= 4:
            add_cp, add_cp_callees = f, callees
            break
    if add_cp is None:
        print("btManifoldResult::addContactPoint: not found  <-- MISMATCH")
        ok = False
    else:
        cache_va = add_cp_callees[1]
        add_point = add_cp_callees[-1]
        sort_va = [t for _, t in direct_calls(pe, function_listing(pe, add_point, 0x400))][0]
        ok &= check_manifold(pe, refresh_va, sort_va, cache_va)
        ok &= check_add_contact_point(pe, add_cp, cache_va)
    ok &= check_inertia(pe, all_ins)
    ok &= check_local_inertia(pe)
    ok &= check_capsule_support(pe)
    ok &= check_convex_plane(pe)
    print("RESULT: " + ("setup rows, gyroscopic term, manifold refresh / sort / cache and the plane contact are compiled as restated" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
