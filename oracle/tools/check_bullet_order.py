#!/usr/bin/env python3
"""Pin the oracle's Bullet conversions (oracle/bullet_math.h) against the reference's COMPILED code, mechanically.

Bullet is not under /root/reference, but its header-inline functions are compiled INTO the reference's committed
build/SandboxCity.dir/RelWithDebInfo/PhysicsSystem.obj, with symbols:
  ?ToBtQuaternion@?A0x...@@...                 the reference's helper with btQuaternion::setEulerZYX inlined
                                               (src/physics/PhysicsSystem.cpp:40-45)
  ?setRotation@btMatrix3x3@@...                btMatrix3x3::setRotation(q)             -> bt::MatFromQuat
  ?getEulerZYX@btMatrix3x3@@...                btMatrix3x3::getEulerZYX(yaw,pitch,roll) -> bt::EulerZYXFromMat
  ?getRotation@btMatrix3x3@@...                btMatrix3x3::getRotation(q)             -> bt::QuatFromMat (kOrientBasis mode only)
The object file is read as bytes (COFF parser below), each function's section is disassembled with objdump, relocations
name the constants (__real@3f000000 ...) and the libm calls (sinf, cosf, asinf, atan2f, sqrtf), and the scalar SSE code is
executed symbolically exactly as in check_bx_order.py.  Data-dependent branches are enumerated path by path.

This pins the euler <-> quaternion <-> matrix conventions and operation order of SURVEY.md 8 a-11 / a-12.  It does not
pin the libm VALUES (the reference calls MSVC's sinf/cosf/asinf/atan2f; the oracle and the GPU path use the deterministic
routines of include/bge_detmath.h, bounded against the platform libm in tests/test_oracle_physics.py), nor the parts of
Bullet that exist only inside the symbol-less SandboxCity.exe (stepSimulation's integrator, updateAabbs).

Run in the build container only:  python oracle/tools/check_bullet_order.py
"""
import os
import re
import struct
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_bx_order import E, compare, execute, norm  # noqa: E402

OBJ = "/root/reference/build/SandboxCity.dir/RelWithDebInfo/PhysicsSystem.obj"


def coff_function(path, wanted):
    """(instructions, {pc_next: constant tree}, {pc_next: callee}) of the COMDAT function whose symbol contains `wanted`."""
    b = open(path, "rb").read()
    _, nsec, _, symoff, nsym, optsz, _ = struct.unpack_from("<HHIIIHH", b, 0)
    secs, off = [], 20 + optsz
    for _ in range(nsec):
        _, _, rsz, raw, reloc, _, nreloc, _, _ = struct.unpack_from("<IIIIIIHHI", b, off + 8)
        secs.append((rsz, raw, reloc, nreloc))
        off += 40
    strtab = symoff + nsym * 18

    def sname(e):
        if e[:4] == b"\0\0\0\0":
            o = struct.unpack_from("<I", e, 4)[0]
            return b[strtab + o: b.index(b"\0", strtab + o)].decode()
        return e[:8].rstrip(b"\0").decode()

    syms, i = {}, 0
    while i < nsym:
        e = b[symoff + i * 18: symoff + i * 18 + 18]
        _, sec, _, _, naux = struct.unpack_from("<IhHBB", e, 8)
        syms[i] = (sname(e), sec)
        i += 1 + naux
    hits = [(n, s) for n, s in syms.values() if wanted in n and s > 0 and not n.startswith("$")]
    assert len(hits) == 1, hits
    rsz, raw, reloc, nreloc = secs[hits[0][1] - 1]
    code = b[raw: raw + rsz]
    rel = {}
    for k in range(nreloc):
        va, si, _ = struct.unpack_from("<IIH", b, reloc + k * 10)
        rel[va] = syms[si][0]
    tmp = "/tmp/_bge_fn.bin"
    open(tmp, "wb").write(code)
    text = subprocess.run(["objdump", "-D", "-b", "binary", "-m", "i386:x86-64", "--no-show-raw-insn", tmp],
                          capture_output=True, text=True, check=True).stdout
    ins = []
    for line in text.splitlines():
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if m:
            ins.append((int(m.group(1), 16), m.group(2), m.group(3).split("#")[0].strip()))
    ins = [x for x in ins if x[1] != "int3"]
    consts, calls = {}, {}
    for k, (pc, mn, ops) in enumerate(ins):
        nxt = ins[k + 1][0] if k + 1 < len(ins) else pc + 16
        names = [s for va, s in rel.items() if pc < va < nxt]
        if not names:
            continue
        name = names[0]
        if mn == "call":
            calls[nxt] = name
        elif name.startswith("__real@"):
            consts[nxt] = ("const", struct.unpack("<f", struct.pack("<I", int(name[7:15], 16)))[0])
        elif name.startswith("__xmm@80000000"):
            consts[nxt] = ("signmask",)
        elif name.startswith("__xmm@7fffffff"):
            consts[nxt] = ("absmask",)
        elif name.startswith("__security"):
            consts[nxt] = ("opaque", name)
        else:
            raise AssertionError(name)
    return hits[0][0], ins, consts, calls


def C(x):
    return E(("const", struct.unpack("<f", struct.pack("<f", x))[0]))


def fn(name, *args):
    return E((name,) + tuple(a.t for a in args))


# ---------------------------------------------------------------- restatements (structure of oracle/bullet_math.h)
def quat_from_transform_euler(ex, ey, ez):
    """bt::QuatFromTransformEuler = QuatFromEulerZYX(yawZ = e.y, pitchY = e.x, rollX = e.z)."""
    yaw, pitch, roll = ey, ex, ez
    hy, hp, hr = yaw * 0.5, pitch * 0.5, roll * 0.5
    cy, sy, cp, sp, cr, sr = fn("cosf", hy), fn("sinf", hy), fn("cosf", hp), fn("sinf", hp), fn("cosf", hr), fn("sinf", hr)
    return {("q", 0): (sr * cp * cy - cr * sp * sy).t, ("q", 1): (cr * sp * cy + sr * cp * sy).t,
            ("q", 2): (cr * cp * sy - sr * sp * cy).t, ("q", 3): (cr * cp * cy + sr * sp * sy).t}


def mat_from_quat(q):
    x, y, z, w = q
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    one = C(1.0)
    m = [[one - (yy + zz), xy - wz, xz + wy], [xy + wz, one - (xx + zz), yz - wx], [xz - wy, yz + wx, one - (xx + yy)]]
    out = {4 * r + c: m[r][c].t for r in range(3) for c in range(3)}
    out.update({3: ("const", 0.0), 7: ("const", 0.0), 11: ("const", 0.0)})   # btVector3's fourth float
    return out


def euler_zyx_from_mat(m, path):
    """bt::EulerZYXFromMat; m[r][c]; returns yaw, pitch, roll."""
    K_PI = 3.1415926535897932384626433832795029
    if path in ("gimbal-up", "gimbal-down"):
        delta = fn("atan2f", m[0][0], m[0][2])
        if path == "gimbal-up":
            pitch = C(K_PI / 2.0)             # the compiler folds SIMD_PI / 2 to 0x3fc90fdb
            roll = pitch + delta
        else:
            pitch = C(-K_PI / 2.0)
            roll = delta - pitch             # -pitch + delta in the source; x - (-y) and y + x are the same float
        return C(0.0), pitch, roll
    clamped = E(("min", ("const", 1.0), ("max", ("const", -1.0), m[2][0].t)))   # btAsin clamps to [-1, 1]
    pitch = -fn("asinf", clamped)
    c = fn("cosf", pitch)
    roll = fn("atan2f", E(("div", m[2][1].t, c.t)), E(("div", m[2][2].t, c.t)))
    yaw = fn("atan2f", E(("div", m[1][0].t, c.t)), E(("div", m[0][0].t, c.t)))
    return yaw, pitch, roll


def quat_mul(a, b):
    """bt::QuatMul (btQuaternion operator*), components x, y, z, w."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return [aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz]


def integrate_orientation(q0, w, dt, path):
    """bt::IntegrateOrientation up to the normalised quaternion handed to setRotation, on one control path."""
    eps_thresh = C(0.5 * (3.1415926535897932384626433832795029 * 0.5))   # ANGULAR_MOTION_THRESHOLD
    f2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
    f = fn("sqrtf", f2)
    if path == "clamped":
        f = E(("div", eps_thresh.t, dt.t))
    if path == "taylor":
        k = dt * 0.5 - dt * dt * (dt * 0.020833333333) * f * f       # association as compiled, see bullet_math.h
    else:
        k = E(("div", fn("sinf", f * 0.5 * dt).t, f.t))
    dorn = [w[0] * k, w[1] * k, w[2] * k, fn("cosf", f * dt * 0.5)]
    p = quat_mul(dorn, q0)
    l2 = (p[0] * p[0] + p[1] * p[1]) + (p[2] * p[2] + p[3] * p[3])     # pairwise, as compiled
    s = E(("div", ("const", 1.0), fn("sqrtf", l2).t))
    return [(x * s).t for x in p]


def check_integrate_transform():
    """btTransformUtil::integrateTransform exists once in the exe (symbol-less); it is the only code that references the
    constant 1/48 = 0x3caaaaab of its Taylor branch, which is how it is found."""
    import numpy as np
    from check_bx_order import EXE, Pe, objdump, parse
    pe = Pe(EXE)
    b = pe.b
    va, _, raw, rs = pe.secs[0]
    rva_c = None
    rd_va, _, rd_raw, rd_rs = pe.secs[1]
    i = b.find(struct.pack("<I", 0x3CAAAAAB), rd_raw, rd_raw + rd_rs)
    assert i != -1 and b.find(struct.pack("<I", 0x3CAAAAAB), i + 1, rd_raw + rd_rs) == -1
    rva_c = rd_va + i - rd_raw
    code = np.frombuffer(b[raw: raw + rs], dtype=np.uint8).astype(np.int64)
    n = len(code) - 4
    d = code[:n] | (code[1:n + 1] << 8) | (code[2:n + 2] << 16) | (code[3:n + 3] << 24)
    d = np.where(d >= 2 ** 31, d - 2 ** 32, d)
    refs = np.nonzero(np.arange(n, dtype=np.int64) + va + 4 + d == rva_c)[0]
    assert len(refs) == 1, refs
    j = int(refs[0])
    while not (code[j - 1] == 0xCC and code[j - 2] == 0xCC):      # back to the int3 padding in front of the function
        j -= 1
    start = va + j
    ins = parse(objdump(EXE, [f"--start-address={pe.base + start:#x}", f"--stop-address={pe.base + start + 0x400:#x}"]))
    calls = [int(x[2], 16) for x in ins if x[1] == "call"]
    assert len(calls) == 4, calls                                   # sinf, cosf, getRotation, setRotation (in this order)
    sinf_t, cosf_t, getrot_t, setrot_t = calls
    pcs = {m: [x[0] for x in ins if x[1] == m] for m in ("jbe", "jae")}
    jbe, jae = pcs["jbe"], pcs["jae"]
    assert len(jbe) == 4 and len(jae) == 1, (jbe, jae)
    # jbe[0]: fAngle2 <= eps (skip sqrt)   jbe[1]: fAngle*dt <= threshold (skip clamp)   jae[0]: fAngle >= 0.001 (sin branch)
    # jbe[2]: length2 <= eps (skip normalise)   jbe[3]: length2 <= eps (copy the basis instead of setRotation)
    dt = E(("in", "dt", 0))
    w = [E(("in", "w", k)) for k in range(3)]
    q0 = [E(("in", "q0", k)) for k in range(4)]
    x0 = [E(("in", "cur", 12 + k)) for k in range(3)]
    v = [E(("in", "v", k)) for k in range(3)]
    probe_pc = next(x[0] for x in ins if x[1] == "movups")        # the predicted origin is assembled and stored here

    def hook_get(reg, mem, sp, out):
        for k in range(4):
            mem[sp + 0x20 + 4 * k] = q0[k].t                       # getRotation(&q) with q at 0x20(%rsp)

    def hook_set(reg, mem, sp, out):
        for k in range(4):
            out[("setRotation", k)] = mem[sp + 0x20 + 4 * k]

    ok = True
    for path, take in (("regular", [jbe[1], jae[0]]), ("taylor", [jbe[1]]), ("clamped", [jae[0]])):
        got = execute(ins, {"%rcx": "cur", "%rdx": "v", "%r8": "w"}, "%none", {"%xmm3": dt.t}, {}, pe.bytes_at_va,
                      calls={sinf_t: "sinf", cosf_t: "cosf"}, hooks={getrot_t: hook_get, setrot_t: hook_set}, take=tuple(take),
                      probes={probe_pc: ["%xmm1", "%xmm2", "%xmm3"]})
        want_q = integrate_orientation(q0, w, dt, path)
        bad = [k for k in range(4) if norm(got.get(("setRotation", k), ("missing",))) != norm(want_q[k])]
        want_x = [(x0[k] + v[k] * dt).t for k in range(3)]
        badx = [k for k, r in enumerate(("%xmm1", "%xmm2", "%xmm3")) if norm(got[("probe", probe_pc, r)]) != norm(want_x[k])]
        print(f"btTransformUtil::integrateTransform, {path}: quaternion {4 - len(bad)} of 4, origin {3 - len(badx)} of 3 identical"
              + ("" if not (bad or badx) else "  <-- MISMATCH"))
        for k in bad[:2]:
            print("   compiled   :", norm(got.get(("setRotation", k), ("missing",))))
            print("   restatement:", norm(want_q[k]))
        ok &= not (bad or badx)
    print(f"  (found at VA {pe.base + start:#x} through its reference to 0x3caaaaab; thresholds: "
          f"{[struct.unpack('<f', pe.bytes_at_va(pe.base + r, 4))[0] for r in sorted({rva_c - 4, rva_c + 4})]})")
    return ok


def check_external_force_impulse():
    """How gravity reaches the velocity.  Two candidates are linked into the exe (btRigidBody layout of this build:
    m_inverseMass at +0x1d0, m_totalForce at +0x220, m_linearVelocity at +0x1b0):
      * btSequentialImpulseConstraintSolver::initSolverBody: m_externalForceImpulse = totalForce * invMass * timeStep,
        added to the velocity when the solver writes back  -> (F * invMass) * dt
      * btRigidBody::integrateVelocities: m_linearVelocity += m_totalForce * (m_inverseMass * step) -> F * (invMass * dt)
    The first is executed symbolically; the second is shown to have no caller."""
    import numpy as np
    from check_bx_order import EXE, Pe, objdump, parse
    pe = Pe(EXE)
    b = pe.b
    va, _, raw, rs = pe.secs[0]
    text = b[raw: raw + rs]
    ok = True
    # initSolverBody: the only `mulss 0x220(%r8),%xmm1`
    sig = bytes.fromhex("f3410f59882002 0000".replace(" ", ""))
    hits = [m.start() for m in re.finditer(re.escape(sig), text)]
    assert len(hits) == 1, hits
    at = va + hits[0]
    ins = parse(objdump(EXE, [f"--start-address={pe.base + at - 0x14:#x}", f"--stop-address={pe.base + at + 0x60:#x}"]), multi_ret=True)
    # window: from the load of m_inverseMass (movss 0x1d0(%r8),%xmm3) to the vector assembly
    first = next(k for k, x in enumerate(ins) if x[1] == "movss" and x[2].startswith("0x1d0(%r8)"))
    last = next(k for k, x in enumerate(ins) if k > first and x[1] == "movss" and x[2] == "%xmm1,%xmm0")
    dt = E(("in", "dt", 0))
    got = execute(ins[first:last + 1], {"%r8": "rb"}, "%none", {"%xmm7": dt.t}, {}, pe.bytes_at_va,
                  probes={ins[last][0]: ["%xmm1", "%xmm2", "%xmm3"]})
    inv = E(("in", "rb", 0x1D0 // 4))
    want = [((E(("in", "rb", (0x220 + 4 * k) // 4)) * inv) * dt).t for k in range(3)]
    bad = [k for k, r in enumerate(("%xmm1", "%xmm2", "%xmm3")) if norm(got[("probe", ins[last][0], r)]) != norm(want[k])]
    print(f"initSolverBody: externalForceImpulse = (totalForce * invMass) * timeStep: {3 - len(bad)} of 3 components identical"
          + ("" if not bad else "  <-- MISMATCH"))
    ok &= not bad
    # btRigidBody::setGravity: the only function that stores a vector to m_gravity (+0x1f0): `movups %xmm3,0x1f0(%rcx)`
    sigg = bytes.fromhex("0f1199f0010000")
    # (a second function stores through the same encoding at another layout; setGravity is the one that divides by +0x1d0)
    hg = [m.start() for m in re.finditer(re.escape(sigg), text)
          if text.find(bytes.fromhex("f30f1099d0010000"), max(m.start() - 0x80, 0), m.start()) != -1]   # movss 0x1d0(%rcx),%xmm3
    assert len(hg) == 1, hg
    j = hg[0]
    while not (text[j - 1] == 0xCC and text[j - 2] == 0xCC):
        j -= 1
    gins = parse(objdump(EXE, [f"--start-address={pe.base + va + j:#x}", f"--stop-address={pe.base + va + hg[0] + 7:#x}"]), multi_ret=True)
    shuf = next(x[0] for x in gins if x[1] == "shufps")
    got = execute(gins, {"%rcx": "rb", "%rdx": "acc"}, "%none", {}, {}, pe.bytes_at_va, probes={shuf: ["%xmm0", "%xmm1", "%xmm2"]})
    want = [("div", ("in", "acc", k), ("in", "rb", 0x1D0 // 4)) for k in range(3)]
    bad = [k for k, r in enumerate(("%xmm0", "%xmm1", "%xmm2")) if norm(got[("probe", shuf, r)]) != norm(want[k])]
    print(f"btRigidBody::setGravity: m_gravity = acceleration / m_inverseMass (one division per component): {3 - len(bad)} of 3 identical"
          + ("" if not bad else "  <-- MISMATCH"))
    ok &= not bad
    # integrateVelocities: `mulss 0x1d0(%rcx),%xmm4` followed within 64 bytes by `mulss 0x220(%rcx),%xmm0`
    s1, s2 = bytes.fromhex("f30f59a1d0010000"), bytes.fromhex("f30f598120020000")
    cand = [m.start() for m in re.finditer(re.escape(s1), text) if text.find(s2, m.start(), m.start() + 64) != -1]
    assert len(cand) == 1, cand
    j = cand[0]
    while not (text[j - 1] == 0xCC and text[j - 2] == 0xCC):
        j -= 1
    fn_rva = va + j
    code = np.frombuffer(text, dtype=np.uint8).astype(np.int64)
    n = len(code) - 5
    d = code[1:n + 1] | (code[2:n + 2] << 8) | (code[3:n + 3] << 16) | (code[4:n + 4] << 24)
    d = np.where(d >= 2 ** 31, d - 2 ** 32, d)
    tgt = np.arange(n, dtype=np.int64) + va + 5 + d

    def refs(rva, opcode):
        return [int(i) + va for i in np.nonzero((code[:n] == opcode) & (tgt == rva))[0]]

    thunks = refs(fn_rva, 0xE9)
    callers = refs(fn_rva, 0xE8) + [c for t in thunks for c in refs(t, 0xE8) + refs(t, 0xE9)]
    print(f"btRigidBody::integrateVelocities (F * (invMass * dt)) at VA {pe.base + fn_rva:#x}: {len(thunks)} thunk(s), "
          f"{len(callers)} callers -> {'never called' if not callers else 'CALLED'}")
    ok &= not callers
    return ok


def main():
    ok = True
    # 1. ToBtQuaternion(euler): result through rcx (hidden return pointer), euler through rdx
    name, ins, consts, calls = coff_function(OBJ, "?ToBtQuaternion@")
    got = execute(ins, {"%rdx": "e"}, {"%rcx": "q"}, named_consts=consts, named_calls=calls)
    e = [E(("in", "e", k)) for k in range(3)]
    want = quat_from_transform_euler(*e)
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    print(f"ToBtQuaternion / btQuaternion::setEulerZYX(euler.y, euler.x, euler.z): {4 - len(bad)} of 4 components identical"
          + ("" if not bad else f"  <-- MISMATCH {bad}"))
    ok &= not bad

    # 2. btMatrix3x3::setRotation(q): this = rcx (3 rows of 4 floats), q = rdx
    name, ins, consts, calls = coff_function(OBJ, "?setRotation@btMatrix3x3@@")
    got = execute(ins, {"%rdx": "q"}, "%rcx", named_consts=consts, named_calls=calls)
    want = mat_from_quat([E(("in", "q", k)) for k in range(4)])
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    print(f"btMatrix3x3::setRotation: {len(want) - len(bad)} of {len(want)} stored floats identical" + ("" if not bad else f"  <-- MISMATCH {bad}"))
    ok &= not bad

    # 3. btMatrix3x3::getEulerZYX(yaw&, pitch&, roll&, solution_number = 1): this = rcx, rdx / r8 / r9 = results
    name, ins, consts, calls = coff_function(OBJ, "?getEulerZYX@btMatrix3x3@@")
    jb = [i[0] for i in ins if i[1] == "jb"]          # |m20| >= 1 ?   (taken: the regular case)
    jbe = [i[0] for i in ins if i[1] == "jbe"]        # m20 > 0 ?      (taken: gimbal locked down)
    je = [i[0] for i in ins if i[1] == "je"]          # solution_number == 1
    assert len(jb) == 1 and len(jbe) == 1 and len(je) == 1, (jb, jbe, je)
    m = [[E(("in", "m", 4 * r + c)) for c in range(3)] for r in range(3)]
    for path, take in (("regular", jb + je), ("gimbal-up", je), ("gimbal-down", jbe + je)):
        got = execute(ins, {"%rcx": "m"}, {"%rdx": "yaw", "%r8": "pitch", "%r9": "roll"}, named_consts=consts, named_calls=calls,
                      take=tuple(take))
        yaw, pitch, roll = euler_zyx_from_mat(m, path)
        want = {("yaw", 0): yaw.t, ("pitch", 0): pitch.t, ("roll", 0): roll.t}
        bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
        print(f"btMatrix3x3::getEulerZYX, {path}: {3 - len(bad)} of 3 angles identical" + ("" if not bad else f"  <-- MISMATCH {bad}"))
        for k in bad:
            print("   compiled   :", norm(got.get(k, ("missing",))))
            print("   restatement:", norm(want[k]))
        ok &= not bad
    # 4. btMatrix3x3::getRotation(q), the trace > 0 side (the other side indexes the matrix with run-time i, j, k; it was
    #    read by hand: i = m00 < m11 ? (m11 < m22 ? 2 : 1) : (m00 < m22 ? 2 : 0), s = sqrt(m[i][i] - m[j][j] - m[k][k] + 1),
    #    t[i] = s / 2, s = 0.5 / s, t[3] = (m[k][j] - m[j][k]) s, t[j] = (m[j][i] + m[i][j]) s, t[k] = (m[k][i] + m[i][k]) s)
    name, ins, consts, calls = coff_function(OBJ, "?getRotation@btMatrix3x3@@")
    ja = [i[0] for i in ins if i[1] == "ja"]
    got = execute(ins, {"%rcx": "m"}, {"%rdx": "q"}, named_consts=consts, named_calls=calls, take=())   # jbe not taken: trace > 0
    m = [[E(("in", "m", 4 * r + c)) for c in range(3)] for r in range(3)]
    trace = m[0][0] + m[1][1] + m[2][2]
    s = fn("sqrtf", trace + 1.0)
    s2 = E(("div", ("const", 0.5), s.t))
    want = {("q", 0): ((m[2][1] - m[1][2]) * s2).t, ("q", 1): ((m[0][2] - m[2][0]) * s2).t,
            ("q", 2): ((m[1][0] - m[0][1]) * s2).t, ("q", 3): (s * 0.5).t}
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    print(f"btMatrix3x3::getRotation, trace > 0: {4 - len(bad)} of 4 components identical" + ("" if not bad else f"  <-- MISMATCH {bad}"))
    ok &= not bad
    ok &= check_integrate_transform()
    ok &= check_external_force_impulse()
    print("RESULT:", "the restatement has the compiled code's operation order" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
