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


def coff_section(path, wanted):
    """(symbol, code bytes, {offset: relocation symbol}) of the COMDAT function whose symbol contains `wanted`."""
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
    hits = [(n, s) for n, s in syms.values() if wanted in n and s > 0 and not n.startswith("$") and not n.startswith("?dtor$")]
    assert len(hits) == 1, hits
    rsz, raw, reloc, nreloc = secs[hits[0][1] - 1]
    code = b[raw: raw + rsz]
    rel = {}
    for k in range(nreloc):
        va, si, _ = struct.unpack_from("<IIH", b, reloc + k * 10)
        rel[va] = syms[si][0]
    return hits[0][0], code, rel


def coff_function(path, wanted):
    """(instructions, {pc_next: constant tree}, {pc_next: callee}) of the COMDAT function whose symbol contains `wanted`."""
    symbol, code, rel = coff_section(path, wanted)
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
    return symbol, ins, consts, calls


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


def concretise_indices(ins, start_pc, gpr0):
    """btMatrix3x3::getRotation indexes the matrix with run-time i, j, k.  For one value of i the integer code that derives
    j = (i + 1) % 3, k = (i + 2) % 3 and the element offsets is executed CONCRETELY (a dozen mov / lea / mul / shr / sub
    forms), and every indexed memory operand `disp(base,index,scale)` is rewritten to the plain `disp'(base)` it denotes, so
    that the float code can then be executed symbolically as usual."""
    g = dict(gpr0)
    legacy = {"%eax": "%rax", "%ecx": "%rcx", "%edx": "%rdx", "%esi": "%rsi", "%edi": "%rdi", "%ebx": "%rbx", "%ebp": "%rbp"}

    def reg64(r):
        m_ = re.fullmatch(r"%r(\d+)[dwb]", r)
        return f"%r{m_.group(1)}" if m_ else legacy.get(r, r)

    out = []
    for pc, mn, ops in ins:
        if pc < start_pc:
            out.append((pc, mn, ops))
            continue
        parts = [p_.strip() for p_ in re.split(r",(?![^(]*\))", ops)] if ops else []
        m = None
        new_ops = ops
        for k_, p_ in enumerate(parts):
            m = re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+)?,(%\w+),(\d)\)", p_) or re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+),(%\w+)()\)", p_)
            if m and mn != "lea":
                disp = int(m.group(1), 16) if m.group(1) else 0
                base, idx, sc = m.group(2), m.group(3), int(m.group(4) or 1)
                assert reg64(idx) in g, (hex(pc), ops)
                parts[k_] = f"{disp + g[reg64(idx)] * sc:#x}({base})"
                new_ops = ",".join(parts)
        if mn == "lea":
            m = re.fullmatch(r"(-?0x[0-9a-f]+|)\((%\w+)?(?:,(%\w+)(?:,(\d))?)?\)", parts[0])
            base, idx = m.group(2), m.group(3)
            if (base is None or reg64(base) in g) and (idx is None or reg64(idx) in g):
                v = (int(m.group(1), 16) if m.group(1) else 0) + (g[reg64(base)] if base else 0) + (g[reg64(idx)] * int(m.group(4) or 1) if idx else 0)
                g[reg64(parts[1])] = v & 0xFFFFFFFFFFFFFFFF
        elif mn in ("mov", "movslq") and len(parts) == 2 and parts[1].startswith("%") and "(" not in parts[1]:
            if parts[0].startswith("$"):
                g[reg64(parts[1])] = int(parts[0][1:], 16)
            elif reg64(parts[0]) in g:
                g[reg64(parts[1])] = g[reg64(parts[0])]
                if mn == "movslq":
                    g[("set", reg64(parts[1]))] = g[reg64(parts[0])]      # remembered: the register is restored later
            else:
                g.pop(reg64(parts[1]), None)
        elif mn == "mul" and reg64(parts[0]) in g and "%rax" in g:     # edx:eax = eax * r32
            prod = (g["%rax"] & 0xFFFFFFFF) * (g[reg64(parts[0])] & 0xFFFFFFFF)
            g["%rax"], g["%rdx"] = prod & 0xFFFFFFFF, prod >> 32
        elif mn == "shr" and len(parts) == 1 and reg64(parts[0]) in g:
            g[reg64(parts[0])] >>= 1
        elif mn == "sub" and len(parts) == 2 and reg64(parts[0]) in g and reg64(parts[1]) in g and parts[1] != "%rsp":
            g[reg64(parts[1])] = (g[reg64(parts[1])] - g[reg64(parts[0])]) & 0xFFFFFFFF
        out.append((pc, mn, new_ops))
    return out, g


def check_get_rotation_other_side():
    """btMatrix3x3::getRotation for trace <= 0: the choice of i as text, then the three cases executed symbolically."""
    name, ins, consts, calls = coff_function(OBJ, "?getRotation@btMatrix3x3@@")
    text = [(x[1], x[2]) for x in ins]
    k0 = text.index(("xor", "%r9d,%r9d"))
    k1 = next(k for k, x in enumerate(text) if k > k0 and x[0] == "lea")
    sel = [t for t in text[k0:k1] if t[0] not in ("mov",) or t[1].startswith("$")]
    # xmm3 = m00, xmm2 = m11, xmm4 = m22 (loaded at the top);  i = m00 < m11 ? (m11 < m22 ? 2 : 1) : (m00 < m22 ? 2 : 0)
    want_sel = [("xor", "%r9d,%r9d"), ("comiss", "%xmm3,%xmm2"), ("jbe", None), ("comiss", "%xmm2,%xmm4"), ("seta", "%r9b"),
                ("inc", "%r9d"), ("jmp", None), ("comiss", "%xmm3,%xmm4"), ("mov", "$0x2,%eax"), ("cmova", "%eax,%r9d")]
    ok = len(sel) == len(want_sel) and all(a[0] == b[0] and (b[1] is None or a[1] == b[1]) for a, b in zip(sel, want_sel))
    top = [t for t in text[:k0] if t[0] == "movss"][:3]
    ok &= top == [("movss", "(%rcx),%xmm3"), ("movss", "0x14(%rcx),%xmm2"), ("movss", "0x28(%rcx),%xmm4")]
    print("btMatrix3x3::getRotation, trace <= 0: i = m00 < m11 ? (m11 < m22 ? 2 : 1) : (m00 < m22 ? 2 : 0): "
          + ("as restated" if ok else "MISMATCH"))
    jbe_trace = next(x[0] for x in ins if x[1] == "jbe")
    start = ins[k1][0]
    m = [[E(("in", "m", 4 * r + c)) for c in range(3)] for r in range(3)]
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        rewritten, g = concretise_indices(ins, start, {"%r9": i})
        good_idx = g.get(("set", "%r14")) == j and g.get(("set", "%r15")) == k
        # skip the selection code: it only sets %r9, which is given
        body = [x for x in rewritten if not (ins[k0][0] <= x[0] < start)]
        got = execute(body, {"%rcx": "m"}, {"%rdx": "q"}, named_consts=consts, named_calls=calls, take=(jbe_trace,))
        s_ = fn("sqrtf", ((m[i][i] - m[j][j]) - m[k][k]) + 1.0)
        s2 = E(("div", ("const", 0.5), s_.t))
        t = [None] * 4
        t[i] = (s_ * 0.5).t
        t[3] = ((m[k][j] - m[j][k]) * s2).t
        t[j] = ((m[j][i] + m[i][j]) * s2).t
        t[k] = ((m[k][i] + m[i][k]) * s2).t
        bad = [c for c in range(4) if norm(got.get(("q", c), ("missing",))) != norm(t[c])]
        print(f"btMatrix3x3::getRotation, trace <= 0, i = {i} (j = {j}, k = {k} from the compiled modulo code: {'yes' if good_idx else 'NO'}): "
              f"{4 - len(bad)} of 4 components identical" + ("" if not bad and good_idx else f"  <-- MISMATCH {bad}"))
        for c in bad[:2]:
            print("   compiled   :", norm(got.get(("q", c), ("missing",))))
            print("   restatement:", norm(t[c]))
        ok &= not bad and good_idx
    return ok


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


def _resolve(pe, va):
    """Through an incremental-link thunk (jmp rel32), if there is one."""
    fo = pe.r2f(va - pe.base)
    return va + 5 + struct.unpack_from("<i", pe.b, fo + 1)[0] if pe.b[fo] == 0xE9 else va


def _disasm(pe, va, size=0x400, multi_ret=False):
    from check_bx_order import EXE, objdump, parse
    return parse(objdump(EXE, [f"--start-address={va:#x}", f"--stop-address={va + size:#x}"]), multi_ret=multi_ret)


def _rip_refs(pe, rva):
    """RVAs of the disp32 fields in .text that resolve (as the last 4 bytes of an instruction) to `rva`."""
    import numpy as np
    va, _, raw, rs = pe.secs[0]
    code = np.frombuffer(pe.b[raw: raw + rs], dtype=np.uint8).astype(np.int64)
    n = len(code) - 4
    d = code[:n] | (code[1:n + 1] << 8) | (code[2:n + 2] << 16) | (code[3:n + 3] << 24)
    d = np.where(d >= 2 ** 31, d - 2 ** 32, d)
    return [int(i) + va for i in np.nonzero(np.arange(n, dtype=np.int64) + va + 4 + d == rva)[0]]


def _function_start(pe, rva):
    va, _, raw, rs = pe.secs[0]
    j = raw + rva - va
    while not (pe.b[j - 1] == 0xCC and pe.b[j - 2] == 0xCC):
        j -= 1
    return va + j - raw


def _virtual(ins, slot, token):
    """`call *slot(%rax)` -> `call token`, so that a hook can model the virtual callee."""
    return [(pc, mn, f"{token:#x}") if mn == "call" and ops == f"*{slot:#x}(%rax)" else (pc, mn, ops) for pc, mn, ops in ins]


def check_box_aabb():
    """btBoxShape (constructor, setSafeMargin, setMargin, getAabb) and btCollisionWorld::updateSingleAabb, all inside the
    symbol-less exe.  The constructor is found through the call PhysicsSystem::CreateShape makes to it (the relocation in
    PhysicsSystem.obj names ??0btBoxShape@@...; the bytes in front of that call are found again in the exe), everything else
    from there: the base constructors through its first call, getAabb / getMargin / setMargin through the vtable it stores.
    btBoxShape of this build: m_localScaling +0x20, m_implicitShapeDimensions +0x30, m_collisionMargin +0x40."""
    from check_bx_order import EXE, Pe
    pe = Pe(EXE)
    ok = True
    _, code, rel = coff_section(OBJ, "?CreateShape@PhysicsSystem@@AEBA")
    sites = sorted(off for off, name in rel.items() if name.startswith("??0btBoxShape@@"))
    assert sites, "CreateShape does not construct a btBoxShape?"
    targets = set()
    for off in sites:
        pat = code[off - 13: off]                                   # 12 bytes of set-up + the E8 opcode
        assert pat[-1] == 0xE8
        for m in re.finditer(re.escape(pat), pe.b):
            targets.add(pe.call_target(m.start() + 12))
    assert len(targets) == 1, targets
    ctor = pe.base + targets.pop()
    ins = _disasm(pe, ctor)
    calls = [int(x[2], 16) for x in ins if x[1] == "call"]
    assert len(calls) == 2, calls                                   # btPolyhedralConvexShape(), setMargin(safeMargin)
    # ---- base constructors: btPolyhedralConvexShape() -> btConvexInternalShape(): scaling (1,1,1), margin 0.04
    base1 = _disasm(pe, _resolve(pe, calls[0]))
    base2 = _disasm(pe, _resolve(pe, next(int(x[2], 16) for x in base1 if x[1] == "call")))
    imm = {x[2] for x in base2 if x[1] in ("movl", "movq")}
    want_imm = {"$0x3f800000,0x20(%rbx)", "$0x3f800000,0x24(%rbx)", "$0x3f800000,0x28(%rbx)", "$0x3d23d70a,0x40(%rbx)"}
    good = want_imm <= imm
    print(f"btConvexInternalShape(): localScaling = (1, 1, 1), collisionMargin = 0x3d23d70a (0.04f): {'as restated' if good else 'MISMATCH'}")
    ok &= good
    # ---- constructor, part 1: implicitShapeDimensions = halfExtents * localScaling - margin
    first_shuf = next(x[0] for x in ins if x[1] == "shufps")
    k_cmp = next(k for k, x in enumerate(ins) if x[1] == "comiss")
    noop = {calls[0]: lambda reg, mem, sp, out: None}
    got = execute(ins[:k_cmp], {"%rdx": "he", "%rcx": "this"}, "%none", {}, {}, pe.bytes_at_va, hooks=noop,
                  probes={first_shuf: ["%xmm3", "%xmm1", "%xmm2"]})
    margin = E(("in", "this", 0x40 // 4))
    want = [(E(("in", "he", k)) * E(("in", "this", 0x20 // 4 + k)) - margin).t for k in range(3)]
    bad = [k for k, r in enumerate(("%xmm3", "%xmm1", "%xmm2")) if norm(got[("probe", first_shuf, r)]) != norm(want[k])]
    print(f"btBoxShape(halfExtents): implicit = halfExtents * localScaling - margin: {3 - len(bad)} of 3 identical"
          + ("" if not bad else "  <-- MISMATCH"))
    ok &= not bad
    # ---- constructor, part 2: setSafeMargin(halfExtents, 0.1).  minAxis() is integer selection, compared as text:
    k_mul = next(k for k, x in enumerate(ins) if k > k_cmp and x[1] == "mulss")
    sel = [(x[1], x[2]) for x in ins[k_cmp - 3: k_mul]]
    want_sel = [("movss", "(%rdi),%xmm2"), ("movss", "0x4(%rdi),%xmm1"), ("movss", "0x8(%rdi),%xmm0"),
                ("comiss", "%xmm1,%xmm2"), ("jae", None), ("mov", "$0x2,%eax"), ("comiss", "%xmm2,%xmm0"), ("cmova", "%ecx,%eax"),
                ("jmp", None), ("mov", "%ecx,%eax"), ("comiss", "%xmm1,%xmm0"), ("setbe", "%al"), ("inc", "%eax"),
                ("movss", "(%rdi,%rax,4),%xmm1")]
    # x < y ? (x < z ? 0 : 2) : (y < z ? 1 : 2)   [ecx = 0]
    good = len(sel) == len(want_sel) and all(a[0] == b[0] and (b[1] is None or a[1] == b[1]) for a, b in zip(sel, want_sel))
    print(f"btVector3::minAxis of the half extents, x < y ? (x < z ? 0 : 2) : (y < z ? 1 : 2): {'as restated' if good else 'MISMATCH'}")
    ok &= good
    seen = {}

    def hook_set_margin(reg, mem, sp, out):
        seen["arg"] = reg["%xmm1"]

    tail = ins[k_mul:]
    cmp_pc = next(x for x in tail if x[1] == "comiss")
    got = execute(tail, {"%rbx": "this"}, "%none", {"%xmm1": ("in", "he", "min"), "%xmm4": margin.t}, {}, pe.bytes_at_va,
                  hooks={calls[1]: hook_set_margin})
    good = (norm(seen.get("arg", ("missing",))) == norm(("mul", ("in", "he", "min"), C(0.1).t))
            and cmp_pc[2] == "%xmm4,%xmm1" and tail[tail.index(cmp_pc) + 1][1] == "jae")
    print("setSafeMargin: safe = 0.1f * min; setMargin(safe) iff safe < margin: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # ---- the vtable the constructor stores: getAabb (slot 1), getMargin (slot 11), setMargin (the call above)
    lea = next(x[0] for x in ins if x[1] == "lea" and "(%rip)" in x[2])
    raw = pe.bytes_at_va(lea, 7)
    assert raw[:3] == b"\x48\x8d\x05", raw.hex()
    vtable = lea + 7 + struct.unpack("<i", raw[3:])[0]
    slot = lambda k: _resolve(pe, struct.unpack("<Q", pe.bytes_at_va(vtable + 8 * k, 8))[0])
    gm = [(x[1], x[2]) for x in _disasm(pe, slot(0x58 // 8))]
    good = gm == [("movss", "0x40(%rcx),%xmm0"), ("ret", "")]
    print(f"btBoxShape vtable at {vtable:#x}: getMargin (slot 11) returns m_collisionMargin: {'yes' if good else 'MISMATCH'}")
    ok &= good

    def get_margin(reg, mem, sp, out):
        return out.get(("this", 0x40 // 4), margin.t)               # the member as it is when the call is made

    # ---- setMargin(new): implicit = (implicit + old) - new
    sm = _virtual(_disasm(pe, _resolve(pe, calls[1])), 0x58, 0xFFFF58)
    packs = [x for x in sm if x[1] == "movss" and x[2].endswith(",%xmm13") and x[2].startswith("%xmm")]
    first_shuf = next(x[0] for x in sm if x[1] == "shufps")
    assert [x[2] for x in packs[:2]] == ["%xmm10,%xmm13", "%xmm12,%xmm13"], packs
    new = E(("in", "new", 0))
    got = execute(sm, {"%rcx": "this"}, {"%rcx": "this"}, {"%xmm1": new.t}, {}, pe.bytes_at_va, hooks={0xFFFF58: get_margin},
                  probes={first_shuf: ["%xmm13"], packs[0][0]: ["%xmm10"], packs[1][0]: ["%xmm12"]})
    trees = [got[("probe", first_shuf, "%xmm13")], got[("probe", packs[0][0], "%xmm10")], got[("probe", packs[1][0], "%xmm12")]]
    want = [((E(("in", "this", 0x30 // 4 + k)) + margin) - new).t for k in range(3)]
    bad = [k for k in range(3) if norm(trees[k]) != norm(want[k])]
    good = not bad and norm(got.get(("this", 0x40 // 4))) == norm(new.t)
    print(f"btBoxShape::setMargin: implicit = (implicit + oldMargin) - newMargin, margin = new: {3 - len(bad)} of 3 identical"
          + ("" if good else "  <-- MISMATCH"))
    ok &= good
    # ---- getAabb(t, mn, mx): half = implicit + margin; extent_i = (|r_i0| hx + |r_i1| hy) + |r_i2| hz; origin -/+ extent
    ga = _virtual(_disasm(pe, slot(1)), 0x58, 0xFFFF58)
    # the origin is loaded packed (movaps 0x30(%rdi),%xmm9) and y / z are taken out with shufps: rewritten as the scalar loads
    # they are, after checking that %xmm9 still holds the origin at that point
    k_mod = next(k for k, x in enumerate(ga) if x[2].endswith(",%xmm9") and x[1] != "movaps")
    lanes = {"$0xaa": "0x38(%rdi)", "$0x55": "0x34(%rdi)"}
    for k, (pc, mn, ops) in enumerate(ga):
        if mn == "shufps":
            imm8, src, dst = ops.split(",")
            assert src == "%xmm9" and k < k_mod and any(g[1:] == ("movaps", f"%xmm9,{dst}") for g in ga[k - 4:k]), (hex(pc), ops)
            ga[k] = (pc, "movss", f"{lanes[imm8]},{dst}")
    got = execute(ga, {"%rcx": "this", "%rdx": "t"}, {"%r8": "mn", "%r9": "mx"}, {}, {}, pe.bytes_at_va, hooks={0xFFFF58: get_margin})
    t = lambda r, c: E(("abs", ("in", "t", 4 * r + c)))
    half = [E(("in", "this", 0x30 // 4 + k)) + margin for k in range(3)]
    ext = [(t(r, 0) * half[0] + t(r, 1) * half[1]) + t(r, 2) * half[2] for r in range(3)]
    want = {}
    for r in range(3):
        want[("mn", r)] = (E(("in", "t", 12 + r)) - ext[r]).t
        want[("mx", r)] = (E(("in", "t", 12 + r)) + ext[r]).t
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    print(f"btBoxShape::getAabb: origin -/+ ((|r0| hx + |r1| hy) + |r2| hz), h = implicit + margin: {6 - len(bad)} of 6 identical"
          + ("" if not bad else f"  <-- MISMATCH {bad}"))
    for k in bad[:2]:
        print("   compiled   :", norm(got.get(k, ("missing",))))
        print("   restatement:", norm(want[k]))
    ok &= not bad
    ok &= check_capsule_aabb(pe, code, rel)
    ok &= check_update_single_aabb(pe)
    return ok



def check_capsule_aabb(pe, code, rel):
    """btCapsuleShape(radius, height) as PhysicsSystem::CreateShape calls it, and btCapsuleShape::getAabb (vtable slot 1).
    The up axis is a run-time member (m_upAxis, +0x50) that the constructor sets to 1: the three indexed accesses of getAabb
    are rewritten for that value before the symbolic execution."""
    ok = True
    sites = sorted(off for off, name in rel.items() if name.startswith("??0btCapsuleShape@@"))
    assert sites
    targets = set()
    for off in sites:
        pat = code[off - 13: off]
        assert pat[-1] == 0xE8
        for m in re.finditer(re.escape(pat), pe.b):
            targets.add(pe.call_target(m.start() + 12))
    assert len(targets) == 1, targets
    ins = _disasm(pe, pe.base + targets.pop())
    text = [(x[1], x[2]) for x in ins]
    call = next(int(x[2], 16) for x in ins if x[1] == "call")
    got = execute(ins, {}, {"%rcx": "this"}, {"%xmm1": ("in", "radius", 0), "%xmm2": ("in", "height", 0)}, {}, pe.bytes_at_va,
                  hooks={call: lambda reg, mem, sp, out: None})
    r, h = E(("in", "radius", 0)), E(("in", "height", 0))
    want = {("this", 0x30 // 4): r.t, ("this", 0x34 // 4): (h * 0.5).t, ("this", 0x38 // 4): r.t, ("this", 0x40 // 4): r.t}
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    good = not bad and ("movl", "$0x1,0x50(%rbx)") in text
    print("btCapsuleShape(radius, height): implicit = (radius, 0.5f * height, radius), margin = radius, upAxis = 1: "
          + ("as restated" if good else f"MISMATCH {bad}"))
    ok &= good
    lea = next(x[0] for x in ins if x[1] == "lea" and "(%rip)" in x[2])
    raw = pe.bytes_at_va(lea, 7)
    vtable = lea + 7 + struct.unpack("<i", raw[3:])[0]
    ga = _disasm(pe, _resolve(pe, struct.unpack("<Q", pe.bytes_at_va(vtable + 8, 8))[0]))
    assert ga[2][1:] == ("movslq", "0x50(%rcx),%r11")               # r11 = m_upAxis;  rax = (r11 + 2) % 3 by the 0x55555556 multiply
    fixed = {"0x30(%rcx,%rax,4),%xmm0": "0x30(%rcx),%xmm0", "0x30(%rcx,%r11,4),%xmm0": "0x34(%rcx),%xmm0",
             "%xmm0,(%rsp,%r11,4)": "%xmm0,0x4(%rsp)"}
    n_fixed = sum(x[2] in fixed for x in ga)
    ga = [(pc, mn, fixed.get(ops, ops)) for pc, mn, ops in ga]
    k_mod = next((k for k, x in enumerate(ga) if x[2].endswith(",%xmm9") and x[1] != "movaps"), len(ga))
    lanes = {"$0xaa": "0x38(%rbx)", "$0x55": "0x34(%rbx)"}
    for k, (pc, mn, ops) in enumerate(ga):
        if mn == "shufps":
            imm8, src, dst = ops.split(",")
            assert src == "%xmm9" and k < k_mod and any(g[1:] == ("movaps", f"%xmm9,{dst}") for g in ga[k - 4:k]), (hex(pc), ops)
            ga[k] = (pc, "movss", f"{lanes[imm8]},{dst}")
    got = execute(ga, {"%rcx": "this", "%rdx": "t"}, {"%r8": "mn", "%r9": "mx"}, {}, {}, pe.bytes_at_va)
    rad, hh = E(("in", "this", 0x30 // 4)), E(("in", "this", 0x34 // 4))
    half = [rad, rad + hh, rad]
    t = lambda r_, c: E(("abs", ("in", "t", 4 * r_ + c)))
    ext = [(t(r_, 0) * half[0] + t(r_, 1) * half[1]) + t(r_, 2) * half[2] for r_ in range(3)]
    want = {}
    for r_ in range(3):
        want[("mn", r_)] = (E(("in", "t", 12 + r_)) - ext[r_]).t
        want[("mx", r_)] = (E(("in", "t", 12 + r_)) + ext[r_]).t
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    good = not bad and n_fixed == 3
    print(f"btCapsuleShape::getAabb (up axis 1): half = (r, r + halfHeight, r), no margin added, same extent sum as the box: "
          f"{6 - len(bad)} of 6 identical" + ("" if good else f"  <-- MISMATCH {bad}"))
    for k in bad[:2]:
        print("   compiled   :", norm(got.get(k, ("missing",))))
        print("   restatement:", norm(want[k]))
    ok &= good
    return ok


def check_update_single_aabb(pe):
    """btCollisionWorld::updateSingleAabb: the function that reads gContactBreakingThreshold (a .data float, 0.02f) and, within
    0x300 bytes after it, the 1e12f of its "AABB too large" test."""
    ok = True
    rd_va, _, rd_raw, rd_rs = pe.secs[1]
    big = [rd_va + m.start() for m in re.finditer(re.escape(struct.pack("<f", 1e12)), pe.b[rd_raw: rd_raw + rd_rs]) if m.start() % 4 == 0]
    big_refs = sorted(r for c in big for r in _rip_refs(pe, c))
    dsec = next(s for s in pe.secs if pe.b[s[2]: s[2] + s[3]].find(struct.pack("<f", 0.02)) != -1 and s not in pe.secs[:2])
    thr = [dsec[0] + m.start() for m in re.finditer(re.escape(struct.pack("<f", 0.02)), pe.b[dsec[2]: dsec[2] + dsec[3]]) if m.start() % 4 == 0]
    cand = sorted({_function_start(pe, r) for c in thr for r in _rip_refs(pe, c)
                   if any(0 < b - r < 0x300 for b in big_refs)})
    assert len(cand) == 1, [hex(c) for c in cand]
    fn_va = pe.base + cand[0]
    ins = _disasm(pe, fn_va, 0x300)
    text = [(x[1], x[2]) for x in ins]
    virt = [k for k, x in enumerate(ins) if x[1] == "call" and x[2] == "*0x8(%rax)"]     # shape->getAabb
    assert len(virt) == 2, virt
    # conditions of the second (interpolation transform) box, in order
    k0 = text.index(("cmpb", "$0x0,0x40(%rdi)"))                    # m_dispatchInfo.m_useContinuous
    k1 = text.index(("cmpl", "$0x2,0x118(%rbx)"))                   # getInternalType() == CO_RIGID_BODY
    k2 = text.index(("testb", "$0x3,0xe8(%rbx)"))                   # !isStaticOrKinematicObject()
    good = virt[0] < k0 < k1 < k2 < virt[1] and ins[k1 + 1][1] == "jne" and ins[k2 + 1][1] == "jne"
    good &= ("add", "$0x10,%rdx") in text[:virt[0]] and ("lea", "0x50(%rbx),%rdx") in text[k2:virt[1]]
    print("updateSingleAabb: getAabb(worldTransform [+0x10]); second box from the interpolation transform [+0x50] iff useContinuous "
          "&& CO_RIGID_BODY && !(flags & (STATIC|KINEMATIC)): " + ("as restated" if good else "MISMATCH"))
    ok &= good
    ins2 = [(pc, mn, "0xffff08") if (mn, ops) == ("call", "*0x8(%rax)") else (pc, mn, ops) for pc, mn, ops in ins]
    noop = {0xFFFF08: lambda reg, mem, sp, out: None}
    kje = next(k for k, x in enumerate(ins) if k > virt[0] and x[1] == "je")
    got = execute(ins2[:kje], {}, {"%rbp": "frame"}, {}, {}, pe.bytes_at_va, hooks=noop)
    thr_c = C(0.02)
    o = lambda d: E(("opaque", f"%rbp{d:#x}"))
    mins, maxs = (-0x19, -0x15, -0x11), (-0x29, -0x25, -0x21)
    want = {("frame", d // 4): (o(d) - thr_c).t for d in mins}
    want.update({("frame", d // 4): (o(d) + thr_c).t for d in maxs})
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    print(f"updateSingleAabb: min -= gContactBreakingThreshold (0.02f), max += it: {6 - len(bad)} of 6 identical"
          + ("" if not bad else f"  <-- MISMATCH {bad}"))
    ok &= not bad
    # the union: every comiss / jae / store triple is setMin on the min lanes and setMax on the max lanes
    kend = next(k for k, x in enumerate(ins) if k > virt[1] and x[1] == "testb")
    seg = ins2[k2 + 2: kend]
    cmps = [x for x in seg if x[1] == "comiss" and "(" not in x[2]]
    reg0 = {"%xmm6": thr_c.t}
    mem0 = {}
    got = execute(seg, {}, {"%rbp": "frame"}, reg0, mem0, pe.bytes_at_va, hooks=noop,
                  probes={x[0]: x[2].split(",") for x in cmps})
    mins2, maxs2 = (-0x9, -0x5, -0x1), (0x7, 0xb, 0xf)
    want = {("frame", a // 4): (o(b) - thr_c).t for a, b in zip(mins, mins2)}
    want.update({("frame", a // 4): (o(b) + thr_c).t for a, b in zip(maxs, maxs2)})
    bad = [k for k in want if norm(got.get(k, ("missing",))) != norm(want[k])]
    dirs = 0
    for x in cmps:
        a, b = x[2].split(",")
        ta, tb = norm(got[("probe", x[0], a)]), norm(got[("probe", x[0], b)])
        # comiss a,b ; jae skip : the store happens iff b < a.  setMin: b is the new (interpolated) minimum, a the current one;
        # setMax: a is the new maximum, b the current one.
        is_min = any(tb == norm((o(m) - thr_c).t) for m in mins2) and "opaque" in str(ta)
        is_max = any(ta == norm((o(m) + thr_c).t) for m in maxs2) and "opaque" in str(tb)
        dirs += is_min or is_max
    good = not bad and dirs == 6 and all(ins[k + 1][1] == "jae" or ins[k + 2][1] == "jae" or ins[k + 3][1] == "jae" or ins[k + 4][1] == "jae"
                                         for k, x in enumerate(ins) if x in cmps)
    print(f"updateSingleAabb: second box also -/+ 0.02f, then setMin / setMax lane by lane: {6 - len(bad)} of 6 stores, {dirs} of 6 comparisons"
          + ("" if good else "  <-- MISMATCH"))
    ok &= good
    kbig = next(k for k, x in enumerate(ins) if k > kend and x[1] == "comiss")
    lim = struct.unpack("<f", pe.bytes_at_va(ins[kbig + 1][0] + int(re.match(r"(-?0x[0-9a-f]+)\(%rip\)", ins[kbig][2]).group(1), 16), 4))[0]
    print(f"  (found at VA {fn_va:#x}; a non-static body whose box has squared diagonal >= {lim:g} is set to DISABLE_SIMULATION instead "
          "of being fed to the broadphase: not restated, the extent of such a body is 1e6 units)")
    return ok


def _find_integrate_transform(pe):
    """VA of btTransformUtil::integrateTransform: the only code referencing the float 1/48 (see check_integrate_transform)."""
    rd_va, _, rd_raw, rd_rs = pe.secs[1]
    i = pe.b.find(struct.pack("<I", 0x3CAAAAAB), rd_raw, rd_raw + rd_rs)
    refs = _rip_refs(pe, rd_va + i - rd_raw)
    assert len(refs) == 1, refs
    return pe.base + _function_start(pe, refs[0])


def _find_by_data_float(pe, value, also=None):
    """Functions that read a float of this value from an initialised-data section other than .rdata (Bullet's tunable
    globals: gDeactivationTime, gContactBreakingThreshold ...)."""
    out = set()
    for sec in pe.secs[2:]:
        blob = pe.b[sec[2]: sec[2] + sec[3]]
        for m in re.finditer(re.escape(struct.pack("<f", value)), blob):
            if m.start() % 4 == 0:
                out |= {pe.base + _function_start(pe, r) for r in _rip_refs(pe, sec[0] + m.start())}
    return sorted(out)


def check_step_order():
    """The ORDER of a simulation step.  btDiscreteDynamicsWorld's virtual functions are identified by what they reach:
      updateActivationState   the function reading gDeactivationTime (.data float 2.0) that walks m_nonStaticRigidBodies
      its vtable              the .rdata table holding that function's (thunk) address; base = nearest address below it that
                              code loads with lea (the constructor)
      internalSingleStepSimulation / stepSimulation
                              the vtable members that call updateActivationState's slot / internalSingleStepSimulation's slot
    and every virtual call they make through the same vtable is resolved and classified by the already-pinned functions it
    reaches within three calls: integrateTransform, updateSingleAabb, updateActivationState; applyGravity / clearForces by
    their accesses to m_gravity (+0x1f0) / m_totalForce (+0x220)."""
    from check_bx_order import EXE, Pe
    pe = Pe(EXE)
    ok = True
    integ = _find_integrate_transform(pe)
    cands = [f for f in _find_by_data_float(pe, 2.0) if any("0x174(" in x[2] for x in _disasm(pe, f, 0x80, multi_ret=True))]
    uas = [f for f in cands if sum(x[1] == "comiss" for x in _disasm(pe, f, 0x400)) >= 3]
    assert len(uas) == 1, [hex(c) for c in cands]
    uas = uas[0]
    # its vtable slot: the 8-byte address of the function or of its incremental-link thunk, inside .rdata
    va, _, raw, rs = pe.secs[0]
    text = pe.b[raw: raw + rs]
    thunks = [pe.base + va + m.start() for m in re.finditer(b"\xe9", text)
              if pe.base + va + m.start() + 5 + struct.unpack_from("<i", text, m.start() + 1)[0] == uas and m.start() + 5 <= len(text)]
    entries = []
    for t in [uas] + thunks:
        rd_va, _, rd_raw, rd_rs = pe.secs[1]
        entries += [pe.base + rd_va + m.start() for m in re.finditer(re.escape(struct.pack("<Q", t)), pe.b[rd_raw: rd_raw + rd_rs])
                    if m.start() % 8 == 0]
    assert len(entries) == 1, [hex(e) for e in entries]
    entry = entries[0]
    vtable = next(b for b in range(entry, entry - 0x400, -8) if _rip_refs(pe, b - pe.base))
    slot_uas = entry - vtable
    members = {}
    k = 0
    while True:                                                      # the table ends where the entries stop being code addresses
        ptr = struct.unpack("<Q", pe.bytes_at_va(vtable + 8 * k, 8))[0]
        if not (pe.base + va <= ptr < pe.base + va + rs):
            break
        members[8 * k] = _resolve(pe, ptr)
        k += 1

    def calls_of(f, size=0x500):
        """Calls of a function: ('d', target) direct, ('v', slot) through THIS object's vtable — `mov (%reg),%rax` with reg a
        copy of the entry %rcx, then `call *slot(%rax)` —, ('?', text) anything else (other objects' virtuals, callbacks)."""
        out = []
        this = {"%rcx"}
        vt_reg_ok = False
        for n_, x in enumerate(_disasm(pe, f, size)):
            parts = x[2].split(",")
            if x[1] == "mov" and len(parts) == 2:
                if parts[0] in this and re.fullmatch(r"%r\w+", parts[1]) and n_ < 24:
                    this.add(parts[1])
                elif parts[1] in this and parts[0] not in this:
                    this.discard(parts[1])                          # the copy is overwritten (rcx itself is an argument register)
                if parts[1] == "%rax":
                    m0 = re.fullmatch(r"\((%r\w+)\)", parts[0])
                    vt_reg_ok = bool(m0) and m0.group(1) in this
            if x[1] != "call":
                continue
            m = re.match(r"\*(0x[0-9a-f]+)\(%rax\)$", x[2])
            if m and vt_reg_ok and int(m.group(1), 16) in members:
                out.append(("v", int(m.group(1), 16)))
            elif x[2].startswith("0x"):
                out.append(("d", _resolve(pe, int(x[2], 16))))
            else:
                out.append(("?", x[2]))
            this.discard("%rcx")                                    # volatile across the call
        return out

    def reaches(f, goal, depth=3, seen=None):
        seen = seen if seen is not None else set()
        if f == goal:
            return True
        if depth == 0 or f in seen:
            return False
        seen.add(f)
        for kind, t in calls_of(f):
            t = members[t] if kind == "v" else t
            if kind != "?" and reaches(t, goal, depth - 1, seen):
                return True
        return False

    def touches(f, needle, depth=2):
        ins = _disasm(pe, f, 0x300)
        if any(needle in x[2] for x in ins):
            return True
        return depth > 0 and any(touches(t, needle, depth - 1) for kind, t in calls_of(f, 0x300) if kind == "d")

    usa = _find_by_data_float(pe, 0.02)
    rd_va, _, rd_raw, rd_rs = pe.secs[1]
    big = [rd_va + m.start() for m in re.finditer(re.escape(struct.pack("<f", 1e12)), pe.b[rd_raw: rd_raw + rd_rs]) if m.start() % 4 == 0]
    big_fns = {pe.base + _function_start(pe, r) for c in big for r in _rip_refs(pe, c)}
    usa = [f for f in usa if f in big_fns]
    assert len(usa) == 1, [hex(f) for f in usa]
    usa = usa[0]

    def classify(slot):
        f = members[slot]
        tags = []
        if f == uas:
            tags.append("updateActivationState")
        if reaches(f, integ):
            tags.append("integrateTransform")
        if reaches(f, usa):
            tags.append("updateSingleAabb")
        return tags

    internal = [f for s_, f in members.items() if ("v", slot_uas) in calls_of(f) and f != uas]
    assert len(internal) == 1, [hex(f) for f in internal]
    internal = internal[0]
    slot_internal = next(s_ for s_, f in members.items() if f == internal)
    seq = [(t, classify(t)) for kind, t in calls_of(internal) if kind == "v"]
    tagged = [(hex(t), tags) for t, tags in seq if tags]
    # (createPredictiveContacts also reaches integrateTransform: it predicts into a local for its CCD test, a no-op at the
    #  default ccdMotionThreshold of 0; it sits between the prediction and the collision detection)
    want = [["integrateTransform"], ["integrateTransform"], ["updateSingleAabb"], ["integrateTransform"], ["updateActivationState"]]
    good = [tags for _, tags in tagged] == want
    print(f"internalSingleStepSimulation (vtable {vtable:#x}, slot {slot_internal:#x}), virtual calls in order: "
          + ", ".join(f"{t:#x}" + (f" [{'+'.join(tags)}]" if tags else "") for t, tags in seq))
    print("  predictUnconstraintMotion (reaches integrateTransform) -> collision detection (reaches updateSingleAabb) -> ... -> "
          "integrateTransforms (reaches integrateTransform) -> updateActivationState: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # predictUnconstraintMotion: every body that is not static / kinematic (asleep or not): applyDamping, then
    # predictIntegratedTransform(timeStep, m_interpolationWorldTransform [+0x50])
    predict = members[seq[0][0]] if not tagged else members[int(tagged[0][0], 16)]
    ptext = [(x[1], x[2]) for x in _disasm(pe, predict, 0x200)]
    good = ("testb", "$0x3,0xe8(%rsi)") in ptext and ("lea", "0x50(%rsi),%r8") in ptext and not any("0xf8(" in o for _, o in ptext)
    print("predictUnconstraintMotion: for every body without STATIC|KINEMATIC flags, whatever its activation state: predicted "
          "pose into the interpolation transform (+0x50): " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # stepSimulation: ... applyGravity; loop { internalSingleStepSimulation; synchronizeMotionStates }; clearForces
    step = [f for s_, f in members.items() if ("v", slot_internal) in calls_of(f)]
    assert len(step) == 1, [hex(f) for f in step]
    sseq = [t for kind, t in calls_of(step[0]) if kind == "v"]
    k_int = sseq.index(slot_internal)
    grav = [t for t in sseq[:k_int] if touches(members[t], "0x1f0(%r") and touches(members[t], "0x220(%r")]
    gtext = [(x[1], x[2]) for t in grav for x in _disasm(pe, members[t], 0x100)]
    active_only = ("cmp", "$0x2,%edx") in gtext or ("cmp", "$0x2,%ecx") in gtext
    # clearForces is called through another register (the last virtual call of the function): zero stores to +0x220..+0x238
    last = [x for x in _disasm(pe, step[0], 0x500) if x[1] in ("call", "jmp") and x[2].startswith("*0x")][-1]
    cf = members[int(re.match(r"\*(0x[0-9a-f]+)\(", last[2]).group(1), 16)]
    ctext = [x[2] for x in _disasm(pe, cf, 0x100)]
    clears = all(any(o.endswith(f",{off:#x}(%rcx)") for o in ctext) for off in (0x220, 0x228, 0x230, 0x238))
    good = len(grav) == 1 and active_only and clears
    print(f"stepSimulation (slot {next(s_ for s_, f in members.items() if f == step[0]):#x}): applyGravity (slot {grav[0] if grav else 0:#x}; "
          "m_totalForce += m_gravity, only for bodies whose state is not ISLAND_SLEEPING / DISABLE_SIMULATION) before the "
          f"sub-steps, clearForces (slot {last[2]}) after them: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # btRigidBody::applyGravity (the callee of the world's applyGravity loop): m_totalForce += m_gravity * m_linearFactor
    # (m_linearFactor +0x1e0 is (1, 1, 1): nothing in the reference sets it), skipped for static / kinematic bodies
    if grav:
        callee = [t for kind, t in calls_of(members[grav[0]]) if kind == "d"]
        ag = _disasm(pe, callee[0], 0x100)
        got = execute(ag, {"%rcx": "rb"}, {"%rcx": "rb"}, {}, {}, pe.bytes_at_va)
        rb_ = lambda off: E(("in", "rb", off // 4))
        bad = [k for k in range(3) if norm(got.get(("rb", (0x220 + 4 * k) // 4), ("missing",))) !=
               norm((rb_(0x1E0 + 4 * k) * rb_(0x1F0 + 4 * k) + rb_(0x220 + 4 * k)).t)]
        good = not bad and (ag[0][1], ag[0][2]) == ("testb", "$0x3,0xe8(%rcx)")
        print(f"btRigidBody::applyGravity: totalForce = linearFactor * gravity + totalForce, not for STATIC|KINEMATIC: {3 - len(bad)} of 3 identical"
              + ("" if good else "  <-- MISMATCH"))
        ok &= good
    # btRigidBody::updateDeactivation + wantsSleeping, inlined into updateActivationState: the two speed tests and the timer as
    # expression trees (m_linearVelocity +0x1b0, m_angularVelocity +0x1c0, sleeping thresholds +0x25c / +0x260,
    # m_deactivationTime +0xfc), the state logic as text
    uins = _disasm(pe, uas, 0x400)
    utext = [(x[1], x[2]) for x in uins]
    cm = [k for k, x in enumerate(uins) if x[1] == "comiss"]
    k_first = next(k for k, x in enumerate(uins) if x[1] == "movss" and x[2].startswith("0x1b4("))
    body = uins[k_first: cm[1] + 1]
    got = execute(body, {"%rbx": "rb"}, "%none", {}, {}, pe.bytes_at_va,
                  probes={uins[cm[0]][0]: ["%xmm3", "%xmm2"], uins[cm[1]][0]: ["%xmm3", "%xmm2"]})
    rb = lambda off: E(("in", "rb", off // 4))
    len2 = lambda o: ((rb(o) * rb(o) + rb(o + 4) * rb(o + 4)) + rb(o + 8) * rb(o + 8)).t
    want = [(len2(0x1B0), (rb(0x25C) * rb(0x25C)).t), (len2(0x1C0), (rb(0x260) * rb(0x260)).t)]
    good = all(norm(got[("probe", uins[cm[k]][0], "%xmm3")]) == norm(want[k][0]) and
               norm(got[("probe", uins[cm[k]][0], "%xmm2")]) == norm(want[k][1]) for k in range(2))
    good &= uins[cm[0]][2] == "%xmm2,%xmm3" and uins[cm[0] + 1][1] == "jae" and uins[cm[1] + 1][1] == "jae"   # slow iff len2 < thr^2
    k_add = next(k for k, x in enumerate(uins) if k > cm[1] and x[1] == "addss")
    good &= utext[k_add - 1] == ("movaps", "%xmm6,%xmm0") and utext[k_add] == ("addss", "0xfc(%rbx),%xmm0") and \
        utext[k_add + 1] == ("movss", "%xmm0,0xfc(%rbx)")                                                  # time = timeStep + time
    # state logic: states 2 / 4 skip the timer; otherwise-not-slow: time = 0, setActivationState(0)
    good &= ("sub", "$0x2,%eax") in utext and ("test", "$0xfffffffd,%eax") in utext
    # wantsSleeping: state 4 never; gDisableDeactivation / gDeactivationTime == 0 never; states 2, 3 yes; else time > gDeactivationTime
    k_time = next(k for k, x in enumerate(uins) if x[1] == "comiss" and x[2].startswith("0xfc("))
    good &= utext[k_time + 1][0] == "jae" and ("cmp", "$0x4,%ecx") in utext and ("ucomiss", "%xmm7,%xmm0") in utext
    # sleeping: both velocities zeroed (two 16-byte stores of zero)
    good &= ("movups", "%xmm0,0x1c0(%rbx)") in utext and ("movups", "%xmm1,0x1b0(%rbx)") in utext
    print("updateActivationState: slow iff (vx^2 + vy^2) + vz^2 < thr * thr for both velocities (else time = 0), time = dt + time, "
          "sleeping wanted iff state 2 / 3 or gDeactivationTime < time, a sleeper's velocities zeroed: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    # btSimulationIslandManager::buildIslands — where a body that wants to sleep is put to sleep.  setActivationState is the
    # callee updateActivationState uses three times; buildIslands is its caller that passes both ISLAND_SLEEPING (2) and
    # WANTS_DEACTIVATION (3) and is reached from the solver step (solveConstraints -> buildAndProcessIslands -> buildIslands).
    sas = [t for kind, t in calls_of(uas) if kind == "d"]
    sas = max(set(sas), key=sas.count)
    stext = [(x[1], x[2]) for x in _disasm(pe, sas, 0x40)]
    good = stext == [("mov", "0xf8(%rcx),%eax"), ("sub", "$0x4,%eax"), ("cmp", "$0x1,%eax"), ("jbe", stext[3][1]),
                     ("mov", "%edx,0xf8(%rcx)"), ("ret", "")]
    print("btCollisionObject::setActivationState: unchanged if the state is DISABLE_DEACTIVATION / DISABLE_SIMULATION (4, 5), else "
          "stored: " + ("as restated" if good else "MISMATCH"))
    ok &= good
    callers = set()
    for m in re.finditer(b"\xe8", text):
        if m.start() + 5 <= len(text):
            t = pe.base + va + m.start() + 5 + struct.unpack_from("<i", text, m.start() + 1)[0]
            if pe.base + va <= t < pe.base + va + rs and _resolve(pe, t) == sas:
                callers.add(pe.base + _function_start(pe, va + m.start()))
    bi = []
    for f in sorted(callers):
        t_ = [(x[1], x[2]) for x in _disasm(pe, f, 0x500)]
        if ("mov", "$0x2,%edx") in t_ and ("mov", "$0x3,%edx") in t_ and any(o.endswith(",0xec(%rdx)") for _, o in t_):
            bi.append((f, t_))
    assert len(bi) == 1, [hex(f) for f, _ in bi]
    f, t_ = bi[0]
    k_state = t_.index(("mov", "0xf8(%rdx),%eax"))
    good = (t_[k_state - 2][0] == "cmp" and t_[k_state - 2][1].endswith(",0xec(%rdx)")          # getIslandTag() == islandId
            and t_[k_state + 1: k_state + 5] == [("cmp", "$0x1,%eax"), ("je", t_[k_state + 2][1]), ("cmp", "$0x4,%eax"),
                                                 ("je", t_[k_state + 2][1])]                       # ACTIVE_TAG or DISABLE_DEACTIVATION
            and ("cmpl", "$0x2,0xf8(%rsi)") in t_ and ("movl", "$0x0,0xfc(%rsi)") in t_)
    solver_slots = [t for t, tags in seq if not tags and reaches(members[t], f, 3)]
    good &= len(solver_slots) == 1 and seq.index((solver_slots[0], [])) > [tags for _, tags in seq].index(["updateSingleAabb"])
    print(f"buildIslands (VA {f:#x}, reached from slot {solver_slots[0] if solver_slots else 0:#x} = solveConstraints, after the collision "
          "detection and before integrateTransforms): an island none of whose bodies is ACTIVE_TAG / DISABLE_DEACTIVATION is set to "
          "ISLAND_SLEEPING; otherwise its sleeping bodies become WANTS_DEACTIVATION with their timer at 0: "
          + ("as restated" if good else "MISMATCH"))
    ok &= good
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
    # 4. btMatrix3x3::getRotation(q), the trace > 0 side (the other side, which indexes the matrix with run-time i, j, k, is
    #    check_get_rotation_other_side)
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
    # 4b. how the reference reads the angles back (src/physics/PhysicsSystem.cpp:937-947): NOT getEulerZYX of the body's basis
    #     but worldTransform.getRotation() -> btMatrix3x3(rotation) [= setRotation] -> getEulerZYX: the call order in the object
    _, _, rel = coff_section(OBJ, "?SyncRigidBodiesFromPhysics@PhysicsSystem@@")
    order = [rel[o].split("@")[0].lstrip("?") for o in sorted(rel) if "btMatrix3x3" in rel[o]]
    good = order == ["getRotation", "setRotation", "getEulerZYX"]
    print("SyncRigidBodiesFromPhysics: basis -> getRotation -> setRotation -> getEulerZYX (the angles come from the round-tripped "
          "matrix): " + ("as restated" if good else f"MISMATCH {order}"))
    ok &= good
    ok &= check_get_rotation_other_side()
    ok &= check_integrate_transform()
    ok &= check_external_force_impulse()
    ok &= check_box_aabb()
    ok &= check_step_order()
    print("RESULT:", "the restatement has the compiled code's operation order" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
