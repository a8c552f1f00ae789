#!/usr/bin/env python3
"""dBoxBox2 (Bullet's btBoxBoxDetector.cpp) as COMPILED into the reference's exe against oracle/boxbox_ref.h — fact 2 and 3 of
check_contact_order.py (run through that script; TEST INFRASTRUCTURE, reads the exe as bytes, runs nothing).

The function has no symbol; it is the only code that reads the float 1.05 (fudge_factor).  Its separating-axis phase — 500
scalar SSE instructions between the prologue and the fifteenth "return 0" — is straight-line code once the conditional jumps are
left untaken (every test's `return 0` is a jump to one far exit; the `if (s2 > s)` blocks only set flags, codes and pointers), so
it is executed symbolically in one pass: every xmm register, every stack slot (through %rsp and through the frame pointer
%rbp = entry %rsp - 0x208) holds an expression tree over the function's inputs p1, R1, side1, p2, R2, side2.  At each
comparison that guards a `return 0` the compared value is s2 of that test; the fifteen trees are compared with the trees of
boxbox_ref.h's BoxBox2, normalised for the commutativity of + and x only (check_bx_order.norm).
"""
import os
import re
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _find_by_rdata_float(pe, bits):
    from check_bullet_order import _rip_refs, _function_start
    out = set()
    for sec in pe.secs:
        blob = pe.b[sec[2]: sec[2] + sec[3]]
        for m in re.finditer(re.escape(struct.pack("<I", bits)), blob):
            if m.start() % 4 == 0:
                out |= {pe.base + _function_start(pe, r) for r in _rip_refs(pe, sec[0] + m.start())}
    return sorted(out)


def _symbolic_sat(pe, ins, exit_target):
    """Straight-line symbolic execution; returns the list of (pc, tree) compared at the guards of `return 0`."""
    from check_bx_order import norm
    reg, mem, gpr = {}, {}, {}
    gpr["%rcx"] = ("p1", 0)
    gpr["%rdx"] = ("R1", 0)
    gpr["%r8"] = ("side1", 0)
    gpr["%r9"] = ("p2", 0)
    ptr_slots = {0x28: ("R2", 0), 0x30: ("side2", 0)}   # stack arguments, relative to %rsp at entry
    sp = 0
    rbp = None
    guards, scaled = [], []

    def const_at(va):
        raw = pe.b[pe.r2f(va - pe.base): pe.r2f(va - pe.base) + 16]
        if raw[:4] == b"\xff\xff\xff\x7f":
            return ("absmask",)
        if raw[:4] == b"\x00\x00\x00\x80":
            return ("signmask",)
        return ("const", struct.unpack("<f", raw[:4])[0])

    def split(ops):
        return [p.strip() for p in re.split(r",(?![^(]*\))", ops)] if ops else []

    def mem_key(op, pc_next):
        m = re.match(r"(-?0x[0-9a-f]+|)\((%\w+)\)", op)
        if not m:
            return None
        d = int(m.group(1), 16) if m.group(1) else 0
        r = m.group(2)
        if r == "%rip":
            return ("rip", pc_next + d)
        if r == "%rsp":
            return ("stack", sp + d)
        if r == "%rbp" and rbp is not None:
            return ("stack", rbp + d)
        if r in gpr and isinstance(gpr[r], tuple):
            return ("in", gpr[r][0], gpr[r][1] + d)
        return ("unknown", r, d)

    def load(op, pc_next):
        k = mem_key(op, pc_next)
        if k is None:
            return ("opaque", op)   # (indexed addressing: only in the contact-generation part, which is not executed symbolically)
        if k[0] == "rip":
            return const_at(k[1])
        if k[0] == "in":
            assert k[2] % 4 == 0
            return ("in", k[1], k[2] // 4)
        return mem.get(k, ("opaque", str(k)))

    for i, (pc, mn, ops) in enumerate(ins):
        pc_next = ins[i + 1][0] if i + 1 < len(ins) else pc + 8
        p = split(ops)
        if mn in ("push",) or (mn == "rex" and ops.startswith("push")):
            sp -= 8
        elif mn == "sub" and len(p) == 2 and p[1] == "%rsp":
            sp -= int(p[0][1:], 16)
        elif mn == "mov" and ops == "%rsp,%rax":
            gpr["%rax"] = ("frame", sp)
        elif mn == "lea" and len(p) == 2 and p[1] == "%rbp" and "(%rax)" in p[0] and gpr.get("%rax", (None,))[0] == "frame":
            rbp = gpr["%rax"][1] + int(p[0].split("(")[0], 16)
        elif mn == "lea" and len(p) == 2:
            k = mem_key(p[0], pc_next)
            gpr[p[1]] = (k[1], k[2]) if k and k[0] == "in" else None
        elif mn == "mov" and len(p) == 2 and p[0].startswith("%r") and p[1].startswith("%r") and "(" not in ops:
            gpr[p[1]] = gpr.get(p[0])
        elif mn == "mov" and len(p) == 2 and "(" in p[0] and p[1].startswith("%r"):
            k = mem_key(p[0], pc_next)
            if k and k[0] == "stack" and k[1] in ptr_slots:
                gpr[p[1]] = ptr_slots[k[1]]
            elif k and k[0] == "stack" and ("ptr", k[1]) in mem:
                gpr[p[1]] = mem[("ptr", k[1])]
            else:
                gpr[p[1]] = None
        elif mn == "mov" and len(p) == 2 and "(" in p[1] and p[0].startswith("%r"):
            k = mem_key(p[1], pc_next)
            if k and k[0] == "stack":
                mem[("ptr", k[1])] = gpr.get(p[0])
        elif mn in ("movss", "movaps", "movups"):
            src, dst = p
            if dst.startswith("%xmm"):
                reg[dst] = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            else:
                k = mem_key(dst, pc_next)
                if k and k[0] == "stack" and mn == "movss":
                    mem[k] = reg.get(src, ("opaque", src))
        elif mn in ("mulss", "addss", "subss", "divss"):
            src, dst = p
            b = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            reg[dst] = ({"mulss": "mul", "addss": "add", "subss": "sub", "divss": "div"}[mn], reg.get(dst, ("opaque", dst)), b)
        elif mn == "andps":
            src, dst = p
            m = reg.get(src) if src.startswith("%xmm") else load(src, pc_next)
            if m == ("absmask",):
                reg[dst] = ("abs", reg.get(dst, ("opaque", dst)))
            else:
                reg[dst] = ("opaque", f"andps@{pc:#x}")
        elif mn == "xorps":
            src, dst = p
            if src == dst:
                reg[dst] = ("const", 0.0)
            else:
                m = reg.get(src) if src.startswith("%xmm") else load(src, pc_next)
                reg[dst] = ("neg", reg.get(dst, ("opaque", dst))) if m == ("signmask",) else ("opaque", f"xorps@{pc:#x}")
        elif mn in ("maxss", "minss"):
            src, dst = p
            b = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
            reg[dst] = (mn[:3], reg.get(dst, ("opaque", dst)), b)
        elif mn == "sqrtss":
            src, dst = p
            reg[dst] = ("sqrtf", reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next))
        elif mn in ("comiss", "ucomiss"):
            src, dst = p
            nxt = [x for x in ins[i + 1: i + 4] if x[1] in ("ja", "jbe", "jb", "jae", "jp", "jne", "je")]
            if nxt and nxt[0][1] == "ja" and int(nxt[0][2], 16) == exit_target:
                bound = reg.get(src, ("opaque", src)) if src.startswith("%xmm") else load(src, pc_next)
                guards.append((pc, norm(reg.get(dst, ("opaque", dst))), bound))
            else:
                t = norm(reg.get(dst, ("opaque", dst)))
                if t[0] == "mul" and ("const", FUDGE) in (t[1], t[2]):
                    scaled.append((pc, t))            # `if (s2 * fudge_factor > s)` of an edge axis
        # everything else (integer code, flags, conditional moves, jumps left untaken): not modelled
    return guards, scaled


FUDGE = struct.unpack("<f", struct.pack("<I", 0x3F866666))[0]


def check(all_ins):
    from check_bullet_order import _disasm
    from check_bx_order import EXE, Pe, norm, E
    pe = Pe(EXE)
    ok = True
    fns = _find_by_rdata_float(pe, 0x3F866666)  # 1.05f
    print(f"functions reading the float 1.05 (dBoxBox2's fudge_factor): {len(fns)}" + (f" at {fns[0]:#x}" if fns else ""))
    if len(fns) != 1:
        return False
    fn = fns[0]
    ins = _disasm(pe, fn, 0x2600, multi_ret=True)
    end = max(i for i, x in enumerate(ins) if x[1] == "ret")
    ins = ins[: end + 1]
    names = [m for _, m, _ in ins]
    sqrt = names.count("sqrtss")
    fudge2 = fn in _find_by_rdata_float(pe, 0x3727C5AC)  # 1e-5f
    direct = sorted({o for _, m, o in ins if m == "call" and re.fullmatch(r"0x[0-9a-f]+", o)})
    virt = sum(1 for _, m, o in ins if m == "call" and o == "*0x18(%rax)")
    good = sqrt == 9 and fudge2 and virt == 3
    print(f"  {len(ins)} instructions, {sqrt} sqrtss (nine edge-edge axes), reads 1e-5 (fudge2): {'yes' if fudge2 else 'NO'}, {virt} virtual calls "
          f"+0x18 (Result::addContactPoint: edge-edge, two face loops), direct calls {[hex(int(d, 16)) for d in direct]}: "
          + ("as the published dBoxBox2" if good else "NOT as expected"))
    ok &= good

    # the far exit every `return 0` of the separating-axis phase jumps to
    targets = {}
    for _, m, o in ins:
        if m == "ja" and re.fullmatch(r"0x[0-9a-f]+", o):
            targets[int(o, 16)] = targets.get(int(o, 16), 0) + 1
    exit_target = max(targets, key=targets.get)
    guards, scaled = _symbolic_sat(pe, ins, exit_target)

    # ---- the restatement's trees (oracle/boxbox_ref.h BoxBox2), over the same inputs
    def I(name, k):
        return E(("in", name, k))

    def C(x):
        return E(("const", struct.unpack("<f", struct.pack("<f", x))[0]))

    def A_(x):
        return E(("abs", x.t))

    p = [I("p2", j) - I("p1", j) for j in range(3)]
    R1 = [I("R1", k) for k in range(12)]
    R2 = [I("R2", k) for k in range(12)]
    pp = [R1[i] * p[0] + R1[4 + i] * p[1] + R1[8 + i] * p[2] for i in range(3)]
    A = [I("side1", i) * C(0.5) for i in range(3)]
    B = [I("side2", i) * C(0.5) for i in range(3)]
    R = [[R1[i] * R2[j] + R1[4 + i] * R2[4 + j] + R1[8 + i] * R2[8 + j] for j in range(3)] for i in range(3)]
    Q = [[A_(R[i][j]) for j in range(3)] for i in range(3)]
    want = []
    for i in range(3):
        want.append(A_(pp[i]) - ((A[i] + B[0] * Q[i][0]) + (B[1] * Q[i][1] + B[2] * Q[i][2])))
    for j in range(3):
        d = R2[j] * p[0] + R2[4 + j] * p[1] + R2[8 + j] * p[2]
        want.append(A_(d) - ((A[0] * Q[0][j] + A[1] * Q[1][j]) + (A[2] * Q[2][j] + B[j])))
    F = [[Q[i][j] + C(1.0e-5) for j in range(3)] for i in range(3)]
    r, q = R, F
    edge = [
        (pp[2] * r[1][0] - pp[1] * r[2][0], (A[1] * q[2][0] + A[2] * q[1][0]) + (B[1] * q[0][2] + B[2] * q[0][1])),
        (pp[2] * r[1][1] - pp[1] * r[2][1], (A[1] * q[2][1] + A[2] * q[1][1]) + (B[0] * q[0][2] + B[2] * q[0][0])),
        (pp[2] * r[1][2] - pp[1] * r[2][2], (A[1] * q[2][2] + A[2] * q[1][2]) + (B[0] * q[0][1] + B[1] * q[0][0])),
        (pp[0] * r[2][0] - pp[2] * r[0][0], (A[0] * q[2][0] + A[2] * q[0][0]) + (B[1] * q[1][2] + B[2] * q[1][1])),
        (pp[0] * r[2][1] - pp[2] * r[0][1], (A[0] * q[2][1] + A[2] * q[0][1]) + (B[0] * q[1][2] + B[2] * q[1][0])),
        (pp[0] * r[2][2] - pp[2] * r[0][2], (A[0] * q[2][2] + A[2] * q[0][2]) + (B[0] * q[1][1] + B[1] * q[1][0])),
        (pp[1] * r[0][0] - pp[0] * r[1][0], (A[0] * q[1][0] + A[1] * q[0][0]) + (B[1] * q[2][2] + B[2] * q[2][1])),
        (pp[1] * r[0][1] - pp[0] * r[1][1], (A[0] * q[1][1] + A[1] * q[0][1]) + (B[0] * q[2][2] + B[2] * q[2][0])),
        (pp[1] * r[0][2] - pp[0] * r[1][2], (A[0] * q[1][2] + A[1] * q[0][2]) + (B[0] * q[2][1] + B[1] * q[2][0])),
    ]
    nvec = [(None, r[2][0], r[1][0]), (None, r[2][1], r[1][1]), (None, r[2][2], r[1][2]), (r[2][0], None, r[0][0]), (r[2][1], None, r[0][1]),
            (r[2][2], None, r[0][2]), (r[1][0], r[0][0], None), (r[1][1], r[0][1], None), (r[1][2], r[0][2], None)]
    want_scaled = []
    for (e1, e2), nv in zip(edge, nvec):
        want.append(A_(e1) - e2)
        a, b = [x for x in nv if x is not None]
        l = E(("sqrtf", (a * a + b * b).t))              # (the zero component's square drops out exactly)
        want_scaled.append(((A_(e1) - e2) * E(("div", C(1.0).t, l.t))) * C(1.05))
    same = 0
    bounds_ok = True
    for k, w in enumerate(want):
        if k < len(guards):
            pc, got, bound = guards[k]
            if got == norm(w.t):
                same += 1
            else:
                print(f"  axis {k + 1} (compare at {pc:#x}) differs\n    compiled   : {got}\n    restatement: {norm(w.t)}")
            exp = ("const", 0.0) if k < 6 else ("const", struct.unpack("<f", struct.pack("<I", 0x34000000))[0])  # 0 / SIMD_EPSILON
            bounds_ok &= bound == exp
    good = len(guards) == 15 and same == 15 and bounds_ok
    print(f"  separating-axis phase, executed symbolically: {len(guards)} guarded `return 0`, s2 of {same} of 15 axes identical as expression "
          f"trees — four-term sums associated (t0 + t1) + (t2 + t3), MSVC /fp:fast; faces compared with 0, edge axes with SIMD_EPSILON: "
          f"{'yes' if bounds_ok else 'NO'}" + ("" if good else "  <-- MISMATCH"))
    ok &= good
    got_scaled = [t for _, t in scaled]
    same2 = sum(1 for w in want_scaled if norm(w.t) in got_scaled)
    good = same2 == 9
    print(f"  edge axes: s2 * (1 / l) * 1.05 with ONE reciprocal per axis (not s2 / l): {same2} of 9 identical as expression trees"
          + ("" if good else "  <-- MISMATCH"))
    if not good:
        for pc, t in scaled[:2]:
            print("    compiled:", hex(pc), t)
        print("    restated:", norm(want_scaled[0].t))
    ok &= good

    # ---- cullPoints2: the direct callee that calls atan2f in a loop
    cull = None
    for d in direct:
        from check_bullet_order import _resolve
        f = _resolve(pe, int(d, 16))
        body = _disasm(pe, f, 0x600, multi_ret=True)
        e = max(i for i, x in enumerate(body) if x[1] == "ret")
        body = body[: e + 1]
        consts = set()
        for i, (pc, m, o) in enumerate(body):
            mm = re.search(r"(-?0x[0-9a-f]+)\(%rip\)", o)
            if mm and i + 1 < len(body):
                va = body[i + 1][0] + int(mm.group(1), 16)
                try:
                    consts.add(struct.unpack_from("<I", pe.b, pe.r2f(va - pe.base))[0])
                except Exception:
                    pass
        if 0x40490FDB in consts:  # 3.14159265f
            cull = (f, consts, body)
    good = cull is not None and 0x40C90FDB in cull[1] and 0x3EAAAAAB in cull[1] and 0x4E6E6B28 in cull[1] and 0x40400000 not in cull[1]
    if cull:
        # 1.f / (3 * (a + q)) is compiled as 0x3eaaaaab / (a + q): the constant is loaded right before a divss
        k = next((i for i, (pc, m, o) in enumerate(cull[2]) if m == "movss" and "(%rip)" in o and
                  struct.unpack_from("<I", pe.b, pe.r2f(cull[2][i + 1][0] + int(o.split("(")[0], 16) - pe.base))[0] == 0x3EAAAAAB), None)
        good &= k is not None and cull[2][k + 1][1] == "divss"
    if cull:
        calls = [o for _, m, o in cull[2] if m == "call"]
        good &= len(set(calls)) >= 1
    print(f"  cullPoints2" + (f" at {cull[0]:#x}" if cull else "") + ": reads 3.14159265 (M__PI), 6.2831853 and 1e9, calls atan2f, and forms 1 / (3 x) as 0x3eaaaaab / x: "
          + ("yes" if good else "NO"))
    ok &= good
    return ok


if __name__ == "__main__":
    sys.exit(0 if check(None) else 1)
