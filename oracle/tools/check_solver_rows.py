#!/usr/bin/env python3
"""Where the reference's committed build fuses multiply-adds, and which split-impulse row it runs.

TEST INFRASTRUCTURE (oracle/): reads /root/reference/build/bin/RelWithDebInfo/SandboxCity.exe as bytes through objdump;
nothing in it is loaded or run.  It pins four facts oracle/contact_ref.h and physics_ref.h rely on:

  1. contraction: every fused multiply-add of the exe (vfmadd* / vfnmadd* / vfmsub* / vfnmsub*) lies inside two functions,
     the _sse4_1_fma3 row solvers of btSequentialImpulseConstraintSolver, and the exe holds no other VEX-encoded float
     arithmetic.  Everything else (engine, glm, bx, the rest of Bullet) is unfused SSE: what -ffp-contract=off restates.
  2. the two fused functions are the ones ResolveRow follows: each has a `dpps $0x7f` dot product chain, one vfnmadd231ps
     (deltaImpulse -= deltaVelDotn * jacDiagABInv) per body and blendvps limit selection; the lower-limit one has one blend
     pair less than the generic one.
  3. the split-impulse row that bumps gNumSplitImpulseRecoveries with packed code (mulps / shufps / addps) is
     gResolveSplitPenetrationImpulse_sse2: its dot products are summed (z + y) + x, it has no FMA and no dpps, its lower
     limit select is cmpltps / andps / andnps / orps, and its velocity updates are mulps then addps.

  4. btDiscreteDynamicsWorld::stepSimulation's clock: m_localTime += timeStep; n = (int)(m_localTime / fixedTimeStep)
     (truncating); m_localTime -= (float)n * fixedTimeStep (mulss then subss); min(n, maxSubSteps) sub-steps are run and the
     unclamped n is returned — what physics_ref.h's `accumulate` restates.

Prints one line per fact and "RESULT: ..." at the end; exit code 1 on any mismatch.
"""
import os
import re
import subprocess
import sys

EXE = "/root/reference/build/bin/RelWithDebInfo/SandboxCity.exe"
FMA = re.compile(r"^v(fn?m(add|sub))\d{3}[ps][sd]$")
VEX_ARITH = re.compile(r"^v(add|sub|mul|div|sqrt|dp)(ss|ps|sd|pd)$")


def disassemble():
    text = subprocess.run(["objdump", "-d", "--no-show-raw-insn", EXE], capture_output=True, text=True, check=True).stdout
    ins = []
    for line in text.splitlines():
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if m:
            ins.append((int(m.group(1), 16), m.group(2), m.group(3).split("#")[0].strip()))
    return ins


def function_bounds(ins, index):
    """[first, last] instruction indices of the function around ins[index]: MSVC pads between functions with int3."""
    a = index
    while a > 0 and ins[a - 1][1] != "int3":
        a -= 1
    b = index
    while b + 1 < len(ins) and ins[b + 1][1] != "int3":
        b += 1
    return a, b


def main():
    if not os.path.exists(EXE):
        print("the reference build is not here; nothing checked")
        return 2
    ins = disassemble()
    ok = True

    # 1. contraction
    fma = [i for i, (_, mn, _) in enumerate(ins) if FMA.match(mn)]
    vex = [i for i, (_, mn, _) in enumerate(ins) if VEX_ARITH.match(mn)]
    funcs = sorted({function_bounds(ins, i) for i in fma})
    print(f"fused multiply-adds in the exe: {len(fma)}, in {len(funcs)} functions "
          f"({', '.join(hex(ins[a][0]) for a, _ in funcs)}); other VEX float arithmetic: {len(vex)}")
    ok &= len(fma) == 12 and len(funcs) == 2 and not vex

    # 2. the two fused row solvers
    shapes = []
    for a, b in funcs:
        body = ins[a:b + 1]
        mn = [m for _, m, _ in body]
        dpps = [o for _, m, o in body if m in ("dpps", "vdpps")]
        shapes.append((ins[a][0], mn.count("vfnmadd231ps"), sum(1 for m in mn if m.startswith("vfmadd")),
                       sum(1 for m in mn if m in ("blendvps", "vblendvps")), len(dpps), all(o.startswith("$0x7f") for o in dpps)))
    shapes.sort(key=lambda s: s[3])
    for va, nfn, nfm, nbl, ndp, imm in shapes:
        print(f"  row solver at {va:#x}: {ndp} dpps (all $0x7f: {'yes' if imm else 'NO'}), {nfn} vfnmadd231ps, {nfm} vfmadd*, {nbl} blendvps")
    lower, generic = shapes
    ok &= lower[1] == 2 and generic[1] == 2 and lower[2] == 4 and generic[2] == 4          # 2 dv terms, 4 velocity updates
    ok &= lower[4] == 4 and generic[4] == 4 and lower[5] and generic[5]                    # lin + ang dot per body
    ok &= lower[3] < generic[3]                                                            # one limit less to select
    print(f"  lower-limit row = {lower[0]:#x}, generic row = {generic[0]:#x}: "
          + ("as ResolveRow restates" if ok else "NOT as restated"))

    # 3. the split-impulse row in use
    cand = []
    for i, (pc, mn, ops) in enumerate(ins):
        if mn == "incl" and "(%rip)" in ops:
            a, b = function_bounds(ins, i)
            body = ins[a:b + 1]
            names = [m for _, m, _ in body]
            if len(body) < 200 and any("0x98(%r8)" in o for _, _, o in body) and "mulps" in names:
                cand.append((a, b))
    cand = sorted(set(cand))
    print(f"packed split-impulse rows (incl of a global counter, reads +0x98 of the row, mulps): {len(cand)}"
          + (f" at {ins[cand[0][0]][0]:#x}" if cand else ""))
    ok &= len(cand) == 1
    if cand:
        a, b = cand[0]
        body = ins[a:b + 1]
        names = [m for _, m, _ in body]
        no_fuse = not any(FMA.match(m) or m in ("dpps", "vdpps") for m in names)
        # a dot product: shufps $0xaa (z) and shufps $0x55 (y) are added first, the $0x0 splat (x) is added to that sum
        order_ok, dots = True, 0
        for k, (pc, mn, o) in enumerate(body):
            if mn == "shufps" and o.startswith("$0xaa,"):
                src, zreg = [s.strip() for s in o.split(",")[1:]]
                yreg = next((o2.split(",")[2].strip() for _, m2, o2 in body[max(0, k - 14):k + 15]
                             if m2 == "shufps" and o2.startswith("$0x55," + src + ",")), None)
                adds = [o2.split(",")[0].strip() for _, m2, o2 in body[k + 1:] if m2 == "addps" and o2.endswith("," + zreg)][:2]
                dots += 1
                order_ok &= adds == [yreg, src]          # (z + y) first, then + x (the $0x0 splat stays in the source register)
        select = all(m in names for m in ("cmpltps", "andps", "andnps", "orps"))
        sep = names.count("mulps") >= 8 and names.count("addps") >= 12
        print(f"  {dots} dot products, z and y summed before x: {'yes' if order_ok and dots == 4 else 'NO'}; no FMA / dpps: "
              f"{'yes' if no_fuse else 'NO'}; cmpltps/andps/andnps/orps select: {'yes' if select else 'NO'}; "
              f"velocity updates by mulps then addps: {'yes' if sep else 'NO'}")
        ok &= order_ok and dots == 4 and no_fuse and select and sep
    # 4. the sub-step clock of btDiscreteDynamicsWorld::stepSimulation (m_localTime at +0x1a0 of the world)
    hits = [i for i, (_, mn, ops) in enumerate(ins) if mn == "addss" and ops.startswith("0x1a0(%rcx),")]
    clocks = []
    for i in hits:
        a, b = function_bounds(ins, i)
        seq = [m for _, m, _ in ins[i:min(b, i + 16) + 1] if m in ("addss", "comiss", "jb", "divss", "cvttss2si", "cvtdq2ps", "mulss", "subss")]
        if seq == ["addss", "comiss", "jb", "divss", "cvttss2si", "cvtdq2ps", "mulss", "subss"]:
            clocks.append((a, b, i))
    print(f"stepSimulation's clock (localTime += timeStep; if (localTime >= fixed) {{ n = (int)(localTime / fixed); "
          f"localTime -= (float)n * fixed; }}): {len(clocks)} function" + (f" at {ins[clocks[0][0]][0]:#x}" if clocks else ""))
    ok &= len(clocks) == 1
    if clocks:
        a, b, i = clocks[0]
        body = ins[a:b + 1]
        names = [m for _, m, _ in body]
        clamp = "cmovg" in names                                                  # min(n, maxSubSteps) sub-steps are run
        ret_k = max(k for k, m in enumerate(names) if m == "ret")
        n_reg = next(o.split(",")[1] for _, m, o in body if m == "cvttss2si")      # the register n is truncated into
        returns_n = any(m == "mov" and o == f"{n_reg},%eax" for _, m, o in body[ret_k - 8:ret_k])
        one_sync = sum(1 for _, m, o in body if m == "call" and o == "*0xa0(%rax)") == 2   # in the loop / when n == 0
        print(f"  truncating conversion, unfused (float)n * fixed: yes; sub-steps clamped by cmovg: {'yes' if clamp else 'NO'}; "
              f"the UNclamped n is returned: {'yes' if returns_n else 'NO'}; synchronizeMotionStates per sub-step or once when n == 0: "
              f"{'yes' if one_sync else 'NO'}")
        ok &= clamp and returns_n and one_sync
    print("RESULT: " + ("the solver rows and the step clock are compiled as restated" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
