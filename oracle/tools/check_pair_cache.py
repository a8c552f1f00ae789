#!/usr/bin/env python3
"""Which pairs reach a trigger ghost's overlap list in the reference's COMPILED Bullet — read off the committed build.

TEST INFRASTRUCTURE (oracle/): reads /root/reference/build/SandboxCity.dir/RelWithDebInfo/PhysicsSystem.obj and
build/bin/RelWithDebInfo/SandboxCity.exe as bytes through objdump; nothing in them is loaded or run.  It pins what the
trigger rule of oracle/physics_ref.h (ProcessTriggerEvents) relies on — VERDICT r02 item 1:

  1. btGhostPairCallback::addOverlappingPair / removeOverlappingPair (inline members; compiled, with symbols, into
     PhysicsSystem.obj because PhysicsSystem.cpp:132 instantiates the class): each of the two proxies' client objects is a
     ghost iff its m_internalType (+0x118) == CO_GHOST_OBJECT (4); for EACH of the two that is one, the virtual
     addOverlappingObjectInternal (+0x30) / removeOverlappingObjectInternal (+0x38) is called with the other proxy.  No
     other test: nothing looks at CF_STATIC_OBJECT, activation state or the other object's type — a ghost lists Static
     bodies, and two ghosts list each other.
  2. btDbvtBroadphase's constructor (found through the call PhysicsSystem::InitializeWorld makes to it: the relocation in
     PhysicsSystem.obj names it, the bytes in front of the call are found again in the exe) stores m_deferedcollide = false,
     m_needcleanup = true (one 16-bit store 0x0100 at +0xdd) and creates a btHashedOverlappingPairCache when none is passed.
  3. btDbvtBroadphase::createProxy (vtable slot 1) and ::setAabb (slot 3): guarded by m_deferedcollide == 0 only, the new /
     moved leaf is collided against BOTH trees (two calls of one function, `this` = &m_sets[0] (+0x8) and &m_sets[1] (+0x48))
     — not "moving set against fixed set" alone; the fixed set holds proxies that did not move for a stage, not static objects.
     setAabb has no early-out for an unchanged box (DBVT_BP_PREVENTFALSEUPDATE off).
  4. btHashedOverlappingPairCache::addOverlappingPair (the cache's vtable slot 1): needsBroadphaseCollision (slot 9) decides,
     then internalAddPair.  needsBroadphaseCollision, instruction for instruction: the overlap filter callback (+0x28) if one
     is set (the reference sets none: no setOverlapFilterCallback under src/), else
         (proxy0->m_collisionFilterGroup [+0x8] & proxy1->m_collisionFilterMask [+0xc]) != 0 &&
         (proxy0->m_collisionFilterMask [+0xc] & proxy1->m_collisionFilterGroup [+0x8]) != 0
     and nothing else.  createProxy stores its group / mask arguments at exactly those offsets.
  5. internalAddPair: when it creates a NEW pair it calls m_ghostPairCallback (+0x70) ->addOverlappingPair (+0x8).

Prints one line per fact and "RESULT: ..." at the end; exit code 1 on any mismatch, 2 when the reference is absent.
"""
import os
import re
import struct
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

OBJ = "/root/reference/build/SandboxCity.dir/RelWithDebInfo/PhysicsSystem.obj"


def _obj_function(wanted):
    from check_bullet_order import coff_section
    _, code, rel = coff_section(OBJ, wanted)
    tmp = "/tmp/_bge_pc.bin"
    open(tmp, "wb").write(code)
    text = subprocess.run(["objdump", "-D", "-b", "binary", "-m", "i386:x86-64", "--no-show-raw-insn", tmp],
                          capture_output=True, text=True, check=True).stdout
    ins = []
    for line in text.splitlines():
        m = re.match(r"\s*([0-9a-f]+):\s+(\S+)\s*(.*)$", line)
        if m and m.group(2) != "int3":
            ins.append((int(m.group(1), 16), m.group(2), m.group(3).split("#")[0].strip()))
    return ins, code, rel


def check_ghost_pair_callback():
    ok = True
    for member, slot in (("?addOverlappingPair@btGhostPairCallback@@", "0x30"), ("?removeOverlappingPair@btGhostPairCallback@@", "0x38")):
        ins, _, rel = _obj_function(member)
        names = [(m, o) for _, m, o in ins]
        type_tests = [o for m, o in names if m == "cmpl" and o.startswith("$0x4,0x118(")]
        clears = sum(1 for m, _ in names if m == "cmovne")
        calls = [o for m, o in names if m == "call"]
        guards = sum(1 for m, _ in names if m == "test") == 2 and sum(1 for m, _ in names if m == "je") == 2
        loads = [o for m, o in names if m == "mov" and re.match(r"\(%(rdx|r8)\),%r", o)]                 # proxy->m_clientObject (+0)
        other = [m for m, _ in names if m.startswith("cmp") and m != "cmpl"] + [o for m, o in names if m in ("cmp", "cmpl", "testb", "cmpb") and "0x118(" not in o]
        good = (len(type_tests) == 2 and clears == 2 and calls == [f"*{slot}(%rax)"] * 2 and guards and len(loads) == 2 and not other and not rel)
        print(f"btGhostPairCallback::{member[1:member.index('@')]}: client objects of both proxies, m_internalType (+0x118) == 4 twice, "
              f"a virtual call +{slot} for each that is a ghost, no other test: {'yes' if good else 'NO'}")
        ok &= good
    return ok


def check_exe():
    from check_bullet_order import coff_section, _resolve, _disasm
    from check_bx_order import EXE, Pe
    pe = Pe(EXE)
    ok = True
    va, _, raw, rs = pe.secs[0]
    text = pe.b[raw: raw + rs]

    # 2. the broadphase's constructor, through InitializeWorld's call
    _, code, rel = coff_section(OBJ, "?InitializeWorld@PhysicsSystem@@")
    sites = sorted(off for off, name in rel.items() if name.startswith("??0btDbvtBroadphase@@"))
    assert len(sites) == 1, sites
    pat = code[sites[0] - 13: sites[0]]
    hits = [m.start() for m in re.finditer(re.escape(pat), text)]
    assert len(hits) == 1, hits
    at = hits[0] + 13
    ctor = _resolve(pe, pe.base + va + at + 4 + struct.unpack_from("<i", text, at)[0])
    ins = _disasm(pe, ctor, 0x400)
    this = next(o.split(",")[1] for _, m, o in ins if m == "mov" and o.startswith("%rcx,%r") and not o.endswith("(%rsp)"))
    vt_ins = next((pc, o) for pc, m, o in ins if m == "lea" and "(%rip)" in o)
    nxt = ins[[pc for pc, _, _ in ins].index(vt_ins[0]) + 1][0]
    vtable = nxt + int(vt_ins[1].split("(")[0], 16)
    flags = [o for _, m, o in ins if m == "movw" and o == f"$0x100,0xdd({this})"]
    release = any(m == "mov" and o == f"%al,0xdc({this})" for _, m, o in ins)
    calls = [int(o, 16) for _, m, o in ins if m == "call" and re.fullmatch(r"0x[0-9a-f]+", o)]
    print(f"btDbvtBroadphase::btDbvtBroadphase at {ctor:#x} (vtable {vtable:#x}): m_deferedcollide = 0 and m_needcleanup = 1 in one store "
          f"at +0xdd: {'yes' if flags else 'NO'}; m_releasepaircache (+0xdc) = (paircache == 0): {'yes' if release else 'NO'}")
    ok &= bool(flags) and release

    def slot(table, k):
        return _resolve(pe, struct.unpack_from("<Q", pe.b, pe.r2f(table - pe.base) + 8 * k)[0])

    # 3. createProxy / setAabb collide the leaf against both trees unless m_deferedcollide
    for name, k in (("createProxy", 1), ("setAabb", 3)):
        f = slot(vtable, k)
        body = _disasm(pe, f, 0x600, multi_ret=True)
        end = next(i for i, x in enumerate(body) if x[1] == "ret")
        body = body[:end + 1]
        this = next(o.split(",")[1] for _, m, o in body if m == "mov" and o.startswith("%rcx,%r") and not o.endswith("(%rsp)"))
        gi = [i for i, (_, m, o) in enumerate(body) if m == "cmpb" and o == f"$0x0,0xdd({this})"]
        good = len(gi) == 1
        trees = []
        if good:
            tail = body[gi[0]:]
            jne = next((o for _, m, o in tail[:4] if m == "jne"), None)
            good &= jne is not None
            pend = None
            for _, m, o in tail:
                if m == "lea" and o in (f"0x8({this}),%rcx", f"0x48({this}),%rcx"):
                    pend = o.split("(")[0]
                if m == "call" and pend:
                    trees.append((pend, o))
                    pend = None
            good &= sorted(t for t, _ in trees) == ["0x48", "0x8"] and len({c for _, c in trees}) == 1
        early = any(m in ("ucomiss", "cmpneqps", "cmpeqps") for _, m, _ in body[:12])        # (a NotEqual(aabb, leaf->volume) early-out would sit here)
        print(f"btDbvtBroadphase::{name} at {f:#x}: one guard (m_deferedcollide == 0), then the leaf against the tree at +0x8 AND the tree "
              f"at +0x48, one collide function: {'yes' if good else 'NO'}" + (f"; no unchanged-box early-out: {'yes' if not early else 'NO'}" if name == "setAabb" else ""))
        ok &= good and not early
        if name == "createProxy":
            st = [o for _, m, o in body if m == "mov" and re.fullmatch(r"%eax,0x[8c]\(%rbx\)", o)]
            good2 = st == ["%eax,0x8(%rbx)", "%eax,0xc(%rbx)"]
            print(f"  ... and stores its group / mask arguments at +0x8 / +0xc of the new proxy: {'yes' if good2 else 'NO'}")
            ok &= good2

    # 4. the hashed pair cache: constructor = the call the broadphase's constructor makes on the `paircache == 0` path
    cache_ctor = None
    for c in calls:
        f = _resolve(pe, c)
        head = _disasm(pe, f, 0x40)
        if any(m == "lea" and "(%rip)" in o for _, m, o in head[:8]) and any(m == "movb" and o == "$0x1,0x20(%rcx)" for _, m, o in head[:10]):
            cache_ctor = f
    assert cache_ctor, [hex(c) for c in calls]
    head = _disasm(pe, cache_ctor, 0x40)
    k = next(i for i, (_, m, o) in enumerate(head) if m == "lea" and "(%rip)" in o)
    cache_vt = head[k + 1][0] + int(head[k][2].split("(")[0], 16)
    add = slot(cache_vt, 1)
    body = _disasm(pe, add, 0x80, multi_ret=True)
    seq = [(m, o) for _, m, o in body]
    i_call = next(i for i, (m, o) in enumerate(seq) if m == "call")
    tail_jmp = next((o for m, o in seq[i_call:] if m == "jmp"), None)
    good = seq[i_call] == ("call", "*0x48(%rax)") and seq[i_call + 1] == ("test", "%al,%al") and tail_jmp is not None
    internal = _resolve(pe, int(tail_jmp, 16)) if good else 0
    print(f"btHashedOverlappingPairCache::addOverlappingPair at {add:#x} (cache vtable {cache_vt:#x}): needsBroadphaseCollision (slot 9) "
          f"decides, then internalAddPair at {internal:#x}: {'yes' if good else 'NO'}")
    ok &= good
    needs = slot(cache_vt, 9)
    body = [(m, o) for _, m, o in _disasm(pe, needs, 0x40, multi_ret=True)]
    body = body[: [i for i, (m, _) in enumerate(body) if m == "ret"][1] + 1]
    want = [("mov", "0x28(%rcx),%rcx"), ("test", "%rcx,%rcx"), ("je", None), ("mov", "(%rcx),%rax"), ("jmp", "*0x8(%rax)"),
            ("mov", "0x8(%rdx),%eax"), ("test", "%eax,0xc(%r8)"), ("je", None), ("mov", "0xc(%rdx),%eax"), ("test", "%eax,0x8(%r8)"), ("je", None),
            ("mov", "$0x1,%al"), ("ret", ""), ("xor", "%al,%al"), ("ret", "")]
    norm = [(m.replace("rex.W ", ""), o) for m, o in body]
    norm = [(("jmp", o.replace("jmp ", "").strip()) if m == "rex.W" else (m, o)) for m, o in norm]
    good = len(norm) == len(want) and all(m == wm and (wo is None or o == wo) for (m, o), (wm, wo) in zip(norm, want))
    print(f"btHashedOverlappingPairCache::needsBroadphaseCollision at {needs:#x}: the filter callback (+0x28) if set, else "
          f"(group0 & mask1) && (mask0 & group1) on +0x8 / +0xc of the proxies, and nothing else: {'yes' if good else 'NO'}")
    if not good:
        print("   ", norm)
    ok &= good

    # 5. internalAddPair hands a new pair to the ghost pair callback
    body = _disasm(pe, internal, 0x600, multi_ret=True) if internal else []
    this = next((o.split(",")[1] for _, m, o in body if m == "mov" and o.startswith("%rcx,%r")), None)
    good = False
    for i, (_, m, o) in enumerate(body):
        if m == "mov" and o == f"0x70({this}),%rcx":
            nxt = [(m2, o2) for _, m2, o2 in body[i + 1: i + 9]]
            good = ("test", "%rcx,%rcx") in nxt and ("call", "*0x8(%rax)") in nxt
    print(f"btHashedOverlappingPairCache::internalAddPair: a new pair goes to m_ghostPairCallback (+0x70) ->addOverlappingPair (+0x8): "
          f"{'yes' if good else 'NO'}")
    ok &= good
    src_uses = subprocess.run(["grep", "-rl", "setOverlapFilterCallback", "/root/reference/src"], capture_output=True, text=True).stdout.strip()
    print(f"the reference installs no overlap filter callback (grep of src/): {'yes' if not src_uses else 'NO: ' + src_uses}")
    ok &= not src_uses
    return ok


def main():
    from check_bx_order import EXE
    if not (os.path.exists(EXE) and os.path.exists(OBJ)):
        print("the reference build is not here; nothing checked")
        return 2
    ok = check_ghost_pair_callback()
    ok &= check_exe()
    print("RESULT: " + ("a ghost's list is the pair cache's: every proxy whose fed box overlaps and whose group / mask pass both ways, "
                        "Static bodies and other ghosts included" if ok else "MISMATCH"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
