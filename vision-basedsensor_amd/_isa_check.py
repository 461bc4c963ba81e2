"""Build-time check of the one place where a kernel's correctness rests on what the register allocator did.

k_blur16's loader wave (csrc/k_blur.hip) keeps four tiles of row loads in flight in four register sets.  The loads and
the `s_waitcnt vmcnt(12)` that leaves the three younger tiles outstanding are inline assembly (left to the compiler every
wait is vmcnt(0), because its release / acquire atomics order global memory too), tied to the registers only by asm
constraints.  Two silent failures were met while that was written: a destination register handed to another
value while its load was still in flight, and a `"+v"` operand the allocator MOVED - copied before the wait, i.e. before the
data had arrived.  Nothing in the language forbids either, so the disassembly is checked after every build:

  (i)   the loader's inline-asm blocks come in the order  L L L L  (W12 L) x 4  W0 W0 W0 W0, the loop's eight in cyclic
        order - a rotated loop has its last refill in front of the header -
        (L = four global_load_dwordx4, W12 / W0 = s_waitcnt vmcnt(12) / vmcnt(0));
  (ii)  load group k of the loop writes the same 16 registers as load group k of the prologue (the sets did not move);
  (iii) outside the asm blocks, a register of set k is mentioned ONLY between the loop's k-th wait and the load group
        that refills it (the staging: xor + ds_write) - no copy, no reuse, no spill anywhere else;
  (iv)  that staging code mentions no register of another set.

With (i)-(iv) the set staged behind a wait was loaded four groups earlier, so exactly 12 younger loads are outstanding
and vmcnt(12) covers it.  Any violation fails build() with the offending line."""
import re

_FN = re.compile(r"^(_Z8k_blur16\w+):")
_VR = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def _regs(text):
    out = set()
    for a, b, c in _VR.findall(text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def _functions(asm):
    name, body = None, []
    for line in asm.splitlines():
        m = _FN.match(line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body
                name = None
            else:
                body.append(line)


def check_blur16(asm: str):
    """-> list of problems (empty = fine) over every k_blur16 instantiation in the device assembly `asm`."""
    problems, seen = [], 0
    for name, body in _functions(asm):
        seen += 1
        # split into asm blocks and the code between them
        items, i = [], 0                                  # ("asm", kind, dest regs, line no) | ("code", line, line no)
        while i < len(body):
            if "#ASMSTART" in body[i]:
                j = i + 1
                blk = []
                while j < len(body) and "#ASMEND" not in body[j]:
                    blk.append(body[j].strip())
                    j += 1
                blk = [b for b in blk if b]
                if blk and all(b.startswith("global_load_dwordx4") for b in blk):
                    dest = set()
                    for b in blk:
                        dest |= _regs(b.split(",")[0])
                    items.append(("asm", "L", dest, i))
                    if len(blk) != 4 or len(dest) != 16:
                        problems.append(f"{name}: load group at +{i} has {len(blk)} loads into {len(dest)} registers")
                elif blk == ["s_waitcnt vmcnt(12)"]:
                    items.append(("asm", "W12", set(), i))
                elif blk == ["s_waitcnt vmcnt(0)"]:
                    items.append(("asm", "W0", set(), i))
                elif blk:
                    items.append(("asm", "?", set(), i))
                i = j + 1
                continue
            t = body[i].split(";")[0].strip()
            if t and not t.endswith(":") and not t.startswith("."):
                items.append(("code", t, i))
            i += 1
        A = [k for k, it in enumerate(items) if it[0] == "asm"]
        kinds = [items[k][1] for k in A]
        # prologue: four load groups; tail: four full waits; between them the loop's eight blocks, (W12 L) x 4 in CYCLIC
        # order (the compiler may rotate the loop: the last refill then sits in the latch block in front of the header)
        if len(kinds) != 16 or kinds[:4] != ["L"] * 4 or kinds[12:] != ["W0"] * 4 or \
                sorted(kinds[4:12]) != ["L"] * 4 + ["W12"] * 4:
            problems.append(f"{name}: inline-asm blocks come as {' '.join(kinds)}: expected L L L L, then (W12 L) x 4 in "
                            f"cyclic order, then W0 W0 W0 W0")
            continue
        loop = A[4:12]
        lk = kinds[4:12]
        if any(lk[i] == lk[(i + 1) % 8] for i in range(8)):
            problems.append(f"{name}: the loop's asm blocks do not alternate wait / load: {' '.join(lk)}")
            continue
        waits = [i for i in range(8) if lk[i] == "W12"]                 # in program order: tiles j = 0 .. 3
        refill = [loop[(i + 1) % 8] for i in waits]                     # the load group that follows each wait, cyclically
        pro = [items[A[k]][2] for k in range(4)]
        for k in range(4):
            if items[refill[k]][2] != pro[k]:
                problems.append(f"{name}: the load group behind wait {k} writes {sorted(items[refill[k]][2])}, the prologue's "
                                f"group {k} wrote {sorted(pro[k])}: the register set moved")
        allregs = set().union(*pro)
        lo, hi = A[3], A[12]                                             # region of the loop: between the prologue and the tail
        for pos in range(A[0], len(items)):
            it = items[pos]
            if it[0] != "code":
                continue
            used = _regs(it[1]) & allregs
            if not used:
                continue
            if pos < lo:                                 # prologue: a set is in flight from ITS load group on
                inflight = set().union(*[pro[k] for k in range(4) if A[k] < pos])
                if used & inflight:
                    problems.append(f"{name}+{it[2]}: `{it[1]}` touches load registers {sorted(used & inflight)} already in "
                                    f"flight in the prologue")
                continue
            if pos > hi:                                 # tail: everything has landed behind the first full wait
                continue
            before = [k for k in loop if k < pos]
            after = [k for k in loop if k > pos]
            if not before:
                # between the prologue and the loop's first asm block: the preheader (all four sets in flight) and, in a
                # rotated loop, the top of the latch block - nothing there has business with these registers
                problems.append(f"{name}+{it[2]}: `{it[1]}` touches load registers {sorted(used)} between the prologue and the "
                                f"loop's first wait / refill")
                continue
            prev_asm = before[-1]
            next_asm = after[0] if after else loop[0]                    # (cyclic: a rotated loop refills in its latch block)
            if items[prev_asm][1] != "W12" or items[next_asm][1] != "L":
                problems.append(f"{name}+{it[2]}: `{it[1]}` touches in-flight load registers {sorted(used)} outside a staging "
                                f"window (between a wait and the refill of its set)")
            elif not used <= items[next_asm][2]:
                problems.append(f"{name}+{it[2]}: `{it[1]}` in the staging window of registers {sorted(items[next_asm][2])[:1]}.. "
                                f"touches {sorted(used - items[next_asm][2])} of another set")
    if not seen:
        problems.append("no k_blur16 instantiation found in the assembly")
    return problems
