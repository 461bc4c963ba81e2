"""Build-time check of the one place where a kernel's correctness rests on what the register allocator did.

k_blur16's loader wave (csrc/k_blur.hip) keeps four tiles of row loads in flight in four register sets.  The loads and
the `s_waitcnt vmcnt(12)` that leaves the three younger tiles outstanding are inline assembly (left to the compiler every
wait is vmcnt(0), because its release / acquire atomics order global memory too), tied to the registers only by asm
constraints.  Two silent failures were met while that was written: a destination register handed to another
value while its load was still in flight, and a `"+v"` operand the allocator MOVED - copied before the wait, i.e. before the
data had arrived.  Nothing in the language forbids either, so the disassembly is checked after every build:

  (i)   the loader's inline-asm blocks come in the order  L L L L  (W12 L) x 4  W0 W0 W0 W0
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
        kinds = [it[1] for it in items if it[0] == "asm" and it[1] != "?"]
        want = ["L"] * 4 + ["W12", "L"] * 4 + ["W0"] * 4
        if kinds != want:
            problems.append(f"{name}: inline-asm blocks come as {' '.join(kinds)}, expected {' '.join(want)}")
            continue
        asm_idx = [k for k, it in enumerate(items) if it[0] == "asm" and it[1] != "?"]
        pro = [items[asm_idx[k]][2] for k in range(4)]
        loop_w = [asm_idx[4 + 2 * k] for k in range(4)]
        loop_l = [asm_idx[5 + 2 * k] for k in range(4)]
        for k in range(4):
            if items[loop_l[k]][2] != pro[k]:
                problems.append(f"{name}: load group {k} of the loop writes {sorted(items[loop_l[k]][2])}, the prologue's "
                                f"wrote {sorted(pro[k])}: the register set moved")
        sets = pro
        allregs = set().union(*sets)
        first, last = asm_idx[0], asm_idx[-1]
        for pos in range(first, last + 1):
            it = items[pos]
            if it[0] != "code":
                continue
            used = _regs(it[1]) & allregs
            if not used:
                continue
            owner = [k for k in range(4) if loop_w[k] < pos < loop_l[k]]
            if not owner:
                problems.append(f"{name}+{it[2]}: `{it[1]}` touches in-flight load registers {sorted(used)} outside any "
                                f"staging window (between a wait and the refill of its set)")
            elif not used <= sets[owner[0]]:
                problems.append(f"{name}+{it[2]}: `{it[1]}` in the staging window of set {owner[0]} touches registers "
                                f"{sorted(used - sets[owner[0]])} of another set")
    if not seen:
        problems.append("no k_blur16 instantiation found in the assembly")
    return problems
