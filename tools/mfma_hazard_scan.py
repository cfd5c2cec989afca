"""MFMA result read too early across a taken branch (gfx950): a scan of the built library's disassembly.

On CDNA the matrix pipe has no interlock for a VALU / LDS / memory instruction that reads a register an MFMA is still writing:
the compiler pads the distance with independent instructions or s_nop.  ROCm 7.2's hazard recognizer counts that distance in
LAYOUT order.  Where a conditional branch right after an MFMA jumps forward over a block, layout order sees the skipped block's
instructions, the taken path does not, and the first read at the branch target gets the accumulator before the last pass has
landed (registers of the last pass — [3] of a 16x16 tile — keep the value of the previous k step).  Round 4 found this in
attn_qblock_kernel (csrc/attention.hip): the score gradient of the last 16-key tile was wrong in ~25 % of launches at
T = 128 / 160 / 192.

  python tools/mfma_hazard_scan.py [libglowtts_hip.so]        exit status 1 when a hazard is found

For every MFMA the scan walks the control-flow graph forward (each instruction one wait state, s_nop n = n + 1) until the
result has been out for `need` wait states, and reports any instruction on the way that reads a register of the MFMA's
destination (an MFMA that takes it as its accumulator input is exempt: the matrix pipe forwards that).  `need` per opcode is
what the compiler itself keeps where it does count right — straight-line code and loop back-edges (it pads those with s_nop to
exactly its requirement): the smallest such distance found in the library for that opcode (passes + 3 from the ISA tables for
an opcode with no sample).  Only paths through a taken FORWARD branch are reported."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_FUNC = re.compile(r"^([0-9a-f]{8,16}) <([^>]+)>:")
_INS = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]{8,16}):")
_TGT = re.compile(r"<[^>]*?\+0x([0-9a-f]+)>\s*$")
_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(op):
    out = set()
    for m in _REG.finditer(op):
        if m.group(4) is not None:
            out.add((m.group(1), int(m.group(4))))
        else:
            out.update((m.group(1), n) for n in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def floor_need(mnem):
    """ISA floor: passes + 3 (gfx950: XDL write -> VALU / memory read = passes + 2 + 1)."""
    if "32x32" in mnem:
        return 16 + 3 if re.search(r"32x32x(1|2|4|16)(_|$|f|b)", mnem) else 8 + 3
    if "4x4" in mnem:
        return 2 + 3
    return 8 + 3 if re.search(r"16x16x(1|4|32|64|128)(_|$|f|b)", mnem) else 4 + 3


_ALL_READ = ("ds_write", "ds_store", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic",
             "flat_atomic", "ds_add", "ds_max", "ds_min", "s_", "v_cmp", "v_cmpx", "exp")


def reads_of(mnem, ops):
    parts = [p.strip() for p in ops.split(",")] if ops else []
    if not parts:
        return set()
    if mnem.startswith(_ALL_READ):
        return regs(ops)
    return regs(",".join(parts[1:]))            # first operand is the destination


def writes_of(mnem, ops):
    parts = [p.strip() for p in ops.split(",")] if ops else []
    if not parts or mnem.startswith(_ALL_READ):
        return set()
    return regs(parts[0])


def scan_function(name, start, ins, stats, found, limit=24):
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    n = len(ins)

    def succ(i):
        a, m, o, tgt = ins[i]
        if m in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap"):
            return []
        nxt = [(i + 1, 0)] if i + 1 < n else []
        kind = 1 if (tgt is not None and tgt <= a) else 2          # 1 = backward (a loop's edge), 2 = forward
        if m == "s_branch":
            return [(index[tgt], kind)] if tgt in index else []
        if m.startswith("s_cbranch") and tgt in index:
            return nxt + [(index[tgt], kind)]
        return nxt

    for i, (a, m, o, _) in enumerate(ins):
        if not (m.startswith("v_mfma") or m.startswith("v_smfmac")):
            continue
        dst = regs(o.split(",")[0])
        need = stats["need"].get(m, floor_need(m))
        # walk: (instruction, wait states elapsed when it issues, 0 = straight line / 1 = loop edges only / 2 = a forward branch taken)
        seen = {}
        todo = [(j, 0, t) for j, t in succ(i)]
        while todo:
            j, w, taken = todo.pop()
            if w >= limit or seen.get((j, taken), 99) <= w:
                continue
            seen[(j, taken)] = w
            _, mj, oj, _ = ins[j]
            is_mfma = mj.startswith("v_mfma") or mj.startswith("v_smfmac")
            rd = reads_of(mj, oj)
            if is_mfma:                           # accumulator input (last register operand) is forwarded by the pipe
                parts = [p.strip() for p in oj.split(",")]
                rd = regs(",".join(parts[1:3]))
            if rd & dst:
                if taken < 2:
                    stats["linear"][m] = min(stats["linear"].get(m, 99), w)
                elif w < need:
                    found.append((name, a, m, o.split(",")[0], ins[j][0], mj, w, need))
                continue                          # the first read on this path settles it
            if writes_of(mj, oj) >= dst and not is_mfma:
                continue
            step = 1
            if mj == "s_nop":
                step = int(oj.strip() or 0) + 1
            for k, t in succ(j):
                todo.append((k, w + step, max(taken, t)))


def disassemble(lib, tmp):
    copy = os.path.join(tmp, "lib.so")
    shutil.copy(lib, copy)
    subprocess.run([OBJDUMP, "--offloading", copy], check=True, capture_output=True, cwd=tmp)
    for f in sorted(os.listdir(tmp)):
        if "gfx950" in f:
            p = subprocess.Popen([OBJDUMP, "-d", os.path.join(tmp, f)], stdout=subprocess.PIPE, text=True)
            yield from p.stdout
            p.wait()


def functions(lines):
    name, start, ins = None, 0, []
    for line in lines:
        f = _FUNC.match(line)
        if f:
            if name and ins:
                yield name, start, ins
            name, start, ins = f.group(2), int(f.group(1), 16), []
            continue
        m = _INS.match(line)
        if m and name:
            tgt = None
            if m.group(1).startswith(("s_cbranch", "s_branch")):
                t = _TGT.search(line)
                if t:
                    tgt = start + int(t.group(1), 16)
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), tgt))
    if name and ins:
        yield name, start, ins


def scan(lib):
    """Two passes over the disassembly: straight-line distances first (what the compiler keeps), then the branch paths."""
    with tempfile.TemporaryDirectory() as tmp:
        funcs = [f for f in functions(disassemble(lib, tmp)) if any(i[1].startswith(("v_mfma", "v_smfmac")) for i in f[2])]
    stats = {"linear": {}, "need": {}}
    for name, start, ins in funcs:
        scan_function(name, start, ins, stats, [])
    stats["need"] = dict(stats["linear"])
    found = []
    for name, start, ins in funcs:
        scan_function(name, start, ins, stats, found)
    return found, stats, len(funcs)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "glow-tts-train_amd", "lib", "libglowtts_hip.so")
    found, stats, nf = scan(lib)
    print(f"{nf} kernels with MFMAs; smallest distance MFMA -> first read of its result in straight-line code and around loops, per opcode:")
    for m, w in sorted(stats["linear"].items()):
        print(f"  {m:36s} {w:3d} wait states (ISA floor {floor_need(m)})")
    by_kernel = collections.Counter(f[0] for f in found)
    for k, c in sorted(by_kernel.items()):
        print(f"  {c:4d}  {k[:150]}")
    for f in found[:int(os.environ.get("SHOW", "12"))]:
        print("HAZARD  %s\n        %x: %s -> %s   read at %x by %s after %d wait states on a path through a taken forward branch (elsewhere the compiler keeps %d)"
              % (f[0][:110], f[1], f[2], f[3], f[4], f[5], f[6], f[7]))
    print(f"{len(found)} early reads behind taken forward branches in {len(by_kernel)} kernels")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main())
