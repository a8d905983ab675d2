#!/usr/bin/env python3
"""Static check of the hand-issued global loads in the wave-specialised conv kernels (convgemm16w / 16q / 16h _kernel<*>) and the weight-
gradient kernels (wgrad16s_kernel, wgrad16s_pair_kernel, wgrad16t_kernel), and of the LDS-DMA instruction counts behind the waits of
convgemm16g_kernel (check_dma_kernel).

The loader waves issue `global_load_dwordx4` from inline asm and retire them with hand-counted `s_waitcnt vmcnt(N)`; the
compiler believes an asm output is valid right after the asm statement, so nothing but OUR waits keeps it from reading, copying
or overwriting a register whose data has not landed.  This script compiles the device code to ISA and walks every
convgemm16w kernel along its control-flow graph (every path, every distinct in-flight state):

  * an asm load puts its destination registers "in flight";
  * an asm `s_waitcnt vmcnt(N)` retires all but the newest N in-flight loads;
  * any compiler-emitted instruction that names an in-flight register is an error.

A second rule covers every hand-issued vector-memory instruction (loads and the epilogues' stores): its scalar base must not have been
written by `v_readfirstlane` fewer than five wait states earlier (sgpr_hazards below).

    python tools/check_asm_loads.py [--defines WG_X,WG_Y] [--keep out.s]
Exit code 0 = clean.  Used by tests/test_abi_cpu.py.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "constant-memory-waveglow_amd", "csrc", "wgflow.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def parse(lines):
    """-> (instructions, label -> index).  An instruction is (line_no, text, in_asm)."""
    ins, labels, inasm = [], {}, False
    for no, raw in enumerate(lines, 1):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t[0] in ";.":
            continue
        t = t.split(";")[0].strip()
        if t:
            ins.append((no, t, inasm))
    return ins, labels


def check_kernel(name, lines):
    """Walks the control-flow graph; the state is the ordered tuple of in-flight asm loads (their destination registers)."""
    ins, labels = parse(lines)
    errors, seen_err = [], set()
    nload = sum(1 for _, t, a in ins if a and t.startswith("global_load"))
    nwait = sum(1 for _, t, a in ins if a and t.startswith("s_waitcnt") and "vmcnt" in t)
    work, visited = [(0, ())], set()
    while work:
        pc, st = work.pop()
        while pc < len(ins):
            if (pc, st) in visited:
                break
            visited.add((pc, st))
            if len(visited) > 2_000_000:
                errors.append("%s: state space too large" % name)
                return nload, nwait, errors
            no, t, inasm = ins[pc]
            ops = [o for o in re.split(r"[ ,]+", t) if o]
            op = ops[0]
            if inasm and op.startswith("global_load"):
                dst = frozenset(regs(ops[1]))
                st = tuple(e for e in st if not (e[0] & dst)) + ((dst, no),)
            elif inasm and op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    st = st[len(st) - keep:] if keep else ()
            elif op == "s_endpgm":
                if st and ("end", no) not in seen_err:
                    seen_err.add(("end", no))
                    errors.append("%s:%d: wave ends with %d asm loads in flight" % (name, no, len(st)))
                break
            else:
                used = set()
                for o in ops[1:]:
                    used |= regs(o)
                for dst, at in st:
                    if used & dst and (no, at) not in seen_err:
                        seen_err.add((no, at))
                        errors.append("%s:%d: `%s` touches v%d.. loaded at line %d before its wait" % (name, no, t, min(dst), at))
                if op == "s_branch":
                    pc = labels[ops[1]]
                    continue
                if op.startswith("s_cbranch"):
                    work.append((labels[ops[1]], st))
            pc += 1
    return nload, nwait, errors


def sgpr_hazards(name, lines):
    """A hand-issued vector-memory instruction whose scalar base (s[a:b]) was written by v_readfirstlane fewer than five wait states
    earlier reads a STALE base: the hardware needs the wait states and the compiler's hazard recogniser does not look into asm
    statements (found the hard way in wf_rowsteps_kernel: memory access faults).  Straight-line check over the listing: the writer
    must be at least five wait states (instructions; `s_nop N` counts N + 1) in front of the use."""
    ins, _ = parse(lines)
    errors = []
    for n, (no, t, inasm) in enumerate(ins):
        m = re.match(r"global_(load|store)_\w+ .*?s\[(\d+):(\d+)\]", t)
        if not (inasm and m):
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        ws = 0
        for k in range(n - 1, max(-1, n - 8), -1):
            p = ins[k][1]
            d = re.match(r"v_readfirstlane_b32 s(\d+)", p)
            if d and lo <= int(d.group(1)) <= hi and ws < 5:
                errors.append("%s:%d: `%s` uses a base written by `%s` %d wait states earlier (needs 5)" % (name, no, t, p, ws))
                break
            mm = re.match(r"s_nop (\d+)", p)
            ws += int(mm.group(1)) + 1 if mm else 1
            if ws >= 5:
                break
    return errors


def check_dma_kernel(name, lines):
    """convgemm16g_kernel (csrc/wg_gemm16g.h): every wave issues SEVEN LDS-DMA instructions per chunk from inline asm and retires them with
    counted waits whose immediates assume exactly that: `s_waitcnt vmcnt(3)` at a chunk's barrier (all but the three B pieces of the
    youngest chunk), `vmcnt(10)` once behind the two chunks of the prologue, `vmcnt(0)` before the wave ends.  Walks every control-flow
    path with the number of LDS-DMA instructions in flight (at most) as the state -- a wait for vmcnt(N) leaves min(in flight, N) --
    and checks that a `vmcnt(3)` always finds 10 (the three B pieces of chunk c + 1 and the seven of chunk c + 2) and the `vmcnt(10)` 14, that every
    DMA's scalar operands follow the M0 recipe (s_mov_b32 m0, sN / s_nop 0 in front, inside the same statement), and that the wave never
    ends with a DMA not waited for (it would land in the next workgroup's LDS)."""
    ins, labels = parse(lines)
    errors, seen = [], set()
    ndma = sum(1 for _, t, a in ins if a and t.startswith("global_load_lds"))
    nwait = sum(1 for _, t, a in ins if a and t.startswith("s_waitcnt") and "vmcnt" in t)
    for n, (no, t, a) in enumerate(ins):
        if a and t.startswith("global_load_lds"):
            if n < 2 or not ins[n - 1][1].startswith("s_nop") or not re.match(r"s_mov_b32 m0, s\d+", ins[n - 2][1]):
                errors.append("%s:%d: LDS-DMA without `s_mov_b32 m0, sN; s_nop 0` in front" % (name, no))
    # the state: the wave's hand-issued vector-memory instructions that may still be in flight, oldest first, as runs of one kind
    # ('d' LDS-DMA, 'l' hand-issued register loads: the accumulate-into tile of convlayer16g_kernel's residual product); a wait for
    # vmcnt(N) keeps the youngest N (the compiler's own loads and the epilogues' stores only make a wait retire MORE: ignored)
    def trim(q, n):
        out, left = [], n
        for k, c in reversed(q):
            if left <= 0:
                break
            t = min(c, left)
            out.append((k, t))
            left -= t
        return tuple(reversed(out))

    def push(q, k):
        if q and q[-1][0] == k:
            return q[:-1] + ((k, min(q[-1][1] + 1, 99)),)
        return q + ((k, 1),)

    def dmas(q):
        return sum(c for k, c in q if k == "d")

    work, visited = [(0, ())], set()
    while work:
        pc, q = work.pop()
        while pc < len(ins):
            if (pc, q) in visited:
                break
            visited.add((pc, q))
            no, t, inasm = ins[pc]
            ops = [o for o in re.split(r"[ ,]+", t) if o]
            op = ops[0]
            if inasm and op.startswith("global_load_lds"):
                q = push(q, "d")
            elif inasm and op.startswith("global_load_dword"):
                q = push(q, "l")
            elif inasm and op == "s_waitcnt" and "vmcnt" in t:
                keep = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                cnt = dmas(q)
                # vmcnt(10): behind the prologue's two chunks, leaves B(0), A(1), B(1); vmcnt(3): at a chunk's barrier, 10 pieces in flight
                # (or 7 when a tile change has drained everything since); vmcnt(14) / vmcnt(48): for the sake of the 48 accumulate-into
                # loads (behind the residual product's 14 prologue pieces / right behind the gate product's last stores: nothing older stays)
                # vmcnt(8): the residual product of convlayer16g_kernel behind its first eight weight pieces (retires the gate product's stores,
                # trailing fetches and the accumulate-into loads)
                want = {10: (14,), 3: (10, 7), 14: (14,)}.get(keep)
                if keep == 8 and (dmas(trim(q, 8)) != 8 or len(trim(q, 8)) != 1) and (no, q) not in seen:
                    seen.add((no, q))
                    errors.append("%s:%d: `%s` does not leave exactly the eight weight pieces in flight" % (name, no, t))
                if want is not None and cnt not in want and (no, q) not in seen:
                    seen.add((no, q))
                    errors.append("%s:%d: `%s` with %d LDS-DMA instructions in flight (the immediate assumes %s)" % (name, no, t, cnt, want))
                if keep not in (0, 3, 8, 10, 14) and no not in seen:
                    seen.add(no)
                    errors.append("%s:%d: unexpected hand-written wait `%s`" % (name, no, t))
                q = trim(q, keep)
            elif op == "s_endpgm":
                if q and ("end", no) not in seen:
                    seen.add(("end", no))
                    errors.append("%s:%d: wave ends with hand-issued loads / LDS-DMA instructions in flight" % (name, no))
                break
            else:
                if op == "s_branch":
                    pc = labels[ops[1]]
                    continue
                if op.startswith("s_cbranch"):
                    work.append((labels[ops[1]], q))
            pc += 1
    return ndma, nwait, errors


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--defines", default="")
    ap.add_argument("--keep", default=None)
    a = ap.parse_args()
    out = a.keep or os.path.join(tempfile.mkdtemp(prefix="wgisa"), "wgflow.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", "-o", out, SRC]
    cmd += ["-D" + d for d in a.defines.split(",") if d]
    subprocess.run(cmd, check=True)
    text = open(out).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z\d+(convgemm16[wxqh]_kernel|convlayer16[hq]_kernel|wgrad16s_kernel|wgrad16s_pair_kernel|wgrad16t_kernel|wf_rowsteps_kernel)\w*:", l)]
    if not starts:
        print("no convgemm16w_kernel instantiation in the ISA")
        return 1
    bad = 0
    for s in [i for i, l in enumerate(text) if re.match(r"^_Z\d+(convgemm16g_kernel|convlayer16g_kernel)\w*:", l)]:
        e = next(i for i in range(s, len(text)) if ".amdhsa_kernel" in text[i] or text[i].startswith(".Lfunc_end"))
        kname = text[s].split(":")[0]
        ndma, nwait, errors = check_dma_kernel(kname, text[s:e])
        errors += sgpr_hazards(kname, text[s:e])
        print("%s: %d LDS-DMA instructions, %d counted waits, %d violations" % (kname, ndma, nwait, len(errors)))
        for m in errors[:10]:
            print("   ", m)
        bad += len(errors) + (1 if ndma == 0 else 0)
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".amdhsa_kernel" in text[i] or text[i].startswith(".Lfunc_end"))
        kname = text[s].split(":")[0]
        nload, nwait, errors = check_kernel(kname, text[s:e])
        errors += sgpr_hazards(kname, text[s:e])
        print("%s: %d asm loads, %d counted waits, %d violations" % (kname, nload, nwait, len(errors)))
        for m in errors[:10]:
            print("   ", m)
        bad += len(errors)
        if nload == 0:
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
