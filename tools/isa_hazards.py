"""Scan the gfx950 ISA of the sweep kernels (make -C epic_amd/csrc asm -> build/asm/*.s) for instruction sequences that are
known to misbehave on the hardware and that the compiler's hazard recognizer does not cover for this code:

  store-data   a VALU instruction (or LDS / VMEM load return register) that WRITES a data register of a 16-byte
               buffer store within `--store-slots` wait states behind the store.  Measured on MI355X: v_pk_fma_f32 directly
               behind buffer_store_dwordx4 ... offen with an SGPR soffset -> lanes 12..15 of every 16 of the second data
               register reach memory with the new value.  (LLVM pads this hazard only for stores without an SGPR soffset.)
  exec-dpp     a DPP instruction within 5 wait states behind a write to EXEC (s_mov_b64 exec / s_and_saveexec / ...): the
               DPP may still see the old mask.  cell_update.h narrows EXEC inside inline assembly, which the compiler
               cannot see.
  scratch      any scratch_ / buffer_*_lds-free spill traffic in a sweep kernel (a spill in the row loop costs more than
               any arithmetic it saves).

Wait states are counted as the hardware does for these hazards: one per instruction issued in between, plus N + 1 for an
`s_nop N`.  Exit code 1 and one line per finding when something is found.

    python tools/isa_hazards.py epic_amd/csrc/build/asm/kernels_2d-hip-amdgcn-amd-amdhsa-gfx950.s [...]
"""
import argparse
import re
import sys

KERNEL_RE = re.compile(r"^(_Z\w*(sweep2d_kernel|sweep3d_kernel|sweep3d_pair_kernel|rb_fused2d_kernel|rb_tol_fused2d_kernel|jacobi_fused2d_kernel)\w*):")


def regs(tok):
    tok = tok.rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def wait_states(ins):
    t = ins.split()
    if t[0] == "s_nop":
        return int(t[1], 0) + 1
    return 1


def kernels(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = KERNEL_RE.match(lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].strip().startswith("s_endpgm"):
            l = lines[j].strip()
            if l and not l.startswith((";", ".", "//")) and not re.match(r"^\.?\w+:", l):
                body.append(l.split(";")[0].strip())
            j += 1
        yield m.group(1), body
        i = j


def scan(name, body, store_slots):
    out = []
    for i, ins in enumerate(body):
        t = ins.split()
        op = t[0]
        if op.startswith("buffer_store_dwordx4") or op.startswith("buffer_store_dwordx3"):
            data = regs(t[1])
            ws = 0
            for k in range(i + 1, len(body)):
                u = body[k].split()
                if ws >= store_slots:
                    break
                if (u[0].startswith("v_") and not u[0].startswith(("v_cmp", "v_readfirstlane", "v_readlane"))) and len(u) > 1:
                    if regs(u[1]) & data:
                        out.append("%s: store-data: '%s' writes data of '%s' %d wait state(s) behind it" % (name, body[k], ins, ws))
                        break
                ws += wait_states(body[k])
        if re.match(r"s_(mov|and|or|andn2|xor|and_saveexec|or_saveexec|andn2_saveexec)_b64$", op) and len(t) > 1 and (
                t[1].startswith("exec") or "saveexec" in op):
            ws = 0
            for k in range(i + 1, len(body)):
                if ws >= 5:
                    break
                u = body[k].split()
                if "_dpp" in u[0] or "row_shr" in body[k] or "row_shl" in body[k] or "wave_sh" in body[k] or "row_bcast" in body[k]:
                    out.append("%s: exec-dpp: '%s' %d wait state(s) behind '%s'" % (name, body[k], ws, ins))
                    break
                ws += wait_states(body[k])
        if op.startswith("scratch_"):
            out.append("%s: scratch: '%s'" % (name, ins))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--store-slots", type=int, default=3, help="wait states behind a wide store in which its data must not be written")
    args = ap.parse_args()
    findings, nk = [], 0
    for path in args.files:
        for name, body in kernels(path):
            nk += 1
            findings += scan(name, body, args.store_slots)
    for f in findings:
        print(f)
    print("%d kernels scanned, %d findings" % (nk, len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
