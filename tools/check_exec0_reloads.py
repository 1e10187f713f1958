"""Scan gfx950 assembly for vector instructions executed while EXEC is zero after a divergent loop.

A loop that retires lanes with `s_andn2_b64 exec, exec, sX` / `s_cbranch_execnz` falls through with EXEC = 0; everything
up to the `s_or_b64 exec, exec, ...` that ends the region runs for no lane.  LLVM's register allocator has been seen
(round 6, tvg_fh.hip: lo_ransac<K_T>) to place a spill RELOAD of a live value there - the value then stays whatever the
loop left in the register.  Usage:  hipcc ... -S --cuda-device-only x.hip -o x.s ; python tools/check_exec0_reloads.py x.s
Exit code 1 when something is found."""
import re, sys

def scan(path):
    lines = open(path).read().split("\n")
    ins = [(n, l.strip()) for n, l in enumerate(lines, 1) if l.strip() and not l.strip().startswith((";", ".", "//")) or re.match(r"^\.LBB\S+:", l.strip() or "")]
    found = []
    for i, (n, l) in enumerate(ins):
        if not l.startswith("s_cbranch_execnz"):
            continue
        back = [x[1] for x in ins[max(0, i - 4):i]]
        if not any(b.startswith("s_andn2_b64 exec, exec") for b in back):
            continue
        for n2, l2 in ins[i + 1:i + 40]:
            if re.match(r"s_(or|mov|and|andn2|xor)_b64 exec", l2) or "saveexec" in l2 or l2.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_swappc")):
                break
            if re.match(r"(v_|scratch_load|scratch_store|ds_|global_|flat_|buffer_)", l2) and not l2.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                found.append((n2, l2))
    return found

if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        f = scan(p)
        print(p, "vector instructions under EXEC = 0 after a loop:", len(f))
        for n, l in f[:40]:
            print("   line", n, l)
        rc |= bool(f)
    sys.exit(rc)
