#!/usr/bin/env python3
"""Audit of the resident steppers' hand-issued loads (tg_step_stream_i8): an inline-asm load's VGPR destination counts as
written at the end of the statement as far as hipcc knows, so the compiler may read, copy or spill it before the data has
arrived.  The kernels tie those registers to their counted wait ("+v"), which pins ORDER, not register allocation: this
script compiles tg_kernels.hip to assembly and checks, for every s4 / s16 stream kernel, that no instruction between an
`sc1` load and the next arrival wait (vmcnt <= 2; <= 13 in the double-buffered lane kernel) touches the load's destination, and that the kernels use no scratch.
(s25_stream_kernel receives its tokens by LDS-DMA: no VGPR destination; it is checked for having no VGPR sc1 loads
outside the compiler-visible polls, which are followed by vmcnt(0).)
    python tools/audit_pending_loads.py        exit code 0 = clean"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mat_mul_amd import build  # noqa: E402


def assembly() -> list:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        flags = [f for f in build.FLAGS if f not in ("-fPIC",)]
        cmd = [build._hipcc(), *flags, "--cuda-device-only", "-S", f"-I{ROOT / 'include'}", f"-I{build.CSRC}",
               str(build.CSRC / "tg_kernels.hip"), "-o", str(out)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise SystemExit(res.stderr[-3000:])
        return out.read_text().split("\n")


def kernel_body(src, name):
    i = next(k for k, l in enumerate(src) if l.startswith(name + ":"))
    body = []
    for l in src[i:]:
        if not l.strip().startswith(";"):
            body.append(l.strip())
        if "s_endpgm" in l:
            break
    return body


def audit(src):
    problems = []
    names = [l.split(":")[0] for l in src if re.match(r"^_ZN2tg\d+s(4|16)_stream_kernel\w*:", l)]
    if len(names) < 4:
        problems.append(f"expected four s4/s16 stream kernels, found {len(names)}")
    for nm in names:
        pend = {}
        for idx, t in enumerate(kernel_body(src, nm)):
            m = re.match(r"global_load_(?:dword|dwordx3|dwordx4|sbyte|ubyte) (v\[?(\d+)(?::(\d+))?\]?), .*\bsc1\b", t)
            if m:
                lo = int(m.group(2))
                hi = int(m.group(3)) if m.group(3) else lo
                pend.update({r: idx for r in range(lo, hi + 1)})
                continue
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if w:
                # arrival waits: vmcnt <= 2 (the loads are the youngest operations); the one-game-per-lane kernel requests the
                # NEXT block's tokens before it works on this one, so its arrival waits leave this block's D + 4 stores and
                # a progress store outstanding (vmcnt 12 / 13); its publish waits (20 / 21) are not arrival waits
                if int(w.group(1)) <= (13 if "lanes" in nm else 2):
                    pend = {}
                continue
            if not pend:
                continue
            regs = set()
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                regs.update(range(int(a), int(b) + 1))
            regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t))
            hit = regs & set(pend)
            if hit and not t.startswith("global_load"):
                problems.append(f"{nm}: `{t}` touches v{sorted(hit)} before its load has been waited for")
        for l in src:
            pass
    for nm in names:
        meta = "\n".join(src[next(k for k, l in enumerate(src) if l.strip().startswith(f".amdhsa_kernel {nm}")):][:60])
        if not re.search(r"\.amdhsa_private_segment_fixed_size 0\b", meta):
            problems.append(f"{nm}: uses scratch (a spilled pending register would be garbage)")
    return names, problems


if __name__ == "__main__":
    names, problems = audit(assembly())
    for p in problems:
        print("PROBLEM", p)
    print(f"audited {len(names)} kernels: {'clean' if not problems else str(len(problems)) + ' problems'}")
    sys.exit(1 if problems else 0)
