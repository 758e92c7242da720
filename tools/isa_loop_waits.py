"""Compiler-inserted `s_waitcnt vmcnt(0)` inside the MFMA loops of a HIP source (no GPU needed).

The ring kernels count their direct-to-LDS DMA with hand-written `s_waitcnt vmcnt(N)` in inline assembly, which the compiler's
own wait-count pass cannot see: any ordinary load it still believes outstanding at a loop header (a conditional block before the
loop, an epilogue path that loads and never uses) makes it guard the first write of that register inside the loop with
`s_waitcnt vmcnt(0)` - a wait for the WHOLE ring on every iteration.  (Round 6: gemm3s_kernel had exactly that for three rounds.)

    python tools/isa_loop_waits.py plankassembly_amd/csrc/gemm.hip [kernel-name-substring]

Lists, per kernel, the loops that contain MFMAs and a compiler `vmcnt(0)` that is NOT directly behind a hand-written one
(those are redundant, not harmful).  Compiles with the product's flags (gemm.hip takes ~4 minutes)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
                        "-I" + os.path.join(ROOT, "include"), "-S", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def scan(lines, want=""):
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s*; @", l)]
    ends = [i for i, l in enumerate(lines) if ".amdhsa_kernel" in l or l.startswith(".Lfunc_end")]
    for st, name in starts:
        if want not in name:
            continue
        en = next((e for e in ends if e > st), len(lines))
        seg = lines[st:en]
        labels = {m.group(1): i for i, l in enumerate(seg) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        seen = set()
        for i, l in enumerate(seg):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            j = labels[m.group(1)]
            body = seg[j:i + 1]
            n_mfma = sum("v_mfma" in x for x in body)
            n_dma = sum(("global_load_lds" in x) or ("buffer_load" in x and " lds" in x) for x in body)
            if n_mfma == 0 or len(body) > 1200:
                continue
            in_asm, after_hand, hits = False, False, []
            for k, x in enumerate(body):
                if "ASMSTART" in x: in_asm = True; continue
                if "ASMEND" in x: in_asm = False; continue
                t = x.strip()
                if not t or t.startswith(";"):
                    continue
                if in_asm:
                    after_hand = "s_waitcnt" in t and "vmcnt(0)" in t
                    continue
                if "s_waitcnt" in t and "vmcnt(0)" in t and not after_hand:
                    hits.append(st + j + k + 1)
                after_hand = False
            key = (j, tuple(hits))
            if hits and key not in seen:
                seen.add(key)
                print(f"{name[:60]:60s} loop at line {st + j + 1:7d}  {i - j:5d} lines  {n_mfma:3d} MFMA  {n_dma:2d} DMA   compiler vmcnt(0) at lines {hits[:6]}")


if __name__ == "__main__":
    scan(assembly(os.path.abspath(sys.argv[1])), sys.argv[2] if len(sys.argv) > 2 else "")
