"""Build libplank_hip.so (gfx950) in-tree with hipcc.  `python -m plankassembly_amd.build [--force]`."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libplank_hip.so")
OBJ = os.path.join(HERE, "csrc", "build")
SOURCES = ["gemm.hip", "rowops.hip", "attention.hip", "decode.hip", "runtime.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode()); h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = OUT, obj_dir: str = OBJ) -> str:
    """``extra_flags`` / ``out`` / ``obj_dir``: A/B builds of a variant (e.g. -DPA_ATTN_TRACE) next to the product library;
    a variant is loaded with PLANK_HIP_LIB=<out>."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, "pa_device.h"), os.path.join(os.path.dirname(HERE), "include", "plank_hip.h")]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    headers = sorted(set(headers))
    flags = [*FLAGS, *extra_flags]
    stamp = os.path.join(obj_dir, "stamp")
    dig = _digest(sorted(set(srcs + headers))) + hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == dig:
        return out
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        # per-object cache: an object is rebuilt only when its own source, a header or the flags changed
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        odig = _digest([src] + headers) + hashlib.sha256(" ".join(flags).encode()).hexdigest()
        ostamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == odig:
            return obj
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        with open(ostamp, "w") as f:
            f.write(odig)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"[plankassembly_amd] built {out} ({os.path.getsize(out) // 1024} KiB)")
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
