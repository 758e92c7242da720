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


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, "pa_device.h"),
                   os.path.join(os.path.dirname(HERE), "include", "plank_hip.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    deps = sorted(set(deps))
    stamp = os.path.join(OBJ, "stamp")
    dig = _digest(deps)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"[plankassembly_amd] built {OUT} ({os.path.getsize(OUT) // 1024} KiB)")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
