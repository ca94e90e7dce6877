"""Build libsr_hip.so for gfx950 with hipcc (no torch, no cmake): one translation unit per .hip file,
objects cached under csrc/_obj by source mtime, linked in-tree next to the sources."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libsr_hip.so")
SOURCES = ["errors.cpp", "hostutil.cpp", "model.cpp", "igemm.hip", "norm.hip", "attention.hip", "eltwise.hip", "overlap.hip", "plan.hip", "raster.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# the rasterizer's fp32 evaluation order is its specification (bit parity with oracle/raster_ref.c): no FMA contraction
EXTRA = {"raster.hip": ["-ffp-contract=off"],
         # softmax row maxima never see a NaN (masked scores are -inf, every tile has a valid key): drop the canonicalising
         # v_max x,x the compiler otherwise adds around every fmaxf in the VALU-bound loop
         "attention.hip": ["-fno-honor-nans"]}
HEADERS = ["sr_common.h", os.path.join("..", "..", "include", "sr_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash():
    """sha256 over every source, header and this file (the flags live here): the identity of what libsr_hip.so must contain"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS) + [os.path.basename(__file__)]:
        fp = os.path.join(HERE, f)
        if os.path.exists(fp):
            h.update(f.encode())
            with open(fp, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:32]


def _needs(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(HERE, src), os.path.abspath(__file__)] + [os.path.join(HERE, h) for h in HEADERS]   # flags live in this file
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    jobs = []
    sh = source_hash()
    stamp = os.path.join(OBJ, "src.hash")
    stale_stamp = not os.path.exists(stamp) or open(stamp).read().strip() != sh
    for s in srcs:
        obj = os.path.join(OBJ, os.path.splitext(s)[0] + ".o")
        if force or _needs(s, obj) or (s == "errors.cpp" and stale_stamp):      # errors.cpp carries the hash
            cmd = [cc] + FLAGS + EXTRA.get(s, []) + ([f'-DSR_SRC_HASH="{sh}"'] if s == "errors.cpp" else []) + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(HERE, s), "-o", obj]
            jobs.append((s, cmd))

    def run(job):
        s, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        if verbose:
            print("compiled", s)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        # link under a private name, then rename: a process that dlopens LIB meanwhile sees the old or the new file, never half of one
        tmp = LIB + ".tmp%d" % os.getpid()
        r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.unlink(tmp)
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        os.replace(tmp, LIB)
    with open(stamp, "w") as f:
        f.write(sh)
    return LIB


def built_hash():
    """source hash the library lying in-tree was built from (the sidecar written after a successful link), or None"""
    stamp = os.path.join(OBJ, "src.hash")
    if os.path.exists(LIB) and os.path.exists(stamp):      # (the loader still checks sr_source_hash() of the image it maps)
        return open(stamp).read().strip()
    return None


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
