"""Build libdbfr.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m diffbindfr_amd.build [--force]

hipcc cross-compiles without a GPU; the .so stays next to the sources so that it
travels to the GPU box with the tree (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdbfr.so")
SOURCES = ["api.cpp", "so3_host.cpp", "conv.hip", "conv2.hip", "conv2h.hip", "convz.hip", "graph.hip", "heads.hip", "export.hip", "mdn.hip", "probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
if os.environ.get("DBFR_BUILD_DEV") == "1":      # developer build: the timing-only kernel variants behind DBFR_CONV*_ABL / _VAR (wrong results)
    FLAGS.append("-DDBFR_DEV_VARIANTS")
# conv: no SLP vectorisation -- it pairs the contraction FMAs of different edge blocks into v_pk_fma_f32 with a v_mov
# shuffle per operand pair, which costs more vector-pipe slots next to the MFMAs than it saves and makes the kernel spill.
# (-Wno-array-bounds: the NB=1 instantiation indexes per-block arrays of length 1 inside `if (NB > 1)` branches)
# graph/heads: every fp op rounded separately (edge-in/out decisions and the SDE update mirror the oracle's
# operation order); conv: contraction allowed (fewer VALU slots next to the MFMAs; results are tolerance-checked)
FILE_FLAGS = {"conv.hip": ["-ffp-contract=fast", "-fno-slp-vectorize", "-Wno-array-bounds"] + ([f"-DCONV_NB={os.environ['DBFR_BUILD_NB']}"] if "DBFR_BUILD_NB" in os.environ else [])
              + ([f"-DCONV_PRIO={os.environ['DBFR_BUILD_PRIO']}"] if "DBFR_BUILD_PRIO" in os.environ else [])
              + ([f"-DCONV_XPF={os.environ['DBFR_BUILD_XPF']}"] if "DBFR_BUILD_XPF" in os.environ else [])}
FILE_FLAGS["mdn.hip"] = ["-ffp-contract=off"]
FILE_FLAGS["conv2.hip"] = ["-ffp-contract=fast", "-fno-slp-vectorize", "-Wno-array-bounds"]
FILE_FLAGS["conv2h.hip"] = FILE_FLAGS["conv2.hip"] + ["-Rpass-analysis=kernel-resource-usage"]
FILE_FLAGS["convz.hip"] = FILE_FLAGS["conv2h.hip"]
DEFAULT_FP = ["-ffp-contract=off"]


def source_hash(dev=False):
    """sha256 over every file the library is built from (sorted names + contents): compiled into api.cpp as
    DBFR_BUILD_ID and returned by dbfr_build_id(), so that a test can tell a stale prebuilt .so from the tree it sits in.
    A developer build (DBFR_BUILD_DEV=1) carries another id: tests/test_gpu_parity.py::test_native_library_is_loaded
    refuses it, and switching between the two rebuilds every object."""
    import hashlib
    h = hashlib.sha256()
    files = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip", ".h", ".inc"))])
    files.append(os.path.join(HERE, "..", "include", "dbfr.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    if dev:
        h.update(b"DBFR_DEV_VARIANTS")
    return h.hexdigest()[:16]


def check_m0(hipcc, src):
    """convz.hip writes m0 in inline assembly (`s_mov_b32 m0, <sgpr>` in front of each `global_load_lds_dword`: the LDS base of an LDS-DMA gather)
    without declaring it clobbered -- hipcc rejects the clobber of a reserved register.  That is sound only while the COMPILER emits nothing that
    reads or writes m0 in that kernel (movrel indexing, LDS-DMA builtins, GWS, sendmsg ...): this check compiles the file to assembly and fails the
    build on any m0 operand that is not one of ours (verified on ROCm 7.2.0 / hipcc of this image)."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "convz.s")
        cmd = [hipcc, "-x", "hip", "-S", "--cuda-device-only", src, "-o", out] + [f for f in FLAGS + FILE_FLAGS["convz.hip"] if not f.startswith("-Rpass")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc -S failed (m0 check):\n" + r.stderr)
        lines = open(out).read().splitlines()
    ours = re.compile(r"^\s*s_mov_b32 m0, s\d+\s*$")
    bad = []
    for i, l in enumerate(lines):
        code = l.split(";")[0]
        if re.search(r"\bm0\b", code) and not ours.match(code):
            bad.append(f"{i + 1}: {l.strip()}")
        if ours.match(code) and not any("global_load_lds_dword" in x for x in lines[i + 1:i + 4]):
            bad.append(f"{i + 1}: s_mov_b32 m0 without an LDS-DMA load behind it: {l.strip()}")
    if bad:
        raise RuntimeError("convz.hip: the compiler (or an edit) uses m0 outside the LDS-DMA gathers -- the undeclared writes of prefetch_x are no longer safe:\n  " + "\n  ".join(bad[:10]))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "residue_tables.inc"), os.path.join(HERE, "..", "include", "dbfr.h")]
    objs, jobs = [], []
    bid = source_hash(dev="-DDBFR_DEV_VARIANTS" in FLAGS)
    stamp = os.path.join(CSRC, ".build_id")
    if not os.path.exists(stamp) or open(stamp).read() != bid:      # any source changed: api.cpp carries the id
        open(stamp, "w").write(bid)
    headers.append(stamp)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, "-x", "hip", "-c", src, "-o", obj] + FLAGS + FILE_FLAGS.get(s, DEFAULT_FP) +
                        ([f'-DDBFR_BUILD_ID="{bid}"'] if s == "api.cpp" else []))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    if any(j[4].endswith("convz.hip") for j in jobs):
        check_m0(hipcc, os.path.join(CSRC, "convz.hip"))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
