"""Developer: build a VARIANT of the library next to the product one, for same-box A/B runs (tools/exp/ab_run.sh, DBFR_LIB=<path>).

    python tools/exp/build_variant.py <name> [--dev] [--src convz.hip[,conv2h.hip]] [-DMACRO[=v] ...]

Compiles the named sources (default convz.hip) with the extra macros into tools/exp/ab/obj_<name>/ and links them with the product's other
objects into tools/exp/ab/libdbfr_<name>.so (git-ignored; it travels to the GPU box with the tree).  --dev adds -DDBFR_DEV_VARIANTS (the
timing-only ablations / the s_memtime timeline).  The product library and its objects are not touched."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from diffbindfr_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    args = sys.argv[2:]
    srcs = ["convz.hip"]
    extra = []
    srcdir = None
    i = 0
    while i < len(args):
        if args[i] == "--dev":
            extra.append("-DDBFR_DEV_VARIANTS")
        elif args[i] == "--src":
            i += 1
            srcs = args[i].split(",")
        elif args[i] == "--srcdir":          # take the named sources from another directory (e.g. an older revision checked out under tools/exp/ab/src_<x>/)
            i += 1
            srcdir = os.path.abspath(args[i])
        else:
            extra.append(args[i])
        i += 1
    B.build(verbose=False)                                   # the product objects the variant links against
    out_dir = os.path.join(ROOT, "tools", "exp", "ab")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for s in B.SOURCES:
        obj = os.path.join(B.CSRC, os.path.splitext(s)[0] + ".o")
        if s in srcs:
            obj = os.path.join(obj_dir, os.path.splitext(s)[0] + ".o")
            cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "-c", os.path.join(srcdir or B.CSRC, s), "-o", obj, "-I", B.CSRC] + B.FLAGS + B.FILE_FLAGS.get(s, B.DEFAULT_FP) + extra
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stderr)
            for line in r.stderr.splitlines():
                if any(k in line for k in ("VGPRs:", "Spill", "ScratchSize", "Function Name")):
                    print(line.split("remark:")[-1].strip())
        objs.append(obj)
    lib = os.path.join(out_dir, f"libdbfr_{name}.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    print(lib)


if __name__ == "__main__":
    main()
