"""Build recipe for libpnpi.so (hipcc, gfx950 only, in-tree so the .so travels with the repo snapshot).

    python -m pnpinversion_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libpnpi.so")
LIB_ABLATIONS = os.path.join(CSRC, "libpnpi_ablations.so")
SOURCES = ["gemm.hip", "norm.hip", "attn.hip", "step.hip", "bwd.hip", "api.hip"]
HEADERS = ["common.h", "ops.h", "model.h", "tile_table.inc", "igemm_dma.inc", "igemm_pp.inc", "api_weights.inc", "api_graph.inc", "api_backward.inc", "api_vae.inc", "api_ctrl.inc", os.path.join("..", "..", "include", "pnpi.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-unused-value"]
# Per-file flags.  step.hip: the reference rounds every multiply / add separately (no FMA contraction).  attn.hip: keep the MFMA
# accumulators in VGPRs (gfx950's unified file) -- the softmax between the two MFMAs is VALU work on the S accumulators, and in
# AGPR form every element costs a v_accvgpr_read (and a write to clear it): 96 of ~300 VALU instructions per 64-key tile.
EXTRA = {"step.hip": ["-ffp-contract=off"], "attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def source_hash():
    """Identity of the kernels a libpnpi.so is built from: sha256 over the HIP sources, their includes and the per-file flags (16 hex
    digits).  Committed profile summaries carry it; bench.py reports counters from a summary only if it matches the running tree."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + [x for x in HEADERS if not x.startswith("..")]):
        h.update(name.encode())
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS, sorted(EXTRA.items()))).encode())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True, ablations=False):
    """ablations: also compile the kernel instances that leave out the MFMAs / the DMA / the fragment reads (tools/pp_ablate.py,
    tools/profile_pp.sh, tuning igemm_vpp / igemm_sched / igemm_v128 = 11, 12, 15) -- not part of the product library: they go to their
    own object directory and their own library, csrc/libpnpi_ablations.so (load it with PNPI_LIBRARY=<path>), so the product .so is
    never overwritten by an ablation build."""
    OBJ = os.path.join(CSRC, "build_ablations" if ablations else "build")
    LIB = LIB_ABLATIONS if ablations else globals()["LIB"]
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        extra = list(EXTRA.get(os.path.basename(s), []))
        if os.path.basename(s)[:-4] in os.environ.get("PNPI_VGPR_FORM", "").split(","):   # experiments: PNPI_VGPR_FORM=gemm
            extra += ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
        if ablations:
            extra += ["-DPNPI_ABLATIONS=1"]
        cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        return o

    if jobs:
        if verbose:
            print("[pnpi build] compiling %d source(s) for gfx950" % len(jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
        if verbose:
            print("[pnpi build] linked %s" % LIB, file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--source-hash" in sys.argv:
        print(source_hash())
    else:
        print(build(force="--force" in sys.argv, ablations="--ablations" in sys.argv))
