"""Reduce rocprofv3 --pmc counter_collection CSVs (one pass FETCH_SIZE, one pass WRITE_SIZE) to per-kernel HBM-side traffic per
launch, applying the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by 2x; checked here
on gn_apply_kernel, which reads exactly what it writes).  Optional third pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):
MFMA utilisation = busy cycles summed over all SIMDs / (GPU-active cycles x 256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE comes back
summed over the 8 XCDs (18.5 cycles per ns of kernel time = 8 x 2.31 GHz), hence the / 8.
Every kernel entry also gets `template` = the C++ template instance the way the library's launch records spell it (igemm_pp_kernel<192,320,1,4,0>),
and the summary carries `source_sha16` (pnpinversion_amd.build.source_hash() of the tree that produced the counters) and the profiled command.
usage: pmc_summary.py fetch.csv write.csv out.json [mfma.csv]      (env PMC_SOURCE: description of the profiled command)"""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnpinversion_amd.build import source_hash


def template_of(mangled):
    """_Z15igemm_pp_kernelILi192ELi320ELi1ELi4ELi0EEv5GemmPPKDF16_ -> igemm_pp_kernel<192,320,1,4,0>"""
    m = re.match(r"_Z\d+(igemm_\w+?_kernel)I((?:Li\d+E)+)E", mangled)
    if m:
        return "%s<%s>" % (m.group(1), ",".join(re.findall(r"Li(\d+)E", m.group(2))))
    m = re.match(r"(?:void )?(igemm_\w+?_kernel)<([\d, ]+)>", mangled)       # rocprofv3 demangles the signatures it can: "void igemm_pp_kernel<192, 320, 1, 4, 0>(GemmP)"
    return "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else None


def agg(fn, cn):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fn)):
        if r["Counter_Name"] != cn:
            continue
        k = r["Kernel_Name"]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d

f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {"unit": "bytes per launch (mean)", "correction": "FETCH_SIZE KB x 2 (gfx950), WRITE_SIZE KB x 1", "source_sha16": source_hash(),
       "source": os.environ.get("PMC_SOURCE", "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), one run each"), "kernels": {}}
for k in sorted(f, key=lambda k: -f[k][1]):
    n, s = f[k]
    wn, ws = w.get(k, [0, 0.0])
    out["kernels"][k] = {"launches": n, "fetch_bytes": 2 * 1024 * s / n, "write_bytes": 1024 * ws / max(wn, 1),
                         "traffic_bytes": 2 * 1024 * s / n + 1024 * ws / max(wn, 1)}
    if template_of(k):
        out["kernels"][k]["template"] = template_of(k)
if len(sys.argv) > 4:
    busy, act = agg(sys.argv[4], "SQ_VALU_MFMA_BUSY_CYCLES"), agg(sys.argv[4], "GRBM_GUI_ACTIVE")
    out["mfma_util_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), summed over the kernel's launches"
    for k, v in out["kernels"].items():
        if k in busy and k in act and act[k][1] > 0:
            v["mfma_util"] = busy[k][1] / (act[k][1] / 8.0 * 1024.0)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:10]:
    print("%-70s n=%6d  fetch %8.2f MB  write %8.2f MB  mfma_util %s" % (k[:70], v["launches"], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6,
                                                                      ("%.3f" % v["mfma_util"]) if "mfma_util" in v else "-"))
