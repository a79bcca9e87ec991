"""Parse rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into per-launch HBM traffic of the dominant
kernel (conv_igemm_kernel<bf16_t,128,128,...>), with the gfx950 FETCH_SIZE x2 correction for wide coalesced
streams (MI355X_MICROARCH.md, HBM section).  Usage: collect_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, hashlib, json, os, sys

def per_launch(d, counter):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "conv_igemm_kernel<bf16_t, 128, 128" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n

fetch, nf = per_launch(sys.argv[1], "FETCH_SIZE")
write, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "conv_igemm_kernel<bf16_t,128,128,*> (every wave-layout / stage / epilogue / segment instantiation)", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_KiB_raw_per_launch": fetch / nf, "write_KiB_per_launch": write / nw,
       "hbm_bytes_per_launch": (2.0 * fetch / nf + write / nw) * 1024.0,
       "note": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane streaming reads); WRITE_SIZE exact; "
               "separate --pmc passes; same bench.py command (B=256, 224x224, bf16)"}
_h = hashlib.sha256()
for _f in ("conv_igemm.hip", "conv_common.h", "common.h"):  # same recipe as bench.py:kernel_source_sha
    _h.update(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "skin-sm3_amd", "csrc", _f), "rb").read())
out["kernel_source_sha"] = _h.hexdigest()[:16]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
