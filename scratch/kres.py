"""Per-kernel register / LDS / spill table of a built object or library (amdhsa metadata notes).
usage: python scratch/kres.py skin-sm3_amd/csrc/build/conv_igemm.o [name-filter]"""
import re, subprocess, sys
path = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
# device code objects are bundled inside host objects: extract with clang-offload-bundler when needed
import tempfile, os
tmp = tempfile.mkdtemp()
out = os.path.join(tmp, "dev.co")
fb = os.path.join(tmp, "fb.bin")
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", path], capture_output=True)
r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
    out = path
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout

blocks = re.split(r"\n\s+- \.agpr_count", txt)
rows = []
for b in blocks[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", b) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    rows.append((name[:150], g("vgpr_count"), b.split("\n")[0].strip(": "), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
print("vgpr agpr sgpr vspill sspill lds wgsize  name")
for r in rows: print(*r[1:], r[0])
