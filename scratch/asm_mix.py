"""Static instruction mix per barrier-delimited phase of one kernel instantiation (device asm from hipcc -S).
usage: python scratch/asm_mix.py /tmp/conv.s '<substring of the mangled name>'"""
import collections, re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.rstrip().endswith(":") or (l.startswith("_Z") and pat in l and ": ;" in l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] and ".amdhsa_kernel" in "".join(lines[i:i + 8]))
body = lines[start:end + 1]
def mix(seg):
    c = collections.Counter()
    for l in seg:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"): continue
        op = l.split()[0]
        k = "mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else \
            "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        c[k] += 1
    return dict(c)
b = [i for i, l in enumerate(body) if "s_barrier" in l]
cuts = [0] + b + [len(body)]
print(lines[start][:160])
tot = collections.Counter()
for i in range(len(cuts) - 1):
    m = mix(body[cuts[i]:cuts[i + 1]])
    tot.update(m)
    print(f"  phase {i}: {m}")
print("  total:", dict(tot))
