"""HIP-event profiler for the kernel wrappers in sm3hip.ops: per kernel class, summed launch time (events
recorded on the launch stream = torch's current stream), algorithmic FLOPs and algorithmic HBM bytes."""
import torch


class Profiler:
    def __init__(self, only=None, detail=False):
        self.only = only
        self.detail = detail  # conv tags carry the GEMM shape
        self.records = []  # (tag, flops, bytes, start, end)

    def wants(self, tag):
        return self.only is None or tag.split("|")[0] in self.only

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, tag, flops, nbytes, start):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((tag, flops, nbytes, start, e))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, flops, nbytes, s, e in self.records:
            d = out.setdefault(tag, {"flops": 0.0, "bytes": 0.0, "ms": 0.0, "launches": 0})
            d["flops"] += flops
            d["bytes"] += nbytes
            d["ms"] += s.elapsed_time(e)
            d["launches"] += 1
        return out

    def format_table(self, title):
        t = self.summary()
        total = sum(d["ms"] for d in t.values()) or 1.0
        lines = [title, f"{'kernel class':28s} {'launches':>8s} {'ms':>10s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s(alg)':>10s}"]
        for tag, d in sorted(t.items(), key=lambda kv: -kv[1]["ms"]):
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0
            gb = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0
            lines.append(f"{tag:28s} {d['launches']:8d} {d['ms']:10.3f} {100 * d['ms'] / total:6.1f} {tf:9.1f} {gb:10.0f}")
        lines.append(f"{'sum of kernel time':28s} {sum(d['launches'] for d in t.values()):8d} {total:10.3f}")
        return "\n".join(lines) + "\n"
