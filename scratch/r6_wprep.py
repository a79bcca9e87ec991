"""weight_prep class time of a single-lane step for a few grid widths (SM3_WPREP_GX is read once per process)."""
import os, subprocess, sys
for gx in ("96", "192", "384", "768"):
    env = dict(os.environ, SM3_WPREP_GX=gx)
    subprocess.run([sys.executable, "bench.py", "--single-lane", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--breakdown", "/tmp/bd_wp.txt"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    line = [l for l in open("/tmp/bd_wp.txt") if l.startswith("weight_prep")]
    print("gx", gx, line[0].strip() if line else "?")
