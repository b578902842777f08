"""bench.py with per-shape kernel names and the side stream off: time and TFLOP/s per (kernel, layer shape).

  python profiles/tools/shape_prof.py [bench.py arguments]
"""
import json, subprocess, sys, os
env = dict(os.environ, RV3D_PROFILE_SHAPES="1", RV3D_OVERLAP="off")
out = subprocess.run([sys.executable, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))), "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extra"] + sys.argv[1:], env=env, capture_output=True, text=True).stdout
j = json.loads(out.strip().splitlines()[-1])
print("ms/step", j["ms_per_step"])
ks = j["kernels"]
tot = 0
for k, d in sorted(ks.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k[:95]:95s} n/step {d['launches']/5:5.1f} ms/step {d['ms']/5:7.2f} avg_us {d['avg_us']:8.1f} TF/s {d['tflops']:7.1f}")
    tot += d["ms"] / 5
print("sum", tot)
