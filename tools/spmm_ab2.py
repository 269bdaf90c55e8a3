"""Same-box A/B of two builds of the library (libricadi_hip_prev.so vs libricadi_hip.so)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)
for rep in range(2):
    for tag, lib in (("prev", "libricadi_hip_prev.so"), ("new", "libricadi_hip.so")):
        env = dict(os.environ, RICADI_LIB=os.path.join(root, "optconpy_amd", lib))
        for cfg in (("58", "16"), ("236", "16")):
            out = subprocess.run([sys.executable, os.path.join(here, "spmm_ab.py"), *cfg], env=env,
                                 capture_output=True, text=True).stdout
            print(tag, [l for l in out.splitlines() if "variant" in l][-1], flush=True)
