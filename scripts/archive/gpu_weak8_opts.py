"""Weak-layout rehearsal (8 x 10 M tets, team transport on one GPU) under several option sets."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for opt in sys.argv[1:]:
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_weak_rehearsal.py"), "8", "300,75,75", opt], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if "stokes its" in l]
    print(opt, "|", lines[0].strip()[60:] if lines else out.stderr[-300:], flush=True)
