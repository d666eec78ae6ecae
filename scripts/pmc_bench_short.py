"""Short bench-shaped run for PMC passes: 2 steps of the one-stream schedule at batch 32 (no CPU leg)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-overlap"]))
