import os, sys
sys.path.insert(0, "/root/repo")
import torch
if len(sys.argv) > 1: os.environ["HIP_FORCE_DEV_KERNARG"] = sys.argv[1]
print("env", os.environ.get("HIP_FORCE_DEV_KERNARG"))
sys.argv = [sys.argv[0], "--shapes", "128:8:768:8", "--eager"]
os.environ.pop("X", None)
exec(open("/root/repo/scripts/bench_rankstep.py").read())
